#!/usr/bin/env python3
"""Build-time check of csrc/fused_bwd.hip's counted waits, run by the Makefile on the assembly of the very object it links
(hipcc -save-temps) and by tests/test_host_cpu.py.

The weight-gradient waves retire their weight-ring LDS-DMA with `s_waitcnt vmcnt(N)` in front of every phase barrier; N is the number
of vector-memory instructions the wave issues AFTER the DMA of the piece the phase reads: the next piece's DMA and the operand fetches
of the two phases in between (fused_bwd.hip: Keep<>).  The fetches are ordinary loads emitted by the compiler: if a toolchain merges,
splits, duplicates or spills anything in that loop the counts are wrong and the chain reads weights that have not landed -- silently
wrong gradients.  Checked per phase of the weight-gradient loop: DMA instructions, loads in the fetch phases only, no store, no scratch
access, and every wait immediate against the Keep table.   usage: check_fused_counts.py <listing.s>   (exit status 0 = ok)"""
import re
import sys

KERNEL = '_ZN3hgn21edge_bwd_fused_kernelILi6EEEvNS_9FusedArgsE'
KERNEL3 = '_ZN3hgn2f322edge_bwd_fused3_kernelENS_9FusedArgsE'      # csrc/fused_bwd3.hip: two fp16 terms, 4 DMA instructions per piece and wave
PHASES = 12
DMA_PER_PHASE = 6            # Ring<6>::DPW
LOADS_PER_FETCH = 8
FETCH_PHASES = {1, 4, 6, 11}  # wg_fetches()


def check(text: str, kernel: str = KERNEL, DMA_PER_PHASE: int = DMA_PER_PHASE, fetch_dma: int = 0) -> None:
    LOADS = 0 if fetch_dma else LOADS_PER_FETCH           # fused_bwd3: the operand fetches are LDS-DMA too (fetch_dma per fetch)
    at = text.index(kernel + ':')
    body = text[at:text.index('.Lfunc_end', at)]
    if 'scratch_' in body:
        raise AssertionError('register spills in the fused backward: their memory traffic breaks the counted waits')
    lines = [l.strip() for l in body.splitlines()]
    total = PHASES * DMA_PER_PHASE + fetch_dma * len(FETCH_PHASES)
    heads = [i for i, l in enumerate(lines) if 'Loop Header' in l]
    dma_lines = [i for i, l in enumerate(lines) if l.startswith('global_load_lds_dwordx4')]
    # the weight-gradient loop: the innermost loop that holds a whole tile's LDS-DMA instructions (the prologue's lie before it)
    start = max(h for h in heads if sum(1 for d in dma_lines if d > h) >= total)
    end = next(i for i in range(start, len(lines))
               if lines[i].startswith('s_cbranch') and sum(1 for d in dma_lines if start < d < i) >= total)
    loop = lines[start:end]
    phases, cur = [], None
    for l in loop:
        if l.startswith('s_barrier'):
            cur = {'dma': 0, 'loads': 0, 'stores': 0}
            phases.append(cur)
        elif cur is not None:
            if l.startswith('global_load_lds'):
                cur['dma'] += 1
            elif l.startswith(('global_load', 'buffer_load', 'flat_load')):
                cur['loads'] += 1
            elif l.startswith(('global_store', 'buffer_store', 'flat_store', 'global_atomic')):
                cur['stores'] += 1
    waits = [int(re.search(r'vmcnt\((\d+)\)', l).group(1)) for l in loop if l.startswith('s_waitcnt vmcnt(') and 'lgkmcnt(0)' in l]
    if len(phases) != PHASES or len(waits) != PHASES:
        raise AssertionError(f'expected {PHASES} phases / waits in the weight-gradient loop, found {len(phases)} / {len(waits)}')
    for p, ph in enumerate(phases):
        want_loads = LOADS if p in FETCH_PHASES else 0
        want_dma = DMA_PER_PHASE + (fetch_dma if p in FETCH_PHASES else 0)
        if ph['dma'] != want_dma or ph['stores'] != 0 or ph['loads'] != want_loads:
            raise AssertionError(f'phase {p}: {ph}, expected {want_dma} DMA, {want_loads} loads, 0 stores')
        per_fetch = fetch_dma or LOADS_PER_FETCH
        keep = DMA_PER_PHASE + per_fetch * (((p - 2) % PHASES) in FETCH_PHASES) + per_fetch * (((p - 1) % PHASES) in FETCH_PHASES)
        if waits[p] != keep:
            raise AssertionError(f'phase {p}: s_waitcnt vmcnt({waits[p]}), Keep<> says {keep}')


def check3(text: str) -> None:
    """csrc/fused_bwd3.hip: two weight-gradient tile loops (wgrad_role<true>: ring waves, 8 LDS-DMA instructions per phase, vmcnt(8) in
    front of every barrier; wgrad_role<false>: rows waves, 8 in the four fetch phases, vmcnt(8) in front of those phases' barriers only).
    Checked on both loops: no scratch anywhere in the kernel, no ordinary load / store inside a loop, the DMA count of every phase, every
    counted wait of a loop is vmcnt(8) and their number is 12 / 4."""
    at = text.index(KERNEL3 + ':')
    body = text[at:text.index('.Lfunc_end', at)]
    if 'scratch_' in body:
        raise AssertionError('register spills in the fused backward (two-term fp16 form): their memory traffic breaks the counted waits')
    lines = [l.strip() for l in body.splitlines()]
    heads = [i for i, l in enumerate(lines) if 'Loop Header' in l]
    found = {}
    for h in heads:
        bars = [i for i in range(h, len(lines)) if lines[i].startswith('s_barrier')]
        if len(bars) < PHASES:
            continue
        try:
            end = next(i for i in range(bars[PHASES - 1], len(lines)) if lines[i].startswith(('s_cbranch', 's_branch')))
        except StopIteration:
            continue
        if any(re.match(r'^\.?L?BB\d+_\d+:', lines[i]) for i in range(h + 1, end)):
            continue                                                  # (a loop with inner labels: not one of the straight-line tile loops)
        seg_dma = [sum(1 for l in lines[bars[p_]:(bars[p_ + 1] if p_ + 1 < PHASES else end)] if l.startswith('global_load_lds')) for p_ in range(PHASES)]
        if sum(seg_dma) == 0:
            continue                                                  # the chain's loop
        other = [l for l in lines[h:end] if l.startswith(('global_load', 'buffer_load', 'flat_load', 'global_store', 'buffer_store', 'flat_store', 'global_atomic'))
                 and not l.startswith('global_load_lds')]
        waits = [int(re.search(r'vmcnt\((\d+)\)', l).group(1)) for l in lines[h:end] if l.startswith('s_waitcnt vmcnt(')]
        ring = [8] * PHASES
        rows = [8 if p_ in FETCH_PHASES else 0 for p_ in range(PHASES)]
        kind = 'ring' if seg_dma == ring else ('rows' if seg_dma == rows else None)
        if kind is None or other:
            raise AssertionError(f'weight-gradient loop at line {h}: DMA per phase {seg_dma}, other vector memory {other[:2]}')
        want = PHASES if kind == 'ring' else len(FETCH_PHASES)
        if any(w != 8 for w in waits) or len(waits) != want:
            raise AssertionError(f'{kind} loop: counted waits {waits} (expected {want} x vmcnt(8))')
        found[kind] = True
    if set(found) != {'ring', 'rows'}:
        raise AssertionError(f'expected one ring loop and one rows loop, found {sorted(found)}')


# ---- csrc/mlp6.hip: mlp6_fwd_edge_kernel (quarter-pipelined weight ring, mlp6_device.h: gemm6q) -------------------------------------
# Its waits `s_waitcnt vmcnt(N) lgkmcnt(0)` with N > 0 leave the N youngest vector-memory operations of the wave in flight while the
# LDS-DMA of a ring piece -- issued BEFORE them -- must have landed.  Safe exactly when, walking back from the wait, the first N
# vector-memory instructions are reached without passing a label (another path could join with a different count), are not LDS-DMA,
# and the kernel has no scratch access (a spill reload is a vector-memory operation nobody counted).
EDGE_KERNELS = ['_ZN3hgn20mlp6_fwd_edge_kernelILi%dEEEv13hgn_mlp_fwd_t' % n for n in (6, 3)]
EDGE_KERNEL = EDGE_KERNELS[0]
_VMEM = ('global_load', 'global_store', 'global_atomic', 'buffer_', 'flat_', 'scratch_')


def check_edge_forward(text: str, kernel: str = None) -> int:
    at = text.index((kernel or EDGE_KERNEL) + ':')
    body = text[at:text.index('.Lfunc_end', at)]
    if 'scratch_' in body:
        raise AssertionError('register spills in mlp6_fwd_edge_kernel: their reloads break the counted waits')
    lines = [l.strip() for l in body.splitlines()]
    n_checked = 0
    for i, l in enumerate(lines):
        m = re.match(r's_waitcnt vmcnt\((\d+)\) lgkmcnt\(0\)', l)
        if not m or int(m.group(1)) == 0:
            continue
        keep, seen, j = int(m.group(1)), 0, i - 1
        while seen < keep:
            if j < 0:
                raise AssertionError(f'wait at line {i}: ran out of instructions after {seen} of {keep}')
            x = lines[j]
            if re.match(r'^\.?L?BB\d+_\d+:', x) or x.startswith('s_cbranch') or x.startswith('s_branch'):
                raise AssertionError(f'wait vmcnt({keep}) at line {i}: control flow joins / leaves within its {keep} youngest operations')
            if x.startswith('global_load_lds'):
                raise AssertionError(f'wait vmcnt({keep}) at line {i}: an LDS-DMA is among the {keep} operations it leaves in flight')
            if x.startswith(_VMEM):
                seen += 1
            j -= 1
        n_checked += 1
    if n_checked < 3:
        raise AssertionError(f'expected at least 3 counted waits in mlp6_fwd_edge_kernel, found {n_checked}')
    return n_checked


if __name__ == '__main__':
    try:
        text = open(sys.argv[1]).read()
        if KERNEL + ':' in text:
            check(text)
            print('check_fused_counts: ok (fused backward: 12 phases, counted waits match the emitted instructions)')
        if KERNEL3 + ':' in text:
            check3(text)
            print('check_fused_counts: ok (fused backward, two-term fp16 form: 12 phases, counted waits match the emitted instructions)')
        for kernel in EDGE_KERNELS:
            if kernel + ':' in text:
                n = check_edge_forward(text, kernel)
                print(f'check_fused_counts: ok (edge forward {kernel[-30:-25]}: {n} counted waits leave only younger, unconditional operations in flight; no scratch)')
    except (AssertionError, ValueError, StopIteration) as e:
        print('check_fused_counts: FAILED:', e, file=sys.stderr)
        sys.exit(1)
