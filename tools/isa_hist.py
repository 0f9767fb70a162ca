#!/usr/bin/env python3
"""Static instruction histogram of one kernel in a hipcc -S listing (tools/isa_hist.py file.s kernel-substring [top]).
Classes: mfma, valu (by opcode), salu, lds, vmem, smem, wait/barrier.  A static count: loops are counted once."""
import collections
import re
import sys


def kernel_lines(path, key):
    out, on = [], False
    for ln in open(path):
        if re.match(r'^_Z\S*:', ln):
            on = key in ln.split(':')[0]
            continue
        if on:
            if ln.startswith('.Lfunc_end') or '.end_amdhsa_kernel' in ln:
                break
            out.append(ln)
    return out


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_load') or op.startswith('s_buffer_load'): return 'smem'
    if op.startswith(('s_waitcnt', 's_barrier', 's_nop', 's_sleep', 's_setprio', 's_sched')): return 'wait'
    if op.startswith(('s_cbranch', 's_branch', 's_endpgm')): return 'branch'
    if op.startswith('s_'): return 'salu'
    return 'other'


def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    cls, ops = collections.Counter(), collections.Counter()
    for ln in kernel_lines(path, key):
        s = ln.strip()
        if not s or s.startswith((';', '.', '//')) or s.endswith(':'):
            continue
        op = s.split()[0]
        c = classify(op)
        cls[c] += 1
        ops[(c, op)] += 1
    print(dict(cls))
    for (c, op), n in ops.most_common(top):
        print(f'{n:6d} {c:5s} {op}')


if __name__ == '__main__':
    main()
