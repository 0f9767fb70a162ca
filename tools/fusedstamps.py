"""Phase timeline of the fused edge backward (csrc/fused_bwd3.hip; csrc/fused_bwd.hip with --precision fp32-bf16x3) from a
-DHGN_FUSED_STAMPS diagnostic build:
    HGN_ABL_EXTRA=-DHGN_FUSED_STAMPS bash tools/build_ablations.sh f0 && HGN_LIB=hyper-graph-nets_amd/hgn_amd/abl/libhgn_mp_ablf0.so python tools/fusedstamps.py
Prints, for one mid-launch workgroup's 11th tile, the shader-clock deltas of wave 0 (chain: before / after every phase barrier) and
waves 4 / 6 (weight gradients: before barrier, after barrier, after publish, after DMA issue + fetch, per phase)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
exec(open(os.path.join(ROOT, 'tools', 'fusedbench.py')).read().split("for rep in range(a.reps)")[0])
run('stamped build')
from hgn_amd import _lib
L = _lib.lib()
L.hgn_debug_fused_stamps.restype = C.c_int
L.hgn_debug_fused_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 192)()
assert L.hgn_debug_fused_stamps(buf) == 0
st = list(buf)
ch = st[:27]
t0 = ch[0]
print('chain wave 0 (cycles from tile start):')
names = ['tile start', 'LN done', 'split+G written'] + [f'{"after" if i % 2 else "before"} barrier {i // 2}' for i in range(0, 24)]
for i, v in enumerate(ch):
    print(f'  {i:2d} {v - t0:7d}  (+{v - ch[i - 1] if i else 0:6d})')
for role, w in ((1, 4),):
    s = st[role * 64: role * 64 + 48]
    print(f'wgrad wave {w}: per phase [wait+barrier, publish, dma+fetch, block -> next phase start]  (start rel. to chain tile start: {s[0] - t0})')
    for P in range(12):
        nxt = s[4 * (P + 1)] if P < 11 else None
        print(f'  phase {P:2d}: bar {s[4 * P + 1] - s[4 * P]:6d}  pub {s[4 * P + 2] - s[4 * P + 1]:6d}  dma/fetch {s[4 * P + 3] - s[4 * P + 2]:6d}  block {(nxt - s[4 * P + 3]) if nxt else -1:6d}')


# every wave's arrival at every phase barrier (fused_bwd3: WSTAMP), relative to the earliest arrival at that barrier
if hasattr(L, 'hgn_debug_fused_wave_stamps'):
    wb = (C.c_ulonglong * 256)()
    L.hgn_debug_fused_wave_stamps.restype = C.c_int
    L.hgn_debug_fused_wave_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
    assert L.hgn_debug_fused_wave_stamps(wb) == 0
    w = [list(wb[32 * i: 32 * i + 32]) for i in range(8)]
    print('arrival at barrier p, cycles after the first wave to arrive (waves 0-3 chain, 4-7 weight gradients; for 4-7: before the vmcnt wait / after it):')
    for p_ in range(12):
        arr = [w[i][p_] for i in range(4)] + [w[i][12 + p_] for i in range(4, 8)]
        t0_ = min(a for a in arr if a)
        print(f'  barrier {p_:2d}: ' + ' '.join(f'{a - t0_:6d}' for a in arr[:4]) + '  |  ' +
              ' '.join(f'{w[i][p_] - t0_:6d}/{w[i][12 + p_] - t0_:6d}' for i in range(4, 8)))
