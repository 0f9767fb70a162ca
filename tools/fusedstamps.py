"""Phase timeline of the fused edge backward (csrc/fused_bwd.hip) from a -DHGN_FUSED_STAMPS diagnostic build:
    HGN_LAB_EXTRA=-DHGN_FUSED_STAMPS tools/lab/build_lab.sh && HGN_LIB=tools/_build/libhgn_mp_lab.so python tools/fusedstamps.py
Prints, for one mid-launch workgroup's 11th tile, the shader-clock deltas of wave 0 (chain: before / after every phase barrier) and
waves 4 / 6 (weight gradients: before barrier, after barrier, after publish, after DMA issue + fetch, per phase)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
exec(open(os.path.join(ROOT, 'tools', 'fusedbench.py')).read().split("print('rows'")[0])
from hgn_amd import _lib
L = _lib.lib()
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

