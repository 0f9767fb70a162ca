"""Diagnostic: where does a wave of mlp_fwd_kernel (edge mode) spend its cycles?  Uses the -DHGN_STAMP build."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
from hgn_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', '_build', 'libhgn_mp_stamp.so')
from hgn_amd import ops, topology, synthetic, modules
import hgn_amd
L = _lib.lib()
L.hgn_debug_set_stamps.argtypes = [C.c_void_p]
g = synthetic.batch([synthetic.grid_graph(seed=i % 4) for i in range(64)])
es = g.edge_sets[0]
N = g.node_features[0].shape[0]; E = es.senders.shape[0]
dev = torch.device('cuda')
topo = topology.EdgeTopology(es.senders, es.receivers, N, dev)
m = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges']).to(dev)
w = modules.weights_of(m.processor.graphnet_blocks[0].edge_models['mesh_edges'], 384)
h = torch.randn(N, 128, device=dev, requires_grad=True)
e = torch.randn(E, 128, device=dev, requires_grad=True)
y = ops.edge_block(h, e, topo, w)          # warm
torch.cuda.synchronize()
st = torch.zeros(4096 * 4 * 16, dtype=torch.int64, device=dev)
assert L.hgn_debug_set_stamps(st.data_ptr()) == 0
y = ops.edge_block(h, e, topo, w)
torch.cuda.synchronize()
L.hgn_debug_set_stamps(None)
s = st.view(4096, 4, 16).cpu().double()
dt = s[:, :, 4] - s[:, :, 0]
drt = s[:, :, 9] - s[:, :, 8]
ok = drt > 0
print('wave lifetime: %.0f shader ticks, %.1f us (realtime 100 MHz)' % (dt[ok].mean(), drt[ok].mean() / 100.0))
print('in-kernel clock = d(memtime)/d(memrealtime) x 100 MHz = %.0f MHz (median %.0f)' % ((dt[ok] / drt[ok]).mean() * 100, (dt[ok] / drt[ok]).median() * 100))
names = ['stage 1 (loads + gemm)', 'stage 2', 'stage 3', 'LN + stores']
for i, n in enumerate(names):
    d = (s[:, :, i + 1] - s[:, :, i])[ok]
    print('%-30s %8.0f ticks (%.1f%%)' % (n, d.mean(), 100 * d.mean() / dt[ok].mean()))
# ---- are workgroups phase-locked chip-wide?  distribution of workgroup start times (wave 0 of each block)
t0 = s[:, 0, 0]
t0 = t0[t0 > 0]
t0 = (t0 - t0.min()) / 2184.0       # us
import numpy as np
h, edges = np.histogram(t0.numpy(), bins=40)
print('start-time histogram of the first 4096 workgroups (us since first start):')
for c, e in zip(h, edges[:-1]):
    print('  %7.1f us  %s' % (e, '#' * int(c // 8)))
