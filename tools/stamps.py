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
st = torch.zeros(4096 * 8 * 16, dtype=torch.int64, device=dev)
assert L.hgn_debug_set_stamps(st.data_ptr()) == 0
y = ops.edge_block(h, e, topo, w)
torch.cuda.synchronize()
L.hgn_debug_set_stamps(None)
s = st.view(4096, 8, 16).cpu().double()
names = ['start->pre-bfrag(b1+gathers+DMA+sync)', 'bfrag loads wait', 'stage1 MFMA', 'relu+store z1+sync', 'DMA W2+sync', 'stage2 MFMA',
         'relu+store z2+sync', 'DMA W3+sync', 'stage3 MFMA', 'LN+stores']
d = s[:, :, 1:11] - s[:, :, 0:10]
tot = s[:, :, 10] - s[:, :, 0]
print('wave lifetime cycles: mean %.0f  (s_memtime ticks)' % tot.mean())
for i, n in enumerate(names):
    print('%-45s mean %8.0f  (%.1f%%)' % (n, d[:, :, i].mean(), 100 * d[:, :, i].mean() / tot.mean()))
