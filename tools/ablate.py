"""Diagnostic (HGN_STAMP build): time the edge-mode forward kernel with parts switched off."""
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
g = synthetic.batch([synthetic.grid_graph(seed=i % 4) for i in range(64)])
es = g.edge_sets[0]
N = g.node_features[0].shape[0]; E = es.senders.shape[0]
dev = torch.device('cuda')
topo = topology.EdgeTopology(es.senders, es.receivers, N, dev)
m = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges']).to(dev)
w = modules.weights_of(m.processor.graphnet_blocks[0].edge_models['mesh_edges'], 384)
h = torch.randn(N, 128, device=dev, requires_grad=True)
e = torch.randn(E, 128, device=dev, requires_grad=True)
ops.prof_enable(True)
def run(flags, grad, label):
    L.hgn_debug_set_flags(flags)
    ops.prof_reset()
    for _ in range(5):
        if grad:
            y = ops.edge_block(h, e, topo, w)
        else:
            with torch.no_grad():
                y = ops.edge_block(h, e, topo, w)
    torch.cuda.synchronize()
    k = ops.prof_collect()['mlp_fwd_edge']
    print('%-50s %.3f ms' % (label, k['ms'] / k['count']))
run(0, True, 'full (training: saves z1,z2,xhat)')
run(0, False, 'inference (no saves)')
run(1, True, 'no MFMA (memory only, training)')
run(2, True, 'no stores')
run(4, True, 'no P gathers')
run(8, True, 'no e loads')
run(2 | 4 | 8, True, 'MFMA + weight DMA only')
run(1 | 2 | 4 | 8, True, 'weight DMA + barriers only')
run(1 | 2 | 4 | 8 | 16, True, 'barriers only (no DMA)')
run(1 | 2 | 4 | 8 | 32, True, 'DMA only (no barriers)')
run(1 | 2 | 4 | 8 | 16 | 32, True, 'nothing (bias/LN loads, launch)')
run(16, True, 'full minus DMA (wrong results)')
run(1 | 2, True, 'loads only')
run(1 | 4 | 8, True, 'stores only')
L.hgn_debug_set_flags(0)
