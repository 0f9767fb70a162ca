"""Diagnostic: time the edge block (fwd / bwd / wgrad kernels) with an experimental build of the library.
   python tools/exp_time.py NAME   (loads tools/_build/libhgn_mp_NAME.so; NAME=ship uses the shipped library)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
from hgn_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else 'ship'
if name != 'ship':
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', '_build', f'libhgn_mp_{name}.so')
from hgn_amd import ops, topology, synthetic, modules
import hgn_amd
g = synthetic.batch([synthetic.grid_graph(seed=i % 4) for i in range(64)])
es = g.edge_sets[0]
N = g.node_features[0].shape[0]; E = es.senders.shape[0]
dev = torch.device('cuda')
topo = topology.EdgeTopology(es.senders, es.receivers, N, dev)
m = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges']).to(dev)
blk = m.processor.graphnet_blocks[0]
w = modules.weights_of(blk.edge_models['mesh_edges'], 384)
wn = modules.weights_of(blk.node_model_cross, 256)
h = torch.randn(N, 128, device=dev, requires_grad=True)
e = torch.randn(E, 128, device=dev, requires_grad=True)
ops.prof_enable(True)
for rep in range(2):
    ops.prof_reset()
    for _ in range(5):
        y, agg = ops.edge_block(h, e, topo, w, ('sum',)) if False else (ops.edge_block(h, e, topo, w), None)
        agg = ops.aggregate([y], [(None, topo.r.rowptr, topo.rcv)], ('sum',))
        hn = ops.fused_mlp([h, agg], wn, None, 0)
        (hn.sum() + y.sum()).backward()
    torch.cuda.synchronize()
k = ops.prof_collect()
print(name, ' '.join('%s=%.3f' % (n, v['ms'] / max(v['count'], 1)) for n, v in k.items() if v['count']))
