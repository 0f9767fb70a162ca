"""Kernel-level micro-benchmark driver (used under rocprofv3 for PMC passes): runs the edge block fwd+bwd and the
segment reduce a few times on a flag_simple-shape batch.  Not part of the product or the tests."""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
from hgn_amd import ops, topology, synthetic, modules
import hgn_amd

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--iters', type=int, default=3)
ap.add_argument('--what', default='edge,seg')
a = ap.parse_args()
g = synthetic.batch([synthetic.grid_graph(seed=i % 4) for i in range(a.batch)])
es = g.edge_sets[0]
N = g.node_features[0].shape[0]
E = es.senders.shape[0]
dev = torch.device('cuda')
topo = topology.EdgeTopology(es.senders, es.receivers, N, dev)
torch.manual_seed(0)
m = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges']).to(dev)
blk = m.processor.graphnet_blocks[0]
w = modules.weights_of(blk.edge_models['mesh_edges'], 384)
wn = modules.weights_of(blk.node_model_cross, 256)
h = torch.randn(N, 128, device=dev, requires_grad=True)
e = torch.randn(E, 128, device=dev, requires_grad=True)
for it in range(a.iters):
    if 'edge' in a.what:
        y = ops.edge_block(h, e, topo, w)
        if 'seg' in a.what:
            agg = ops.aggregate([y], [(None, topo.r.rowptr, topo.rcv)], ('sum',))
            hn = ops.fused_mlp([h, agg], wn, None, 0)
            (hn.sum() + y.sum()).backward()
        else:
            y.sum().backward()
    elif 'seg' in a.what:
        with torch.no_grad():
            agg = ops.aggregate([e], [(None, topo.r.rowptr, topo.rcv)], ('sum',))
torch.cuda.synchronize()
print('done', N, E)
