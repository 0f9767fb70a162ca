"""Segment-reduce (scatter-add aggregation) alone on the benchmark shape, for PMC passes (FETCH_SIZE / WRITE_SIZE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
from hgn_amd import ops, topology, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = synthetic.batch([synthetic.grid_graph(seed=i % 4) for i in range(B)])
es = g.edge_sets[0]
N = g.node_features[0].shape[0]; E = es.senders.shape[0]
dev = torch.device('cuda')
topo = topology.EdgeTopology(es.senders, es.receivers, N, dev)
e = torch.randn(E, 128, device=dev)
with torch.no_grad():
    for _ in range(5):
        a = ops.aggregate([e], [(None, topo.r.rowptr, topo.rcv)], ('sum',))
        b = ops.aggregate([e], [(topo.s.perm, topo.s.rowptr, topo.s.seg)], ('sum',))
        c = ops.aggregate([e], [(None, topo.r.rowptr, topo.rcv)], ('sum', 'mean', 'max', 'min'))
torch.cuda.synchronize()
print('N', N, 'E', E, 'algorithmic bytes sum:', 4 * 128 * E + 4 * (N + 1) + 4 * 128 * N, 'pna:', 4 * 128 * E + 4 * (N + 1) + 4 * 4 * 128 * N)
