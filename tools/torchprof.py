"""Diagnostic: which PyTorch-side (non-hgn) GPU kernels run inside one eager training step, by aten op and input shapes.
    python tools/torchprof.py [--batch 128]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
import hgn_amd
from hgn_amd import synthetic, parallel
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=128)
ap.add_argument('--arch', default='none'); ap.add_argument('--agg', default='sum'); ap.add_argument('--layers', type=int, default=15)
ap.add_argument('--clusters', type=int, default=0); a = ap.parse_args()
dev = torch.device('cuda')
big = synthetic.batch([synthetic.grid_graph(seed=i % 4, clusters=a.clusters) for i in range(a.batch)])
graph = hgn_amd.MultiGraph([x.to(dev) for x in big.node_features],
                           [hgn_amd.EdgeSet(e.name, e.features.to(dev), e.senders.to(dev), e.receivers.to(dev)) for e in big.edge_sets])
N = graph.node_features[0].shape[0]
target = torch.randn(N, 3, device=dev); mask = torch.ones(N, dtype=torch.bool, device=dev)
torch.manual_seed(0)
model = hgn_amd.MeshGraphNet(3, 128, 2, a.agg, a.layers, a.arch, [e.name for e in graph.edge_sets]).to(dev)
with torch.no_grad():
    model(graph)
tr = parallel.DataParallelTrainer(model, wgrad_stream=False)
for _ in range(2):
    tr.step(graph, target, mask)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(graph, target, mask)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='cuda_time_total', row_limit=40, max_name_column_width=60,
                                                         max_shapes_column_width=70))
