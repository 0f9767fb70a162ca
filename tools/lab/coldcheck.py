"""Where the time of an eager training step with a cleared topology cache goes (bench.py: cold_step, second figure)."""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
import hgn_amd
from hgn_amd import synthetic, parallel, topology
dev = torch.device('cuda')
g = synthetic.batch([synthetic.grid_graph(seed=i % 4) for i in range(128)])
graph = hgn_amd.MultiGraph([x.to(dev) for x in g.node_features], [hgn_amd.EdgeSet(e.name, e.features.to(dev), e.senders.to(dev), e.receivers.to(dev)) for e in g.edge_sets])
N = graph.node_features[0].shape[0]
target = torch.randn(N, 3, device=dev); mask = torch.ones(N, dtype=torch.bool, device=dev)
torch.manual_seed(0)
model = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 15, 'none', ['mesh_edges']).to(dev)
with torch.no_grad():
    model(graph)
tr = parallel.DataParallelTrainer(model, lr=1e-4)
def fresh():
    return hgn_amd.MultiGraph(list(graph.node_features), [hgn_amd.EdgeSet(e.name, e.features, e.senders.clone(), e.receivers.clone()) for e in graph.edge_sets])
for _ in range(2):
    tr.step(graph, target, mask)
torch.cuda.synchronize()
for clear in (False, True, True):
    t0 = time.perf_counter()
    if clear:
        topology.clear_cache()
    tr.step(fresh(), target, mask)
    torch.cuda.synchronize()
    print('clear' if clear else 'warm', round((time.perf_counter() - t0) * 1e3, 1), 'ms', flush=True)
pr = cProfile.Profile()
topology.clear_cache()
pr.enable()
tr.step(fresh(), target, mask)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
