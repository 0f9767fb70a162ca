"""12-wave / 192-row-tile variants of the split-bf16 edge kernels against the 4-wave / 64-row ones on the same inputs: every output,
saved activation and gradient must be EQUAL (same per-row arithmetic), then the timing of both (library HIP-event profiler).
    python tools/bigcheck.py [--batch 128]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
from hgn_amd import ops, topology, synthetic, modules, _lib
import hgn_amd

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=128)
ap.add_argument('--iters', type=int, default=6)
a = ap.parse_args()
g = synthetic.batch([synthetic.grid_graph(seed=i % 4) for i in range(a.batch)])
es = g.edge_sets[0]
N, E = g.node_features[0].shape[0], es.senders.shape[0]
dev = torch.device('cuda')
topo = topology.EdgeTopology(es.senders, es.receivers, N, dev)
torch.manual_seed(0)
m = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges']).to(dev)
w = modules.weights_of(m.processor.graphnet_blocks[0].edge_models['mesh_edges'], 384)
h = torch.randn(N, 128, device=dev, requires_grad=True)
e = torch.randn(E, 128, device=dev, requires_grad=True)
L = _lib.lib()
params = [p for p in m.processor.graphnet_blocks[0].edge_models['mesh_edges'].parameters()]


def run(big):
    L.hgn_set_big_tiles(1 if big else 0)
    for t in [h, e] + params:
        t.grad = None
    y, agg = ops.edge_block(h, e, topo, w, ('sum',))
    saves = [t.clone() for t in y.grad_fn.saves]
    (y.square().sum() + agg.square().sum()).backward()
    torch.cuda.synchronize()
    return [y.detach().clone(), agg.detach().clone()] + saves + [h.grad.clone(), e.grad.clone()] + [p.grad.clone() for p in params]


ref, new = run(False), run(True)
names = ['out', 'agg', 'z1', 'z2', 'xhat', 'rstd', 'bits', 'dh', 'de'] + [f'dparam{i}' for i in range(len(params))]
ok = True
for nm, r, x in zip(names, ref, new):
    same = torch.equal(r, x)
    if not same:
        err = float((r.double() - x.double()).norm() / max(float(r.double().norm()), 1e-30))
        print(f'{nm:8s} DIFFERS rel {err:.3e}')
    ok &= same
print('BIG_CHECK', 'OK (all equal)' if ok else 'FAILED', 'rows', E)
for big in (False, True):
    L.hgn_set_big_tiles(1 if big else 0)
    for it in range(a.iters + 2):
        if it == 2:
            torch.cuda.synchronize(); ops.prof_reset(); ops.prof_enable(True)
        y, agg = ops.edge_block(h, e, topo, w, ('sum',))
        (y.sum() + agg.sum()).backward()
    torch.cuda.synchronize()
    k = ops.prof_collect(); ops.prof_enable(False)
    print('big ' if big else 'base', ' '.join(f"{n}={v['ms'] / v['count']:.4f}ms" for n, v in k.items()))
L.hgn_set_big_tiles(0)
sys.exit(0 if ok else 1)
