"""Diagnostic: per-tensor gradient error of the HIP path and of the fp32 oracle against the fp64 oracle on the headline graph."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')]
import torch
from oracle import mgn_oracle as O
from tests import helpers as H, synth

agg = sys.argv[1] if len(sys.argv) > 1 else 'sum'
graph = synth.grid_graph(seed=0)
shapes = O.param_shapes('none', agg, 15, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 128)
N = 1600
target = torch.randn(N, 3, generator=torch.Generator().manual_seed(4))
mask = torch.ones(N, dtype=torch.bool); mask[:3] = False
for seed in (3, 4):
    sd = O.init_state_dict_like(shapes, seed=seed)
    out_o, loss_o, g64, _ = H.oracle_run(sd, graph, 'none', agg, target, mask)
    _, _, g32, _ = H.oracle_run(sd, graph, 'none', agg, target, mask, dtype=torch.float32)
    model = H.hip_model('none', agg, 15, ['mesh_edges'], sd)
    out, loss, g, _ = H.hip_run(model, graph, target, mask)
    rows = []
    for k in g64:
        e = g64[k].double()
        if float(e.abs().max()) == 0:
            continue
        a, b = g[k].double().cpu(), g32[k].double()
        sc = float(e.abs().max())
        rows.append((k, H.rel_err(a, e), H.rel_err(b, e), float((a - e).norm() / e.norm()), float((b - e).norm() / e.norm()),
                     float(((a - e).abs() > 5e-5 * sc).double().mean()), float(((b - e).abs() > 5e-5 * sc).double().mean())))
    rows.sort(key=lambda r: -r[1])
    print(f'seed {seed}: worst-norm ours {rows[0][1]:.2e} ref32 {max(r[2] for r in rows):.2e}; worst L2 ours {max(r[3] for r in rows):.2e} ref32 {max(r[4] for r in rows):.2e}')
    for r in rows[:12]:
        print('  %-60s norm %.2e / %.2e   l2 %.2e / %.2e   frac>5e-5 %.1e / %.1e' % r)
