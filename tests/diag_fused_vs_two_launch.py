"""Fused edge backward vs the two-launch backward on one small graph: relative error of every gradient (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # (diagnostic script, run by hand: python tests/diag_...py [nx ny])
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
from oracle import mgn_oracle as O
from tests import helpers as H, synth
from hgn_amd import ops
nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7, 5)
graph = synth.grid_graph(seed=4, nx=nx, ny=ny)
shapes = O.param_shapes('none', 'sum', 2, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 128)
sd = O.init_state_dict_like(shapes, seed=2)
N = nx * ny
target = torch.randn(N, 3, generator=torch.Generator().manual_seed(1)); mask = torch.ones(N, dtype=torch.bool)
model = H.hip_model('none', 'sum', 2, ['mesh_edges'], sd)
res = {}
for fused in (True, False):
    ops.set_fused_edge_backward(fused)
    res[fused] = H.hip_run(model, graph, target, mask)
ops.set_fused_edge_backward(None)
gf, gu = res[True][2], res[False][2]
for k in gu:
    if float(gu[k].abs().max()) > 0:
        print(f'{H.rel_err(gf[k], gu[k]):.3e}  {k}')
print('node in-grad', H.rel_err(res[True][3]['node'][0], res[False][3]['node'][0]), 'edge in-grad', H.rel_err(res[True][3]['edge']['mesh_edges'], res[False][3]['edge']['mesh_edges']))
k = 'processor.graphnet_blocks.0.edge_models.mesh_edges.0.layers.linear_1.weight'
d = (gf[k] - gu[k]).abs().cpu()
ref = gu[k].abs().max().item()
print('dW2 err by row (first 16 of 128, relative):', [round(float(x) / ref, 3) for x in d.max(1).values[:16]])
print('rows with err > 1e-4:', int((d.max(1).values / ref > 1e-4).sum()), ' cols with err > 1e-4:', int((d.max(0).values / ref > 1e-4).sum()))
print('err by col block of 16:', [round(float(d[:, 16 * i:16 * i + 16].max()) / ref, 3) for i in range(8)])
print('err by row block of 16:', [round(float(d[16 * i:16 * i + 16].max()) / ref, 3) for i in range(8)])
r = (gf[k] / gu[k]).cpu()
print('ratio fused/ref median', float(r.median()), 'mean', float(r[gu[k].cpu().abs() > 0.1 * ref].mean()))
