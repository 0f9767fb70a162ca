"""Diagnostic: edge MLP with z1 == 1 and z2 == 2 for every row (W1 = 0, b1 = 1, W2 = 0, b2 = 2): dW2[j][:] must equal db2[j], dW3[j][:] = 2 db3[j]."""
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
shapes = O.param_shapes('none', 'sum', 1, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 128)
sd = O.init_state_dict_like(shapes, seed=2)
pre = 'processor.graphnet_blocks.0.edge_models.mesh_edges.0.layers.'
sd[pre + 'linear_0.weight'].zero_(); sd[pre + 'linear_0.bias'].fill_(1.0)
sd[pre + 'linear_1.weight'].zero_(); sd[pre + 'linear_1.bias'].fill_(2.0)
N = nx * ny
target = torch.randn(N, 3, generator=torch.Generator().manual_seed(1)); mask = torch.ones(N, dtype=torch.bool)
model = H.hip_model('none', 'sum', 1, ['mesh_edges'], sd)
ops.set_fused_edge_backward(True)
_, _, g, _ = H.hip_run(model, graph, target, mask)
ops.set_fused_edge_backward(None)
dW2, db2 = g[pre + 'linear_1.weight'].cpu(), g[pre + 'linear_1.bias'].cpu()
dW3, db3 = g[pre + 'linear_2.weight'].cpu(), g[pre + 'linear_2.bias'].cpu()
r2 = dW2 / db2.unsqueeze(1)
r3 = dW3 / db3.unsqueeze(1)
print('dW2 / db2: min', float(r2.min()), 'max', float(r2.max()), ' (expected 1)')
print('dW3 / db3: min', float(r3.min()), 'max', float(r3.max()), ' (expected 2)')
