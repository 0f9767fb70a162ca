import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
from oracle import mgn_oracle as O
from tests import helpers as H, synth
arch, agg, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
wseed = int(sys.argv[4]) if len(sys.argv) > 4 else 11
sets = ['mesh_edges', 'intra_cluster_to_mesh', 'intra_cluster_to_cluster', 'inter_cluster']
graph = synth.grid_graph(seed=5, nx=12, ny=8, clusters=5)
edge_in = {e.name: e.features.shape[1] for e in graph.edge_sets}
shapes = O.param_shapes(arch, agg, steps, sets, 5, edge_in, 8, 3, 128)
sd = O.init_state_dict_like(shapes, seed=wseed)
N = 96
target = torch.randn(N, 3, generator=torch.Generator().manual_seed(1))
mask = torch.ones(N, dtype=torch.bool); mask[:3] = False
out_o, loss_o, grads_o, ing_o = H.oracle_run(sd, graph, arch, agg, target, mask)
o32 = H.oracle_run(sd, graph, arch, agg, target, mask, dtype=torch.float32)
print('fp32 oracle vs fp64 oracle worst grad err', max(H.rel_err(o32[2][k], grads_o[k]) for k in grads_o if float(grads_o[k].abs().max())>0))
model = H.hip_model(arch, agg, steps, sets, sd)
for rep in range(2):
    out, loss, grads, ing = H.hip_run(model, graph, target, mask)
    print('rep', rep, 'out err', H.rel_err(out, out_o))
    for k in grads_o:
        if float(grads_o[k].abs().max()) > 0:
            e = H.rel_err(grads[k], grads_o[k])
            if e > 1e-5: print('  ', k, e)
