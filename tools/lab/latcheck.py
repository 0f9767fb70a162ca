"""Latency-form forward (mlp6_fwd_kernel<1, NP, 5>) against a torch fp64 statement of the same MLP, with sizes either side of LAT_MAX_TILES."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
import hgn_amd
from hgn_amd import ops, modules
torch.manual_seed(0)
dev = torch.device('cuda')
m = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges']).to(dev)
from hgn_amd import synthetic
g = synthetic.grid_graph(seed=1, nx=40, ny=40)
graph = hgn_amd.MultiGraph([x.to(dev) for x in g.node_features], [hgn_amd.EdgeSet(e.name, e.features.to(dev), e.senders.to(dev), e.receivers.to(dev)) for e in g.edge_sets])
with torch.no_grad():
    m(graph)
blk = m.processor.graphnet_blocks[0]
mlp = blk.node_model_cross
w = modules.weights_of(mlp, 256)
for M in (1, 31, 64, 333, 1600, 16384, 20000):
    x0 = torch.randn(M, 128, device=dev); x1 = torch.randn(M, 128, device=dev)
    with torch.no_grad():
        y = ops.fused_mlp([x0, x1], w, [None, None], 0)
    torch.cuda.synchronize()
    lin = [l for l in mlp.modules() if isinstance(l, torch.nn.Linear)]
    ln = [l for l in mlp.modules() if isinstance(l, torch.nn.LayerNorm)][0]
    h = torch.cat([x0, x1], 1).double()
    for i, l in enumerate(lin):
        h = h @ l.weight.double().t() + l.bias.double()
        if i < len(lin) - 1: h = torch.relu(h)
    h = torch.nn.functional.layer_norm(h, (128,), ln.weight.double(), ln.bias.double()) + x0.double()
    print(M, 'rel err', float((y.double() - h).norm() / h.norm()), 'y abs mean', float(y.abs().mean()), flush=True)
