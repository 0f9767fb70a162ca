"""A/B of the forward tiling at the headline row count (product mode 3): the training edge kernel (128-row tiles, 2 workgroups / CU),
the general kernel on 128-row tiles, the general kernel on 64-row tiles (HGN_F_TILE64_FWD: 3 workgroups / CU).  python tools/exp_tile64.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
from hgn_amd import ops, topology, synthetic, modules
import hgn_amd

dev = torch.device('cuda')
g = synthetic.batch([synthetic.grid_graph(seed=i % 4) for i in range(128)])
es = g.edge_sets[0]
N, E = g.node_features[0].shape[0], es.senders.shape[0]
topo = topology.EdgeTopology(es.senders, es.receivers, N, dev)
torch.manual_seed(0)
m = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges']).to(dev)
with torch.no_grad():
    m(hgn_amd.MultiGraph([g.node_features[0].to(dev)], [hgn_amd.EdgeSet(es.name, es.features.to(dev), es.senders.to(dev), es.receivers.to(dev))]))
blk = m.processor.graphnet_blocks[0]
w = modules.weights_of(blk.edge_models['mesh_edges'], 384)
wn = modules.weights_of(blk.node_model_cross, 256)
h = torch.randn(N, 128, device=dev, requires_grad=True)
e = torch.randn(E, 128, device=dev, requires_grad=True)
ref = None
for rep in range(2):
    for tag, ctx, env in (('edge-kernel', ops.Context(), None), ('general-128', ops.Context(general_fwd=True), None),
                          ('general-64', ops.Context(general_fwd=True), '1')):
        if env:
            os.environ['HGN_TILE64_FWD'] = env
        else:
            os.environ.pop('HGN_TILE64_FWD', None)
        with ops.using(ctx):
            for it in range(6):
                if it == 2:
                    torch.cuda.synchronize(); ops.prof_reset(); ops.prof_enable(True)
                y, agg = ops.edge_block(h, e, topo, w, ('sum',))
                hn = ops.fused_mlp([h, agg], wn, None, 0)
                (y.sum() + hn.sum()).backward()
            torch.cuda.synchronize()
            k = ops.prof_collect()
            ops.prof_enable(False)
        if ref is None:
            ref = y.detach().clone()
        print(tag, 'same bits as edge kernel:', bool(torch.equal(y.detach(), ref)),
              ' '.join(f"{n}={v['ms'] / v['count']:.4f}ms" for n, v in k.items() if n.startswith('mlp_fwd')), flush=True)
os.environ.pop('HGN_TILE64_FWD', None)
