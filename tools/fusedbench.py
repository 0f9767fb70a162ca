"""Times the edge-block kernels (forward, fused backward / two-launch backward) on a flag_simple-shape batch through the
library's own HIP-event profiler.  Diagnostic: HGN_FUSED_DBG ablation bits (csrc/fused_bwd.hip), HGN_NO_FUSED_BWD=1.
    python tools/fusedbench.py [--batch 128] [--iters 6]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
from hgn_amd import ops, topology, synthetic, modules
import hgn_amd

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=128)
ap.add_argument('--iters', type=int, default=6)
ap.add_argument('--reps', type=int, default=1)
ap.add_argument('--ab-edge-fwd', action='store_true')
a = ap.parse_args()
g = synthetic.batch([synthetic.grid_graph(seed=i % 4) for i in range(a.batch)])
es = g.edge_sets[0]
N, E = g.node_features[0].shape[0], es.senders.shape[0]
dev = torch.device('cuda')
topo = topology.EdgeTopology(es.senders, es.receivers, N, dev)
torch.manual_seed(0)
m = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges']).to(dev)
blk = m.processor.graphnet_blocks[0]
w = modules.weights_of(blk.edge_models['mesh_edges'], 384)
h = torch.randn(N, 128, device=dev, requires_grad=True)
e = torch.randn(E, 128, device=dev, requires_grad=True)
def run(tag):
    for it in range(a.iters + 2):
        if it == 2:
            torch.cuda.synchronize(); ops.prof_reset(); ops.prof_enable(True)
        y, agg = ops.edge_block(h, e, topo, w, ('sum',))
        (y.sum() + agg.sum()).backward()
    torch.cuda.synchronize()
    k = ops.prof_collect()
    ops.prof_enable(False)
    print('rows', E, tag, ' '.join(f"{n}={v['ms'] / v['count']:.4f}ms" for n, v in k.items()), flush=True)


for rep in range(a.reps):
    run('dbg ' + os.environ.get('HGN_FUSED_DBG', '0'))
    if a.ab_edge_fwd:                    # the general forward kernel on the same box, same process (csrc/mlp6.hip: edge_block_shape)
        with ops.using(ops.Context(general_fwd=True)):
            run('general-forward-kernel')
