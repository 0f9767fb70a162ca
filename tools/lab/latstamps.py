"""Timeline of one workgroup of the latency-form forward (mlp6_fwd_kernel<1, 6, 4 + loaders>) on ONE flag_simple-shape graph, the
rollout regime.  Build: HGN_ABL_EXTRA=-DHGN_STAMP_BLOCK=70 bash tools/build_ablations.sh 16 ; run with HGN_LIB=<abl16 library>."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
from hgn_amd import ops, topology, synthetic, modules, _lib
import hgn_amd
g = synthetic.grid_graph(seed=1, nx=40, ny=40)
es = g.edge_sets[0]
N, E = g.node_features[0].shape[0], es.senders.shape[0]
dev = torch.device('cuda')
topo = topology.EdgeTopology(es.senders, es.receivers, N, dev)
torch.manual_seed(0)
m = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges']).to(dev)
w = modules.weights_of(m.processor.graphnet_blocks[0].edge_models['mesh_edges'], 384)
h = torch.randn(N, 128, device=dev); e = torch.randn(E, 128, device=dev)
L = _lib.lib()
L.hgn_debug_mlp6_stamps.argtypes = [C.c_void_p, C.c_void_p]
buf = (C.c_uint64 * 256)(); n = C.c_int(0)
with torch.no_grad():
    for it in range(200):
        y, agg = ops.edge_block(h, e, topo, w, ('sum',))
        if it % 50 == 49 or it >= 198:
            torch.cuda.synchronize()
            L.hgn_debug_mlp6_stamps(buf, C.byref(n))      # reads and resets
names = ['kernel entered'] + [f'block {b} half {h}: {s}' for b in range(3) for h in range(2) for s in
         ('loads issued / previous sweep done', 'past the barrier', 'split / rotated', 'products issued')] + ['epilogue stores issued', 'segment sums done']
t = [buf[i] for i in range(n.value)]
cyc = [buf[128 + i] for i in range(n.value)]
print('stamps', n.value, f'clock {(cyc[-1] - cyc[0]) / ((t[-1] - t[0]) * 0.01):.0f} MHz' if n.value > 1 else '')
for i in range(1, len(t)):
    print(f'{(t[i] - t[0]) / 100:8.2f} us  (+{(t[i] - t[i - 1]) / 100:6.2f})  {names[i] if i < len(names) else i}')
