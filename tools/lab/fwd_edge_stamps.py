"""Where one workgroup's time goes in the training edge forward (laboratory build tools/lab/build_fwd_stamps.sh):
    HGN_LIB=$PWD/hyper-graph-nets_amd/hgn_amd/abl/libhgn_mp_fstamp.so python tools/lab/fwd_edge_stamps.py
Shader-clock stamps of the four waves of workgroup 5000 of the last forward launch (1 188 096 rows = 9 282 workgroups)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
from hgn_amd import ops, topology, synthetic, modules, _lib
import hgn_amd

NAMES = ['kernel entry (indices loaded)']
for b in (1, 2, 3):
    NAMES += [f'B{b} entry', f'B{b} loads issued', f'B{b} landed', f'B{b} barrier 0', f'B{b} split done', f'B{b} sweep 0', f'B{b} piece 1 landed',
              f'B{b} barrier 1', f'B{b} sweep 1', f'B{b} piece 2 landed', f'B{b} barrier 2', f'B{b} sweep 2', f'B{b} piece 3 landed', f'B{b} barrier 3',
              f'B{b} sweep 3']
    if b < 3:
        NAMES += [f'save z{b}: begin', f'save z{b}: stores issued + next piece 0 landed']
NAMES += ['block 3 done']

g = synthetic.batch([synthetic.grid_graph(seed=i % 4) for i in range(128)])
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
L.hgn_debug_edge_fwd_stamps.argtypes = [C.c_void_p]
buf = (C.c_uint64 * (4 * 96))()
for rep in range(3):
    for it in range(5):
        y, agg = ops.edge_block(h, e, topo, w, ('sum',))
    torch.cuda.synchronize()
    assert L.hgn_debug_edge_fwd_stamps(buf) == 0
    t0 = min(buf[wv * 96] for wv in range(4))
    print(f'== launch {rep}: cycles from the first wave\'s entry; waves 0..3 | step of wave 0')
    prev = None
    for i, name in enumerate(NAMES):
        v = [buf[wv * 96 + i] - t0 for wv in range(4)]
        step = '' if prev is None else f'(+{v[0] - prev:6d})'
        print(f'  {i:2d} {name:48s} ' + ' '.join(f'{x:7d}' for x in v) + '  ' + step)
        prev = v[0]
