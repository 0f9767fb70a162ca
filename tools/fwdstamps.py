"""Where one workgroup's time goes in the split-bf16 edge forward (diagnostic build `tools/build_ablations.sh 16`):
    HGN_LIB=$PWD/hyper-graph-nets_amd/hgn_amd/abl/libhgn_mp_abl16.so python tools/fwdstamps.py
Stamps (s_memrealtime, 10 ns) of wave 0 of workgroup 5000 of the last forward launch; see csrc/mlp6_device.h: gemm6."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch
from hgn_amd import ops, topology, synthetic, modules, _lib
import hgn_amd

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
L.hgn_debug_mlp6_stamps.argtypes = [C.c_void_p, C.c_void_p]
L.hgn_set_big_tiles.argtypes = [C.c_int]
buf = (C.c_uint64 * 256)(); n = C.c_int(0)
for big in (0, 1):
    L.hgn_set_big_tiles(big)
    print('== 12-wave / 192-row workgroups' if big else '== 4-wave / 64-row workgroups')
    for it in range(400):
        y, agg = ops.edge_block(h, e, topo, w, ('sum',))
        if it % 50 == 49 or it >= 398:
            torch.cuda.synchronize()
            L.hgn_debug_mlp6_stamps(buf, C.byref(n))      # reads and resets
    names = ['kernel entered'] + [f'block {b}: {s}' for b in range(3) for s in
             ('entered', 'stage free', 'DMA + loads issued', 'half 0 landed', 'split', 'products 0 issued', 'all waves done with half 0',
              'half 1 landed', 'products 1 issued')] + ['epilogue stores issued', 'segment sums done']
    t = [buf[i] for i in range(n.value)]
    cyc = [buf[128 + i] for i in range(n.value)]
    if n.value > 1:
        print(f'in-kernel clock over the workgroup: {(cyc[-1] - cyc[0]) / ((t[-1] - t[0]) * 0.01):.0f} MHz  ({cyc[-1] - cyc[0]} shader cycles in {(t[-1] - t[0]) / 100:.2f} us)')
    print('stamps', n.value)
    for i in range(1, len(t)):
        print(f'{(t[i] - t[0]) / 100:8.2f} us  (+{(t[i] - t[i - 1]) / 100:6.2f})  {names[i] if i < len(names) else i}')
