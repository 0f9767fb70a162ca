"""Experiment driver for product mode 3 (two scaled fp16 terms, three MFMAs per product; csrc/mlp6_device.h: Prod<3>):
accuracy of an edge block + node MLP against fp64 beside the six-product bf16 mode, at several operand magnitudes, and the
kernels' times at the headline row count.   python tools/exp_f16x2.py [--batch 128]"""
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
ap.add_argument('--modes', default='fp32,fp32-f16x2')
a = ap.parse_args()
dev = torch.device('cuda')


def rel(x, y):
    return float((x.double() - y.double()).abs().max() / y.double().abs().max().clamp_min(1e-300))


def accuracy(nx, ny, h_scale, e_scale, w_scale, row_spread):
    g = synthetic.grid_graph(seed=3, nx=nx, ny=ny)
    es = g.edge_sets[0]
    N, E = g.node_features[0].shape[0], es.senders.shape[0]
    topo = topology.EdgeTopology(es.senders.cuda(), es.receivers.cuda(), N, dev)
    torch.manual_seed(0)
    m = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges']).cuda()
    with torch.no_grad():
        m(hgn_amd.MultiGraph([g.node_features[0].cuda()], [hgn_amd.EdgeSet(es.name, es.features.cuda(), es.senders.cuda(), es.receivers.cuda())]))
        for p in m.parameters():
            if p.dim() == 2:
                p.mul_(w_scale)
    blk = m.processor.graphnet_blocks[0]
    we = modules.weights_of(blk.edge_models['mesh_edges'], 384)
    wn = modules.weights_of(blk.node_model_cross, 256)
    gen = torch.Generator().manual_seed(1)
    h0 = (torch.randn(N, 128, generator=gen) * h_scale).cuda()
    rs = torch.exp(torch.randn(E, 1, generator=gen) * row_spread)          # rows of very different magnitude
    e0 = (torch.randn(E, 128, generator=gen) * e_scale * rs).cuda()
    e0[5] = 0                                                              # an all-zero row
    P = {n: p.detach().double() for n, p in blk.named_parameters()}
    snd, rcv = topo.snd.long(), topo.rcv.long()

    def mlp(x, pre):
        z = torch.relu(x @ P[pre + '.0.layers.linear_0.weight'].T + P[pre + '.0.layers.linear_0.bias'])
        z = torch.relu(z @ P[pre + '.0.layers.linear_1.weight'].T + P[pre + '.0.layers.linear_1.bias'])
        z = z @ P[pre + '.0.layers.linear_2.weight'].T + P[pre + '.0.layers.linear_2.bias']
        return torch.nn.functional.layer_norm(z, (128,), P[pre + '.1.weight'], P[pre + '.1.bias'], 1e-5)
    with torch.no_grad():
        h, e = h0.double(), e0.double()
        y64 = e + mlp(torch.cat([h[snd], h[rcv], e], 1), 'edge_models.mesh_edges')
        agg = torch.zeros(N, 128, dtype=torch.float64, device='cuda').index_add(0, rcv, y64)
        hn64 = h + mlp(torch.cat([h, agg], 1), 'node_model_cross')
    out = {}
    for mode in a.modes.split(','):
        with ops.using(ops.Context(precision=mode)), torch.no_grad():
            y, ag = ops.edge_block(h0, e0, topo, we, ('sum',))
            hn = ops.fused_mlp([h0, ag], wn, None, 0)
        # per-row error of the MLP part alone (y - e), relative to that row's own scale
        d = ((y.double() - e0.double()) - (y64 - e0.double())).abs().max(1).values
        out[mode] = (rel(y, y64), rel(hn, hn64), float(d.max()), bool(torch.isfinite(y).all()))
    print(f'E={E} h*{h_scale:g} e*{e_scale:g} W*{w_scale:g} spread {row_spread:g}: ' +
          '  '.join(f'{k}: y {v[0]:.2e} hn {v[1]:.2e} LN-part max abs {v[2]:.2e} finite {v[3]}' for k, v in out.items()), flush=True)


for cfg in [(20, 20, 1, 1, 1, 0), (20, 20, 1, 1, 1, 3), (20, 20, 1e-6, 1e-6, 1, 0), (20, 20, 1e3, 1e4, 1, 0), (20, 20, 1, 1, 30, 0),
            (20, 20, 1, 1, 1e-3, 0), (7, 5, 1, 1, 1, 0)]:
    accuracy(*cfg)

def grad_accuracy(nx, ny, gscale):
    """forward + backward of an edge block and a two-source node MLP against fp64 autograd; `gscale` scales the loss (gradient magnitude)"""
    g = synthetic.grid_graph(seed=3, nx=nx, ny=ny)
    es = g.edge_sets[0]
    N, E = g.node_features[0].shape[0], es.senders.shape[0]
    topo = topology.EdgeTopology(es.senders.cuda(), es.receivers.cuda(), N, dev)
    torch.manual_seed(0)
    m = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges']).cuda()
    with torch.no_grad():
        m(hgn_amd.MultiGraph([g.node_features[0].cuda()], [hgn_amd.EdgeSet(es.name, es.features.cuda(), es.senders.cuda(), es.receivers.cuda())]))
    blk = m.processor.graphnet_blocks[0]
    we = modules.weights_of(blk.edge_models['mesh_edges'], 384)
    wn = modules.weights_of(blk.node_model_cross, 256)
    h0 = torch.randn(N, 128, generator=torch.Generator().manual_seed(1)).cuda()
    e0 = torch.randn(E, 128, generator=torch.Generator().manual_seed(2)).cuda()
    rowsc = torch.exp(torch.randn(E, 1, generator=torch.Generator().manual_seed(3)) * 2).cuda()      # rows whose gradients differ by orders of magnitude

    def run():
        h = h0.clone().requires_grad_(True); e = e0.clone().requires_grad_(True)
        for p in list(blk.parameters()):
            p.grad = None
        y, agg = ops.edge_block(h, e, topo, we, ('sum',))
        hn = ops.fused_mlp([h, agg], wn, None, 0)
        ((hn.square().sum() + (y * rowsc).square().sum()) * gscale).backward()
        return [y.detach(), hn.detach(), h.grad, e.grad] + [p.grad.clone() for p in blk.parameters()]

    snd, rcv = topo.snd.long(), topo.rcv.long()
    h = h0.double().requires_grad_(True); e = e0.double().requires_grad_(True)
    ps = [p.detach().double().requires_grad_(True) for p in blk.parameters()]
    P = dict(zip([n for n, _ in blk.named_parameters()], ps))

    def mlp(x, pre):
        z = torch.relu(x @ P[pre + '.0.layers.linear_0.weight'].T + P[pre + '.0.layers.linear_0.bias'])
        z = torch.relu(z @ P[pre + '.0.layers.linear_1.weight'].T + P[pre + '.0.layers.linear_1.bias'])
        z = z @ P[pre + '.0.layers.linear_2.weight'].T + P[pre + '.0.layers.linear_2.bias']
        return torch.nn.functional.layer_norm(z, (128,), P[pre + '.1.weight'], P[pre + '.1.bias'], 1e-5)
    y = e + mlp(torch.cat([h[snd], h[rcv], e], 1), 'edge_models.mesh_edges')
    agg = torch.zeros(N, 128, dtype=torch.float64, device='cuda').index_add(0, rcv, y)
    hn = h + mlp(torch.cat([h, agg], 1), 'node_model_cross')
    ((hn.square().sum() + (y * rowsc.double()).square().sum()) * gscale).backward()
    r64 = [y.detach(), hn.detach(), h.grad, e.grad] + [p.grad for p in ps]
    for mode in a.modes.split(','):
        with ops.using(ops.Context(precision=mode)):
            got = run()
        errs = [rel(x, y_) for x, y_ in zip(got, r64)]
        print(f'grads E={E} loss*{gscale:g} {mode}: out {max(errs[:2]):.2e}  dh {errs[2]:.2e} de {errs[3]:.2e}  worst param grad {max(errs[4:]):.2e}  '
              f'finite {all(bool(torch.isfinite(x).all()) for x in got)}', flush=True)


for cfg in [(20, 20, 1.0), (20, 20, 1e-9), (20, 20, 1e6), (33, 29, 1e-4)]:
    grad_accuracy(*cfg)

# ---- timing at the headline row count ----------------------------------------------------------------------------------
g = synthetic.batch([synthetic.grid_graph(seed=i % 4) for i in range(a.batch)])
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
for rep in range(2):
    for mode in a.modes.split(','):
        with ops.using(ops.Context(precision=mode)):
            for it in range(a.iters + 2):
                if it == 2:
                    torch.cuda.synchronize(); ops.prof_reset(); ops.prof_enable(True)
                y, agg = ops.edge_block(h, e, topo, w, ('sum',))
                hn = ops.fused_mlp([h, agg], wn, None, 0)
                (y.sum() + hn.sum()).backward()
            torch.cuda.synchronize()
            k = ops.prof_collect()
            ops.prof_enable(False)
        print('rows', E, mode, ' '.join(f"{n}={v['ms'] / v['count']:.4f}ms" for n, v in k.items()), flush=True)
