"""Data parallelism ON THE HIP KERNELS (BASELINE.json configs[3]): two ranks (gloo, sharing the one GPU
of the test box) run DataParallelTrainer / GraphedShardStep on the real model and must end up where single-process training on
the concatenated batch ends up -- the reference's semantics: one optimiser step on the mean over all NORMAL nodes of the whole
batch (/root/reference src/model/flag.py:146-154 inside src/algorithms/MeshSimulator.py:141-152).

Also: `python bench.py --gpus 2 --backend gloo` must start two ranks by itself and report n_gpus 2.
"""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import mgn_oracle as O
from tests import helpers as H
from tests import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

N_GRAPHS, NX, NY, LAYERS, LR, STEPS = 4, 7, 6, 2, 1e-3, 3


def _problem(arch='none', agg='sum'):
    graphs = [synth.grid_graph(seed=40 + i, nx=NX, ny=NY) for i in range(N_GRAPHS)]
    shapes = O.param_shapes(arch, agg, LAYERS, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 128)
    n = NX * NY
    gen = torch.Generator().manual_seed(7)
    targets = torch.randn(N_GRAPHS, n, 3, generator=gen)
    masks = torch.ones(N_GRAPHS, n, dtype=torch.bool)
    masks[:, :3] = False
    masks[0, :11] = False                                   # unequal NORMAL-node counts across ranks
    return graphs, shapes, targets, masks


def _to_dev(g):
    import hgn_amd
    return hgn_amd.MultiGraph([x.cuda() for x in g.node_features],
                              [hgn_amd.EdgeSet(e.name, e.features.cuda(), e.senders.cuda(), e.receivers.cuda()) for e in g.edge_sets])


def _dp_worker(rank, world, port, out, graphed):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from hgn_amd import parallel, graphs as hg
    gs, shapes, targets, masks = _problem()
    sd = O.init_state_dict_like(shapes, 11 + rank)          # different per rank: the trainer's broadcast must fix it
    model = H.hip_model('none', 'sum', LAYERS, ['mesh_edges'], sd)
    mine = parallel.shard_indices(N_GRAPHS, rank, world)
    g = _to_dev(synth.batch([gs[i] for i in mine]))
    target = torch.cat([targets[i] for i in mine]).cuda()
    mask = torch.cat([masks[i] for i in mine]).cuda()
    with torch.no_grad():
        model(g)
    tr = parallel.DataParallelTrainer(model, lr=LR, device_step=graphed)
    assert len(tr.bucket_starts) >= 1                        # the flat buffer goes out in >= 2 ranges (eager: overlapped with the backward pass)
    reduces = []
    orig = tr._reduce_from
    tr._reduce_from = lambda start, _o=orig: (reduces.append(start), _o(start))[1]
    step = hg.GraphedShardStep(tr, g, target, mask, warmup=1) if graphed else (lambda: tr.step(g, target, mask))
    losses, grad1 = [], None
    for i in range(STEPS):
        losses.append(step())
        if i == 0:
            grad1 = tr.fp.grad.clone()                       # gradient of the GLOBAL mean after the collective
    torch.cuda.synchronize()
    total = torch.stack(losses).cpu()
    dist.all_reduce(total)                                   # shares of the global mean add up to it
    flat0 = tr.fp.flat.clone().cpu()
    dist.broadcast(flat0, src=0)
    assert torch.equal(flat0, tr.fp.flat.cpu()), 'replicas diverged'
    if not graphed:                                          # every step: the boundary's range from inside the backward pass, then the rest
        assert reduces[-2:] == [tr.bucket_starts[-1], 0] or reduces[-len(tr.bucket_starts) - 1:] == list(reversed(tr.bucket_starts)) + [0], reduces
    if rank == 0:
        torch.save({'flat': tr.fp.flat.cpu(), 'grad1': grad1.cpu(), 'losses': total}, out)
    dist.destroy_process_group()


@pytest.mark.parametrize('graphed', [False, True], ids=['eager', 'hip-graph'])
def test_two_rank_data_parallel_on_hip_kernels_equals_single_process(tmp_path, graphed):
    from hgn_amd import parallel
    port = 29500 + (os.getpid() % 2000) + (7 if graphed else 0)
    out = str(tmp_path / 'dp.pt')
    mp.spawn(_dp_worker, args=(2, port, out, graphed), nprocs=2, join=True)
    res = torch.load(out)
    # single process, whole batch, same rank-0 weights: mean over ALL NORMAL nodes (flag.py:150-152), torch.optim-free path
    gs, shapes, targets, masks = _problem()
    sd = O.init_state_dict_like(shapes, 11)
    order = parallel.shard_indices(N_GRAPHS, 0, 2) + parallel.shard_indices(N_GRAPHS, 1, 2)
    g = _to_dev(synth.batch([gs[i] for i in order]))
    target = torch.cat([targets[i] for i in order]).cuda()
    mask = torch.cat([masks[i] for i in order]).cuda()
    single = parallel.DataParallelTrainer(H.hip_model('none', 'sum', LAYERS, ['mesh_edges'], sd), lr=LR)
    losses, grad1 = [], None
    for i in range(STEPS):
        losses.append(float(single.step(g, target, mask)))
        if i == 0:
            grad1 = single.fp.grad.clone().cpu()
    # and the fp64 oracle's gradient of the same global-mean loss on the concatenated batch
    ob = O.batch_graphs([O.MultiGraph(x.node_features, [O.EdgeSet(*e) for e in x.edge_sets]) for x in (gs[i] for i in order)])
    _, loss_o, grads_o, _ = H.oracle_run(sd, ob, 'none', 'sum', target.cpu(), mask.cpu())
    assert abs(float(res['losses'][0]) - float(loss_o)) <= 1e-5 * abs(float(loss_o))
    for (k, p), off in zip(single.model.named_parameters(), single.fp.offsets):
        if float(grads_o[k].abs().max()) == 0:
            continue
        got = res['grad1'][off:off + p.numel()].view(p.shape)
        assert H.rel_err(got, grads_o[k]) <= 2e-5, k                     # 2 ranks vs fp64 oracle
        assert H.rel_err(got, grad1[off:off + p.numel()].view(p.shape)) <= 1e-5, k     # 2 ranks vs 1 process
    for a, b in zip(res['losses'].tolist(), losses):
        assert abs(a - b) <= 5e-5 * abs(b), (a, b)
    # parameters after 3 Adam steps: on the scale of the updates they received (Adam turns rounding-level gradient entries
    # into +-lr moves): all but 1 % of the entries within 0.5 % of 3 * lr, hard cap 2 * 3 * lr
    d = (res['flat'] - single.fp.flat.cpu()).abs()
    assert int((d > 0.005 * STEPS * LR).sum()) <= max(2, d.numel() // 100)
    assert float(d.max()) <= 2 * STEPS * LR


def test_bench_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` outside a launcher starts the two ranks itself (children of a process that never touched the
    GPU) and rank 0 prints ONE JSON line with n_gpus 2 -- gloo here because the box has one GPU; the driver uses RCCL."""
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '3', '--warmup', '1',
                        '--batch', '4', '--layers', '3', '--no-prof'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['config']['parallelism'] == 'dp2' and res['config']['ranks'] == 2
    assert res['config']['global_batch'] == 8 and res['value'] > 0


@pytest.mark.parametrize('global_batch,per_rank', [(2, 1), (8, 4)])
def test_bench_gpus_2_strong_scaling_global_batch(global_batch, per_rank):
    """BASELINE.json configs[3] on two ranks: `bench.py --gpus 2 --global-batch B` splits a FIXED global batch over the ranks (one graph
    ... four graphs per rank; `scaling: strong`), the line reports the world size torch.distributed saw in its `collective` object, the
    flat gradient that was all-reduced (9.33 MB at 15 layers: SURVEY section 8e; here 3 layers) and whole-job edges/s.  gloo, because
    this box has one GPU; the driver's 8-GPU node runs the same code over RCCL."""
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '3', '--warmup', '1',
                        '--global-batch', str(global_batch), '--layers', '3', '--no-prof', '--no-cpu-baseline', '--no-cold', '--no-secondary'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['scaling'] == 'strong' and res['config']['parallelism'] == 'dp2'
    assert res['config']['global_batch'] == global_batch and res['config']['graphs_per_gpu'] == per_rank
    col = res['collective']
    assert col['ranks_reported_by_torch_distributed'] == 2 and col['backend'] == 'gloo'
    assert col['flat_gradient_bytes'] > 4 * 500_000 and sum(col['bucket_bytes']) == col['flat_gradient_bytes']
    assert res['value'] > 0 and abs(res['value'] - res['config']['edges_per_step'] / (res['ms_per_step'] * 1e-3)) <= 1e-6 * res['value']
