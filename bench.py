"""Headline benchmark: processed edges/sec (fwd+bwd) on flag_simple-shape meshes (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
        N>1 without WORLD_SIZE in the environment: this process starts `python -m torch.distributed.run --nproc-per-node N
        bench.py ...` as a CHILD (before anything touches the GPU) and exits with its code; under torch.distributed.run
        (WORLD_SIZE set) it is one rank of the job: one process per GPU, RCCL over xGMI.

A "step" is one full training step on this rank's batch of synthetic flag_simple-shape graphs: forward of the 15-layer
MeshGraphNet (architecture none, latent 128, sum aggregation), masked-MSE loss, backward, gradient all-reduce (N>1)
and the Adam update.  `value` = (graphs on all ranks x edges per graph) / step time, inputs resident in HBM.
Rank 0 prints ONE JSON line; it also carries `roofline` (dominant kernel, timed live with HIP events on the launch
stream) and, at N=1, `cpu_baseline` (the CPU oracle = port of the reference path, timed on this box's host cores on a
bounded sample: one graph, full depth).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_HBM_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E spec peak
PEAK_BF16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 matrix peak



PRODUCT_TEXT = {3: 'fp32 operands split into 2 fp16 terms scaled by powers of two, 3 fp16 MFMAs per product, fp32 accumulate',
                6: 'fp32 operands split into 3 bf16 terms, 6 bf16 MFMAs per product, fp32 accumulate',
                1: 'operands rounded to bf16, 1 bf16 MFMA per product', 2: 'forward: operands rounded to fp16, 1 fp16 MFMA per product; backward as mode 3'}


def limited_by(ent: dict, hbm_frac: float) -> str:
    """The bound label of the roofline object, DERIVED from the counters of profiles/sq_counters.json (rocprofv3 SQ_* passes of the
    same kernel sources) and the live HBM fraction: a roof binds above 70 % of it; otherwise the waves' own cycle split says whether
    they mostly wait parked (s_waitcnt / barriers: latency) or mostly stand at issue (dependencies, a busy pipe: issue)."""
    mp, parked = ent.get('matrix_pipe_busy_frac'), ent.get('waves_parked_frac')
    stalled, issuing = ent.get('waves_stalled_at_issue_frac'), ent.get('waves_issuing_frac')
    if mp is not None and mp >= 0.7:
        label = 'mfma'
    elif hbm_frac >= 0.7:
        label = 'hbm'
    elif parked is not None and stalled is not None and parked > stalled:
        label = 'latency (waves parked at waits / barriers)'
    elif stalled is not None:
        label = 'issue (waves stalled at issue)'
    else:
        label = 'unknown (counters incomplete)'
    return (f"{label}: matrix pipe {100 * (mp or 0):.0f} % busy, HBM {100 * hbm_frac:.0f} % of peak, waves issuing {100 * (issuing or 0):.0f} % / "
            f"stalled at issue {100 * (stalled or 0):.0f} % / parked {100 * (parked or 0):.0f} % of their cycles")


def log(*a):
    print('[bench]', *a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=128,
                    help='graphs per GPU (weak scaling: fixed per-GPU work); 128 = the saturating batch of SURVEY 8d')
    ap.add_argument('--layers', type=int, default=15)
    ap.add_argument('--agg', default='sum')
    ap.add_argument('--arch', default='none')
    ap.add_argument('--nx', type=int, default=40)
    ap.add_argument('--ny', type=int, default=40)
    ap.add_argument('--clusters', type=int, default=0, help='hyper nodes per graph (remote message passing edge sets)')
    ap.add_argument('--world-edges', type=int, default=0, help='extra world edges per graph (plate-style second edge set)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-cold', action='store_true', help='skip the cold-step (fresh index tensors) figures')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary configurations (pna; plate-shape hetero K=31)')
    ap.add_argument('--no-prof', action='store_true', help='do not record per-kernel HIP events in the timed region')
    ap.add_argument('--no-pack-plan', action='store_true', help='A/B: one pack launch per MLP and form instead of the one-launch table (ops.PackPlan)')
    ap.add_argument('--eager', action='store_true',
                    help='N=1: launch every kernel from the host in the timed region (per-kernel HIP events recorded live). '
                         'Default at N=1 is to replay the whole training step from one HIP graph, which keeps the measurement '
                         'independent of host-side launch jitter; per-kernel events are then taken in a short eager pass on the '
                         'same buffers right after the timed region (events cannot be timed inside a replayed graph).')
    ap.add_argument('--graph', action='store_true', help='(default at N=1; kept for compatibility)')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'fp32-bf16x3', 'fp32-f16x2', 'bf16', 'fp16'],
                    help="'bf16': one bf16 MFMA per product instead of the six fp32-accurate split products (reduced precision: NOT the "
                         "headline metric, outside the 1e-5 parity tolerance; BASELINE.json configs[4] asks for such an edge MLP)")
    ap.add_argument('--side-stream', action='store_true',
                    help='weight-gradient launches on a second stream (co-run with the next data-gradient kernels); paid off while those '
                         'kernels were MFMA bound, measured 1.7 %% slower since they are row-traffic bound')
    ap.add_argument('--no-side-stream', action='store_true', help='(default; kept for compatibility)')
    ap.add_argument('--workload', default='flag', choices=['flag', 'plate', 'cylinder'],
                    help="'flag' (headline): flag_simple-shape grids.  'plate': deforming_plate-shape frames (11x11x11 plate grid of 4-vertex "
                         "cells + an obstacle block) through PlateModel: radius-search world edges, spectral clustering into --clusters hyper "
                         "nodes, hetero connector (BASELINE.json configs[2] / plateCluster.yaml).  'cylinder': cylinder_flow-shape frames "
                         "(65x29 grid) through CylinderModel + hyper remote sets + a balance set (configs[4]; run with --precision fp16)")
    ap.add_argument('--global-batch', type=int, default=0,
                    help='strong-scaling mode: TOTAL graphs over all ranks (BASELINE.json configs[3]: 8 graphs over 8 GPUs = 1 per rank); '
                         'default 0 = weak scaling with --batch graphs per GPU')
    ap.add_argument('--buckets', type=int, default=4, help='N>1 with --eager: contiguous ranges the flat gradient buffer is all-reduced in, each launched as soon as the '
                    'backward pass has left its layers; with the captured step (default) the buffer goes out in one collective after the replay')
    ap.add_argument('--dp-rehearsal', action='store_true',
                    help='one rank, but through the N>1 code path: RCCL communicator of one rank, forward+backward replayed (or eager with '
                         'bucketed all-reduces), collective + scaling + Adam eager -- what a rank of the multi-GPU job runs besides waiting for its peers')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)')
    return ap.parse_args()


def kernel_source_sha() -> str:
    """sha256 over the kernel sources (csrc/*.hip, *.h, *.cpp, sorted): stamps PMC-derived figures so that a number measured on
    other kernel code is never reported as current (tools/make_traffic_json.py writes the same stamp)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'hyper-graph-nets_amd', 'csrc')
    for f in sorted(glob.glob(os.path.join(d, '*.hip')) + glob.glob(os.path.join(d, '*.h')) + glob.glob(os.path.join(d, '*.cpp'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def secondary_configs(args):
    """Driver-visible secondary figures, each a short run of this same script as a CHILD process (never exec: the parent holds the
    GPU), each on the shape its name says:
      * the YAML default aggregation (pna, flag.yaml:32) on the headline flag_simple-shape batch;
      * BASELINE.json configs[2]: deforming_plate-shape frames THROUGH PlateModel (4-vertex cells, radius-search world edges,
        spectral clustering K = 31, hetero connector, pna, 5 MP layers: plateCluster.yaml);
      * BASELINE.json configs[4] at one GPU: cylinder_flow-shape frames, hyper block, 25 MP layers, + balance set, fp16 forward
        products (`--precision fp16`: reduced precision, its own tolerance, tests/test_gpu_parity.py);
      * the plate configuration's EDGE-SET STRUCTURE on the flag grids (what round 2 reported as "plate config"): kept for
        continuity under a name that says what it is;
      * the headline model at the reference's own batch sizes: 1 graph and 21 graphs per step."""
    out = {}
    base = [sys.executable, os.path.abspath(__file__), '--steps', '5', '--warmup', '2', '--no-cold', '--no-cpu-baseline',
            '--no-secondary', '--batch', str(args.batch)]
    for name, extra in (('flag_simple_shape_pna_L15', ['--agg', 'pna']),
                        ('deforming_plate_shape_PlateModel_spectral_K31_hetero_pna_L5',
                         ['--workload', 'plate', '--arch', 'hetero', '--agg', 'pna', '--layers', '5', '--clusters', '31']),
                        ('cylinder_flow_shape_hyper_pna_L25_balance_fp16_products',
                         # (--no-prof: the per-kernel eager pass behind the captured step needs a second working set, which
                         # 25 layers of saved activations for 128 graphs do not leave room for next to the graph's pool)
                         ['--workload', 'cylinder', '--arch', 'hyper', '--agg', 'pna', '--layers', '25', '--clusters', '16',
                          '--precision', 'fp16', '--no-prof']),
                        ('flag_grid_40x40_with_plate_edge_set_structure_hetero_pna_L5_K31',
                         ['--arch', 'hetero', '--agg', 'pna', '--layers', '5', '--clusters', '31', '--world-edges', '300']),
                        # the reference's own batch sizes (configs/flag.yaml:10 batch_size; one trajectory step at a time): launch-latency regime
                        ('flag_simple_shape_1_graph_per_step', ['--batch', '1', '--steps', '40', '--warmup', '5', '--no-prof']),
                        ('flag_simple_shape_21_graphs_per_step_reference_batch_size', ['--batch', '21', '--steps', '30', '--warmup', '5', '--no-prof'])):
        try:
            r = subprocess.run(base + extra, capture_output=True, text=True, timeout=420)
            line = [l for l in r.stdout.splitlines() if l.startswith('{')]
            if r.returncode == 0 and line:
                d = json.loads(line[-1])
                out[name] = {'edges_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'edges_per_step': d['config']['edges_per_step'],
                             'graphs_per_gpu': d['config']['graphs_per_gpu'], 'steps': d['steps'], 'dtype': d['dtype'],
                             'workload': d['config']['workload']}
                if 'graph_build' in d['config']:
                    out[name]['graph_build'] = d['config']['graph_build']
                if 'roofline' in d:            # the child's own dominant kernel (HIP events of its eager pass): bytes, time, fraction
                    r_ = d['roofline']
                    out[name]['roofline'] = {k_: r_.get(k_) for k_ in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac',
                                                                         'algorithmic_bytes_per_launch', 'rows_per_launch', 'ms_per_launch')}
                    out[name]['roofline']['share_of_step'] = d.get('kernels', {}).get(r_.get('kernel'), {}).get('share_of_step')
                if 'roofline_aggregation' in d:
                    out[name]['roofline_aggregation'] = {k_: d['roofline_aggregation'].get(k_) for k_ in ('kernel', 'achieved', 'frac', 'ms_per_launch')}
            else:
                out[name] = {'error': (r.stderr or '')[-300:]}
        except Exception as ex:                             # a secondary figure must never cost the headline line
            out[name] = {'error': f'{type(ex).__name__}: {ex}'}
        log(f'secondary {name}: {out[name]}')
    return out


def host_cores() -> int:
    """CPU threads this process may really use: min(affinity, cgroup quota, 16 = one GPU's share of the box)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, min(n, 16))


def cpu_baseline(args, graph1, state_dict):
    """The oracle (CPU port of the reference op sequence) on ONE graph of the workload at full depth, fwd+loss+bwd, with the
    benchmarked model's own parameters."""
    from oracle import mgn_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in state_dict.items()}
    N = graph1.node_features[0].shape[0]
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(0))
    mask = torch.ones(N, dtype=torch.bool); mask[:3] = False
    g = O.MultiGraph(list(graph1.node_features), [O.EdgeSet(*e) for e in graph1.edge_sets])
    E = sum(e.senders.shape[0] for e in graph1.edge_sets)

    def one():
        for v in sd.values():
            v.grad = None
        out = O.mesh_graph_net(sd, g, args.arch, args.agg)
        O.masked_mse(out, target, mask).backward()
    log(f'cpu baseline: {cores} threads')
    one()
    iters, t0 = 0, time.perf_counter()
    while iters < 2 or (time.perf_counter() - t0 < 12.0 and iters < 10):
        one(); iters += 1
        log(f'cpu baseline iter {iters}: {time.perf_counter() - t0:.1f} s')
    dt = (time.perf_counter() - t0) / iters
    return {'value': E / dt, 'unit': 'edges/s', 'cores': cores, 'kind': 'port',
            'sample': f'1 graph of the workload ({N} nodes, {E} edges), L={args.layers}, {args.agg}, fwd+loss+bwd, {iters} iters, '
                      f'{dt:.3f} s/iter, torch {torch.__version__} CPU fp32'}


def cpu_baseline_forward(state_dict, graph1, arch, agg, iters=5):
    """cpu_baseline leg for the rollout regime (tools/rolloutbench.py): the oracle's forward of ONE graph on the host cores.
    -> (ms per forward, threads, output tensor)"""
    from oracle import mgn_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu() for k, v in state_dict.items()}
    og = O.MultiGraph(list(graph1.node_features), [O.EdgeSet(*e) for e in graph1.edge_sets])
    with torch.no_grad():
        out = O.mesh_graph_net(sd, og, arch, agg)
        t0 = time.perf_counter()
        for _ in range(iters):
            O.mesh_graph_net(sd, og, arch, agg)
    return (time.perf_counter() - t0) / iters * 1e3, cores, out


def cpu_baseline_features(frames, iters=10):
    """cpu_baseline leg for the frame -> graph rows (tools/featbench.py): the oracle's FlagModel.build_graph and
    build_graph + hierarchical connect (16 clusters of 100 nodes) per frame on the host cores."""
    from oracle import features_oracle as FO
    cores = host_cores()
    torch.set_num_threads(cores)
    ff = FO.FlagFeatures()
    n_nodes = frames[0]['world_pos'].shape[0]
    lab = [i // 100 for i in range(n_nodes)]
    clusters = [torch.tensor([i for i in range(n_nodes) if lab[i] == k]) for k in range(max(lab) + 1)]
    g = ff.build_graph(frames[0], True)
    nb = FO.neighboring_clusters(g['edge_sets'][0].senders, g['edge_sets'][0].receivers, lab)
    t0 = time.perf_counter()
    for i in range(iters):
        g = ff.build_graph(frames[i % len(frames)], True)
    t1 = time.perf_counter()
    for i in range(iters):
        g = ff.build_graph(frames[i % len(frames)], True)
        FO.hierarchical_connect(g, clusters, nb, ff.intra_edge, ff.inter_edge, ff.hyper_node, True)
    t2 = time.perf_counter()
    return {'cores': cores, 'build_graph ms/frame': (t1 - t0) / iters * 1e3, 'build_graph+connect ms/frame': (t2 - t1) / iters * 1e3}



def _params(arch, agg, layers, clustering, K):
    """The `model` section of a reference YAML (configs/plateCluster.yaml:24-60 key set) for the system models."""
    remote = arch not in ('none',)
    return {'size': 3, 'aggregation': agg, 'message_passing_steps': layers,
            'rmp': {'clustering': clustering if remote else 'none', 'connector': arch if remote else 'none', 'num_clusters': K,
                    'hyper_noise': 'none', 'hyper_node_features': True, 'frequency': 1, 'fully_connect': False,
                    'intra_cluster_sampling': {'enabled': False, 'alpha': 0.1, 'spotter_threshold': 0}},
            'graph_balancer': {'algorithm': 'none', 'frequency': 1}}


def build_flag(args, gids, dev):
    import hgn_amd
    from hgn_amd import synthetic
    graphs = [synthetic.grid_graph(seed=1000 + g, nx=args.nx, ny=args.ny, clusters=args.clusters, world=args.world_edges)
              for g in gids[:min(len(gids), 4)]]
    while len(graphs) < len(gids):                       # reuse topologies, fresh features (host generation is slow)
        src = graphs[len(graphs) % 4]
        gen = torch.Generator().manual_seed(2000 + gids[len(graphs)])
        graphs.append(synthetic.MultiGraph([torch.randn(x.shape, generator=gen) for x in src.node_features],
                                           [synthetic.EdgeSet(e.name, torch.randn(e.features.shape, generator=gen), e.senders,
                                                              e.receivers) for e in src.edge_sets]))
    big = synthetic.batch(graphs)
    graph = hgn_amd.MultiGraph([x.to(dev) for x in big.node_features],
                               [hgn_amd.EdgeSet(e.name, e.features.to(dev), e.senders.to(dev), e.receivers.to(dev))
                                for e in big.edge_sets])
    N_nodes = graph.node_features[0].shape[0]
    per = N_nodes // len(gids)
    target = torch.randn(N_nodes, 3, generator=torch.Generator().manual_seed(gids[0])).to(dev)
    node_type = torch.zeros(N_nodes, dtype=torch.bool)
    for i in range(len(gids)):
        node_type[i * per:i * per + 3] = True            # 3 HANDLE nodes per graph are masked out of the loss
    E_graph = sum(e.senders.shape[0] for e in graph.edge_sets) // len(gids)
    extra_sets = (f', {args.clusters} hyper nodes per graph' if args.clusters else '') + \
        (f', {2 * args.world_edges} synthetic world edges per graph' if args.world_edges else '')
    wl = (f'flag_simple-shape MeshGraphNets baseline: architecture {args.arch}, {args.layers} MP layers, latent 128, aggregation '
          f'{args.agg}, 1xMI355X config; {args.nx}x{args.ny} triangulated grid per graph ({per} nodes, {E_graph} directed edges'
          f'{extra_sets}); full training step fwd+loss+bwd+allreduce+Adam')
    return {'graph': graph, 'target': target, 'mask': (~node_type).to(dev), 'workload': wl, 'nodes_per_graph': per,
            'edges_per_graph': E_graph, 'cpu_graph': lambda: synthetic.grid_graph(seed=1000, nx=args.nx, ny=args.ny, clusters=args.clusters,
                                                                                 world=args.world_edges)}


def build_plate(args, gids, dev):
    """deforming_plate shape through the system model itself: per frame PlateModel.build_graph (4-vertex cells ->
    two-way mesh edges, world edges by the 0.03 radius search, plate.py:69-200) and expand_graph (obstacle removal, spectral
    clustering once per trajectory, hetero connector: plate.py:202-216); the frames of this rank are batched into one disjoint
    union (MeshSimulator.py:159-234 semantics, correct hyper ids).  All frames share one mesh (one trajectory), as in the
    reference's batches."""
    import random
    import numpy as np
    from hgn_amd import synthetic, system_model, batching
    K = args.clusters or 31
    random.seed(0); np.random.seed(0)
    pm = system_model.PlateModel(_params(args.arch, args.agg, args.layers, 'spectral', K))
    frames = [synthetic.plate_frame(seed=1000 + g, nx=11, ny=11, nz=11, obstacle=(6, 6, 3)) for g in gids]
    cells = frames[0]['cells'].to(dev)
    graphs, targets, masks = [], [], []
    t_build = t_expand = 0.0
    t_first_expand = None
    for i, fr in enumerate(frames):
        cf = {k: v.to(dev) for k, v in fr.items()}
        cf['cells'] = cells                               # one mesh per trajectory: the topology cache is hit
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g = pm.build_graph(cf, True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        mg = pm.expand_graph(g, i, len(frames), True)     # step 0 clusters (host, scikit-learn), later frames reuse the clusters
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if i == 0:
            t_first_expand = t2 - t1
        else:
            t_build += t1 - t0; t_expand += t2 - t1
        graphs.append(mg)
        targets.append(pm.get_target(cf, True))
        masks.append(cf['node_type'][:, 0] == 0)
    big = batching.batch_graphs(graphs)
    n = max(1, len(frames) - 1)
    per = frames[0]['world_pos'].shape[0]
    counts = {e.name: e.senders.shape[0] // len(gids) for e in big.edge_sets}
    E_graph = sum(e.senders.shape[0] for e in big.edge_sets) // len(gids)
    wl = (f'deforming_plate-shape HyperGraphNets (BASELINE.json configs[2], plateCluster.yaml): PlateModel frames of an 11x11x11 plate '
          f'grid of 4-vertex cells + a 6x6x3 obstacle block ({per} nodes), mesh edges from the deform rule, world edges by the 0.03 '
          f'radius search, spectral clustering K={K}, connector {args.arch}, {args.layers} MP layers, latent 128, aggregation {args.agg}; '
          f'per graph on average {counts}; full training step fwd+loss+bwd+allreduce+Adam on the batched frames')
    def cpu_graph():
        g1 = graphs[0]
        return synthetic.MultiGraph([x.detach().cpu() for x in g1.node_features],
                                    [synthetic.EdgeSet(e.name, e.features.detach().cpu(), e.senders.cpu(), e.receivers.cpu()) for e in g1.edge_sets])
    return {'graph': big, 'target': torch.cat(targets), 'mask': torch.cat(masks), 'workload': wl, 'nodes_per_graph': per,
            'edges_per_graph': E_graph, 'cpu_graph': cpu_graph, 'model': pm.learned_model,
            'graph_build': {'build_graph_ms_per_frame': t_build / n * 1e3, 'expand_graph_ms_per_frame': t_expand / n * 1e3,
                            'first_expand_graph_incl_clustering_ms': t_first_expand * 1e3,
                            'note': 'outside the timed step (the metric is fwd+bwd): PlateModel.build_graph = cells->edges (cached per mesh) + '
                                    'radius search + features + normalisers; expand_graph = hetero remote sets on the cached clusters; the '
                                    'first expand_graph includes obstacle removal and spectral clustering on the host (once per trajectory)'}}


def build_cylinder(args, gids, dev):
    """cylinder_flow shape (BASELINE.json configs[4]): CylinderModel.build_graph frames (65x29 grid: 1885 nodes, 3-dim mesh-edge
    features, cylinder.py:65-106) + a balance edge set + the hyper connector's remote sets on K strips (the reference cannot run
    this combination, SURVEY.md section 9-6: the sets are built by synthetic.cylinder_remote_sets)."""
    import hgn_amd
    from hgn_amd import synthetic, system_model
    K = args.clusters or 16
    cm = system_model.CylinderModel(_params('none', args.agg, 1, 'none', K))
    graphs = []
    for g in gids[:min(len(gids), 4)]:
        fr = synthetic.cylinder_frame(seed=1000 + g, nx=65, ny=29)
        mg = cm.build_graph({k: v.to(dev) for k, v in fr.items()}, True)
        me = mg.edge_sets[0]
        mesh = synthetic.EdgeSet('mesh_edges', me.features.detach().cpu(), me.senders.cpu(), me.receivers.cpu())
        graphs.append(synthetic.cylinder_remote_sets(mg.node_features[0].detach().cpu(), fr['mesh_pos'], mesh, K, 100, 1000 + g))
    while len(graphs) < len(gids):
        src = graphs[len(graphs) % 4]
        gen = torch.Generator().manual_seed(2000 + gids[len(graphs)])
        graphs.append(synthetic.MultiGraph([torch.randn(x.shape, generator=gen) for x in src.node_features],
                                           [synthetic.EdgeSet(e.name, torch.randn(e.features.shape, generator=gen), e.senders,
                                                              e.receivers) for e in src.edge_sets]))
    big = synthetic.batch(graphs)
    graph = hgn_amd.MultiGraph([x.to(dev) for x in big.node_features],
                               [hgn_amd.EdgeSet(e.name, e.features.to(dev), e.senders.to(dev), e.receivers.to(dev))
                                for e in big.edge_sets])
    N_nodes = graph.node_features[0].shape[0]
    per = N_nodes // len(gids)
    target = torch.randn(N_nodes, 3, generator=torch.Generator().manual_seed(gids[0])).to(dev)
    mask = torch.ones(N_nodes, dtype=torch.bool, device=dev)
    E_graph = sum(e.senders.shape[0] for e in graph.edge_sets) // len(gids)
    wl = (f'cylinder_flow-shape HyperGraphNets (BASELINE.json configs[4]): CylinderModel frames of a 65x29 triangulated grid ({per} nodes, '
          f'3-dim mesh-edge features) + balance set (200 directed edges) + hyper connector remote sets on K={K} strips, architecture '
          f'{args.arch}, {args.layers} MP layers, latent 128, aggregation {args.agg}, products {args.precision} ({E_graph} directed edges per '
          f'graph over all sets); full training step fwd+loss+bwd+allreduce+Adam')
    return {'graph': graph, 'target': target, 'mask': mask, 'workload': wl, 'nodes_per_graph': per, 'edges_per_graph': E_graph,
            'cpu_graph': lambda: graphs[0]}


def spawn_ranks(args) -> int:
    """--gpus N > 1 outside a launcher: run the N ranks as children of this (GPU-untouched) process.  Rank 0 of the job prints
    the JSON line on the shared stdout; a failed rank makes the launcher, and therefore this process, exit non-zero."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f'--gpus {args.gpus}: starting {args.gpus} ranks: {" ".join(cmd)}')
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))                       # nothing above this line initialises the GPU
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        log(f'note: --gpus {args.gpus} but the launcher started {world} rank(s); reporting the {world} that run')
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    local = local % torch.cuda.device_count()          # several ranks may share a GPU only in a gloo rehearsal
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dp = world > 1 or args.dp_rehearsal
    if dp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1 and 'MASTER_PORT' not in os.environ:
            with socket.socket() as s:
                s.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(s.getsockname()[1])
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import hgn_amd
    from hgn_amd import ops, synthetic, parallel, _lib
    _lib.lib()
    ops.set_matmul_precision(args.precision)

    # ---- this rank's shard of the global batch: graphs {g : g mod world == rank}, each with its own seed ----------
    strong = args.global_batch > 0
    total_graphs = args.global_batch if strong else args.batch * world
    if strong and total_graphs % world:
        raise SystemExit(f'--global-batch {total_graphs} is not a multiple of the {world} ranks')
    B = total_graphs // world
    gids = parallel.shard_indices(total_graphs, rank, world)
    wk = {'flag': build_flag, 'plate': build_plate, 'cylinder': build_cylinder}[args.workload](args, gids, dev)
    graph, target, mask = wk['graph'], wk['target'], wk['mask']
    N_nodes = graph.node_features[0].shape[0]
    E_rank = sum(e.senders.shape[0] for e in graph.edge_sets)
    E_graph, per = wk['edges_per_graph'], wk['nodes_per_graph']

    torch.manual_seed(0)
    sets = [e.name for e in graph.edge_sets]
    model = wk.get('model')                              # plate: the system model's own learned_model (built by get_model's class)
    if model is None:
        model = hgn_amd.MeshGraphNet(output_size=3, latent_size=128, num_layers=2, message_passing_aggregator=args.agg,
                                     message_passing_steps=args.layers, architecture=args.arch, edge_sets=sets).to(dev)
    log(f'rank {rank}: batch built: {N_nodes} nodes, {E_rank} edges')
    with torch.no_grad():
        model(graph)                                     # materialise lazy layers, build + cache the CSR topology
    torch.cuda.synchronize()
    log('first forward done')
    use_graph = not args.eager
    if args.no_pack_plan:
        ops.begin_step_packs = lambda c: None
    trainer = parallel.DataParallelTrainer(model, lr=1e-4, device_step=use_graph, wgrad_stream=args.side_stream, buckets=args.buckets,
                                           force_collectives=args.dp_rehearsal)
    n_params = trainer.fp.numel

    def barrier():
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    step = lambda: trainer.step(graph, target, mask)
    for i in range(args.warmup):
        step()
        if i == 0:
            torch.cuda.synchronize(); log('first training step done')
    if use_graph:
        from hgn_amd import graphs
        try:
            # N=1: the whole step (incl. Adam) is one graph; N>1: forward+backward is the graph, all-reduce and Adam eager
            gstep = (graphs.GraphedShardStep if dp else graphs.GraphedTrainStep)(trainer, graph, target, mask, warmup=1)
            gstep()
            step = lambda: gstep()
        except Exception as ex:                          # capture refused: measure the eager launches instead
            log(f'HIP-graph capture failed ({type(ex).__name__}: {ex}); falling back to eager launches')
            torch.cuda.synchronize()
            use_graph = False
    barrier()
    log('warmup done')
    prof = not args.no_prof
    if prof and not use_graph:
        ops.prof_reset(); ops.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    log(f'timed region done: {dt:.3f} s for {args.steps} steps')
    if prof and use_graph:            # events cannot be recorded inside a replayed graph: short eager pass on the same buffers
        barrier()
        ops.prof_reset(); ops.prof_enable(True)
        for _ in range(2):
            trainer.step(graph, target, mask)
        barrier()
    if prof:
        ops.prof_enable(False)
    # ---- "cold" steps: what the reference's real loop hands over -- FRESH index tensors every step (MeshSimulator.py:136,
    # 159-234 re-batch each trajectory), same mesh.  (i) topology found again by content fingerprint, captured step replayed
    # (graphs.GraphedStepCache); (ii) nothing cached: both radix sorts and their read-backs every step, eager launches.
    cold = None
    if not dp and not args.no_cold:
        from hgn_amd import graphs as hg, topology as topo_mod

        def fresh():
            return hgn_amd.MultiGraph(list(graph.node_features),
                                      [hgn_amd.EdgeSet(e.name, e.features, e.senders.clone(), e.receivers.clone()) for e in graph.edge_sets])
        cache = hg.GraphedStepCache(trainer)
        cache.step(fresh(), target, mask)                      # first sight: fingerprint hit (built above), capture
        torch.cuda.synchronize()
        n_cold = max(3, min(10, args.steps))
        t0 = time.perf_counter()
        for _ in range(n_cold):
            cache.step(fresh(), target, mask)
        torch.cuda.synchronize()
        t_hit = (time.perf_counter() - t0) / n_cold * 1e3
        # (capturing a HIP graph empties the allocator's cache -- torch.cuda.graph.__enter__ --, so the first eager step after the
        # capture above pays for ~45 GB of device mallocs: 0.1 s on some boxes, 2 s on others; that step is timed on its own
        # (first_eager_step_after_capture_ms) and kept out of the figure that is about the topology cache)
        topo_mod.clear_cache()
        t0 = time.perf_counter()
        trainer.step(fresh(), target, mask)
        torch.cuda.synchronize()
        t_first_eager = (time.perf_counter() - t0) * 1e3
        ms0 = torch.cuda.memory_stats()
        t0 = time.perf_counter()
        for _ in range(3):
            topo_mod.clear_cache()
            trainer.step(fresh(), target, mask)
        torch.cuda.synchronize()
        t_miss = (time.perf_counter() - t0) / 3 * 1e3
        ms1 = torch.cuda.memory_stats()
        log('cold eager steps: reserved %.1f -> %.1f GB, device mallocs %d, allocator retries %d' % (
            ms0['reserved_bytes.all.current'] / 2**30, ms1['reserved_bytes.all.current'] / 2**30,
            ms1['segment.all.allocated'] - ms0['segment.all.allocated'], ms1['num_alloc_retries'] - ms0['num_alloc_retries']))
        cold = {'fresh_index_tensors_topology_found_by_content_ms': t_hit, 'captures': cache.captures,
                'fresh_index_tensors_topology_rebuilt_eager_ms': t_miss,
                # a real once-per-(mesh, batch size) cost of the reference's loop, reported instead of warmed away: capturing a HIP graph
                # empties the caching allocator, so the first eager step behind a capture re-mallocs its working set from the driver
                'first_eager_step_after_capture_ms': t_first_eager,
                'note': 'per step, 128-graph batch; replayed step with warm tensors = ms_per_step above',
                'topology_cache': dict(topo_mod.stats)}
        log(f'cold steps: content hit {t_hit:.2f} ms, rebuild {t_miss:.2f} ms')
    collective = None
    if dp:                    # the collective by itself, outside the timed region: the whole flat gradient buffer, then its ranges
        buf = torch.zeros_like(trainer.fp.grad_ext)
        def timed(fn, n=10):
            fn(); barrier()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e6
        bounds = [0] + list(trainer.bucket_starts) + [buf.numel()]
        def ranges():
            ws = [dist.all_reduce(buf[lo:hi], async_op=True) for lo, hi in zip(bounds[:-1], bounds[1:])]
            for w in ws:
                w.wait()
        collective = {'ranks_reported_by_torch_distributed': dist.get_world_size(), 'backend': dist.get_backend(),
                      'flat_gradient_bytes': buf.numel() * 4, 'buckets': len(bounds) - 1,
                      'bucket_bytes': [(hi - lo) * 4 for lo, hi in zip(bounds[:-1], bounds[1:])],
                      'all_reduce_whole_buffer_us': timed(lambda: dist.all_reduce(buf)),
                      'all_reduce_in_ranges_us': timed(ranges),
                      'overlapped_with_backward': bool(trainer.overlap),
                      'note': 'measured after the timed region, nothing else running; in the step the ranges are launched from inside the backward '
                              'pass (eager) or back to back behind the replayed graph (default)'}
        log(f'collective: {collective}')
    tmax = torch.tensor([dt], device=dev)
    if dp:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    ms_per_step = dt / args.steps * 1e3
    value = E_rank * world * args.steps / dt

    if rank == 0:
        res = {'metric': 'processed edges/sec (fwd+bwd) on ' + {'flag': 'flag_simple mesh', 'plate': 'deforming_plate-shape graphs', 'cylinder': 'cylinder_flow-shape graphs'}[args.workload], 'value': value, 'unit': 'edges/s',
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
               'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
               'dtype': 'f32' if args.precision.startswith('fp32') else args.precision + ' (reduced precision run, not the headline metric)', 'data': 'synthetic',
               'config': {'workload': wk['workload'],
                          'graphs_per_gpu': B, 'global_batch': total_graphs, 'edges_per_step': E_rank * world,
                          'params': n_params, 'parallelism': f'dp{world}' + (' (one rank through the N>1 code path)' if args.dp_rehearsal and world == 1 else ''), 'loss': float(loss), 'hip_graph': bool(use_graph),
                          'ranks': dist.get_world_size() if dp else 1,
                          'collective_backend': (dist.get_backend() if dp else None),
                          'gpus_visible_per_rank': torch.cuda.device_count()}}
        if prof:
            k = ops.prof_collect()
            log('profile collected')
            psteps = 2 if use_graph else args.steps
            kernels = {n: {'ms_per_launch': v['ms'] / v['count'], 'launches_per_step': v['count'] / psteps,
                           'share_of_step': v['ms'] / psteps / ms_per_step} for n, v in k.items()}
            res['kernels'] = kernels
            # ---- roofline of the dominant kernel (largest accumulated time of the step) --------------------------------------
            # With split-bf16 products the fused kernels are bounded by their ROW TRAFFIC, not by the matrix pipe: the line is
            # priced against HBM (algorithmic bytes / launch time / 8 TB/s); the matrix-pipe figures ride along.  When the
            # weight gradients run on a side stream, backward-pass event times include the co-runner and only forward
            # kernels are eligible.
            overlapped = trainer.side is not None
            fwd_only = ('mlp_fwd_edge', 'mlp_fwd', 'linear_fwd')
            # algorithmic HBM bytes: per row of the launch, plus node-level arrays touched once per launch
            #   edge forward : read e, write e' + z1 + z2 + x-hat (5 x 512), ReLU sign words 32, rstd 4;  P rows (N x 1024)
            #                  gathered, fused `sum` aggregate (N x 512) written
            #   edge backward: read d(e') + x-hat (2 x 512), sign words 32, rstd 4, write dz3 + dz2 + dz1 + de (4 x 512);
            #                  d(agg) rows (N x 512) gathered, receiver sums of dz1 (N x 512) written
            #   weight grads : two operand rows (2 x 512) per task and row
            #   fused backward (default where eligible): read d(e') + x-hat + z2 + z1 (4 x 512), sign words, rstd, write dz1 + de
            #                  (2 x 512); d(agg) rows (N x 512) gathered.  It does the work of the edge backward AND of the dW3 / dW2
            #                  weight-gradient tasks (2 x 1024 per row more in the two-launch path): `replaces_bytes_per_launch`
            algo = {'mlp_fwd_edge': (5 * 512 + 36, 1536 * N_nodes), 'mlp_bwd_edge': (6 * 512 + 36, 1024 * N_nodes),
                    'wgrad': (1024, 0), 'wgrad_node': (1024, 0), 'edge_bwd_fused': (6 * 512 + 36, 512 * N_nodes)}
            # bytes the arithmetic NEEDS: without the dz3 / dz2 / dz1 hand-off to the weight-gradient launch
            necessary = {'mlp_bwd_edge': (3 * 512 + 36, 1024 * N_nodes), 'edge_bwd_fused': (5 * 512 + 36, 512 * N_nodes)}
            replaces = {'edge_bwd_fused': (6 * 512 + 36 + 2 * 1024, 1024 * N_nodes)}      # hgn_mlp_bwd + the dW3 / dW2 tasks of hgn_mlp_wgrad
            cand = {n: v for n, v in k.items() if n in algo and (n in fwd_only or not overlapped)}
            name, v = max(cand.items(), key=lambda kv: kv[1]['ms'])
            t_launch = v['ms'] / v['count'] * 1e-3
            rows = v['units'] / v['count']
            bytes_launch = algo[name][0] * rows + algo[name][1]
            ach = bytes_launch / t_launch / 1e9
            per_row = {'mlp_fwd_edge': 3, 'mlp_bwd_edge': 3, 'wgrad': 1, 'wgrad_node': 1, 'edge_bwd_fused': 5}[name] * 2 * 128 * 128
            tf = per_row * rows / t_launch / 1e12
            fp32_only = bool(os.environ.get('HGN_FP32_MFMA'))
            n_prod = ops._PRODUCTS[args.precision]
            traffic, traffic_note = None, None
            sha = kernel_source_sha()
            try:        # HBM bytes per launch from the rocprofv3 PMC passes of this configuration (profiles/README.md)
                pm = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
                ent = pm.get(name)
                if pm.get('kernel_source_sha') != sha:
                    traffic_note = (f"profiles/pmc_traffic.json was measured on kernel sources {pm.get('kernel_source_sha')} "
                                    f"(commit {pm.get('commit')}); sources now {sha}: dropped as stale")
                elif ent and int(ent['rows_per_launch']) == int(rows) and not fp32_only:
                    traffic = ent['traffic_bytes_per_launch']
                    traffic_note = (f"rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes of this configuration at commit "
                                    f"{pm.get('commit')}, kernel sources {sha}: profiles/pmc_traffic.json")
            except Exception as ex:
                traffic_note = f'no PMC record ({type(ex).__name__})'
            res['roofline'] = {'kernel': name, 'bound': 'hbm', 'achieved': ach, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                               'frac': ach / PEAK_HBM_GBS, 'traffic': traffic,
                               'algorithmic_bytes_per_launch': bytes_launch, 'rows_per_launch': rows,
                               'ms_per_launch': t_launch * 1e3,
                               'traffic_source': traffic_note, 'kernel_source_sha': sha,
                               'necessary_bytes_per_launch': (necessary[name][0] * rows + necessary[name][1]) if name in necessary else bytes_launch,
                               'frac_on_necessary_bytes': ((necessary[name][0] * rows + necessary[name][1]) if name in necessary
                                                           else bytes_launch) / t_launch / 1e9 / PEAK_HBM_GBS,
                               'matrix_pipe': {'fp32_equivalent_TFLOPs': tf, 'flop_per_row': per_row,
                                               'frac_of_fp32_mfma_peak': tf / PEAK_F32_MFMA_TFLOPS,
                                               'mfmas_per_product': None if fp32_only else n_prod,
                                               'frac_of_16bit_mfma_peak': None if fp32_only else n_prod * tf / PEAK_BF16_MFMA_TFLOPS,
                                               'products': 'fp32 MFMA' if fp32_only else PRODUCT_TEXT[n_prod]},
                               'selection': 'largest accumulated time' + (' among forward kernels (side stream on)' if overlapped else '')}
            # what the hardware counters of the same launch say (profiles/sq_counters.json: rocprofv3 --pmc SQ_* passes of
            # tools/fusedbench.py on these kernel sources; dropped when the sources have changed since)
            try:
                sq = json.load(open(os.path.join(ROOT, 'profiles', 'sq_counters.json')))
                ent = sq.get('kernels', {}).get(name)
                if sq.get('kernel_source_sha') == sha and ent:
                    res['roofline']['counters'] = dict(ent, source=f"profiles/sq_counters.json (commit {sq.get('commit')}, kernel sources {sha})")
                    res['roofline']['limited_by'] = limited_by(ent, ach / PEAK_HBM_GBS)
                else:
                    res['roofline']['counters'] = None
                    res['roofline']['limited_by'] = (f"(profiles/sq_counters.json is for kernel sources {sq.get('kernel_source_sha')}, now {sha}: dropped as stale)")
            except Exception as ex:
                res['roofline']['counters'] = None
                res['roofline']['limited_by'] = f'no SQ counter record ({type(ex).__name__})'
            if name in replaces:      # one launch doing the work of several: the bytes THOSE would move, over this kernel's time
                rb = replaces[name][0] * rows + replaces[name][1]
                res['roofline']['replaces_bytes_per_launch'] = rb
                res['roofline']['frac_vs_replaced_launches'] = rb / t_launch / 1e9 / PEAK_HBM_GBS
                res['roofline']['replaces'] = 'hgn_mlp_bwd (edge) + the dW3 / dW2 tasks of hgn_mlp_wgrad: their algorithmic bytes'
            # whole-step matrix utilisation: algorithmic flops of every MFMA launch of the step / step time
            step_flops = sum(v2['units'] / psteps * ({'wgrad': 1, 'wgrad_node': 1, 'linear_fwd': 2, 'linear_bwd': 2, 'edge_bwd_fused': 5}.get(n2, 3)) * 2 * 128 * 128
                             for n2, v2 in k.items() if n2.startswith(('mlp', 'wgrad', 'linear', 'edge_bwd')))
            step_tf = step_flops / (ms_per_step * 1e-3) / 1e12
            res['roofline_step'] = {'bound': 'mfma', 'achieved': step_tf, 'peak': PEAK_F32_MFMA_TFLOPS,
                                    'unit': 'TFLOP/s', 'frac': step_tf / PEAK_F32_MFMA_TFLOPS,
                                    # the pipe these products really run on: n_prod 16-bit MFMAs per fp32-accurate product (3: two scaled fp16
                                    # terms; 6: three bf16 terms), so the hardware roof for fp32-equivalent work is the 16-bit peak / n_prod
                                    # (833 / 417 TFLOP/s) -- `frac` above prices that work against the fp32-MFMA peak and is NOT a hardware fraction
                                    'mfmas_per_product': None if fp32_only else n_prod,
                                    'split_product_roof_TFLOPs': None if fp32_only else PEAK_BF16_MFMA_TFLOPS / n_prod,
                                    'frac_of_split_product_roof': None if fp32_only else step_tf / (PEAK_BF16_MFMA_TFLOPS / n_prod),
                                    'note': 'fp32-equivalent flops (one per fp32 product term, not per bf16 MFMA); lower bound: node MLPs with '
                                            'more than one 128-wide source do more than 3 products'}
            # the scatter-add (segment-sum) kernel vs HBM.  With `sum` aggregation the forward aggregate is formed inside the
            # edge kernel, so the stand-alone launches left in the step are the sender sums of dz1 in the backward (same
            # kernel, rows gathered through the sender permutation); with pna the forward aggregation launch is reported.
            # Algorithmic bytes, sum: 4*D*E + 4*(N+1) + 4*D*N (+ 4*E for the permutation)
            s, which = k.get('seg_fwd_agg'), 'forward aggregation'
            pair = bool(getattr(ops, '_SEG_PAIR', False)) and args.arch == 'none'
            if not s:
                s, which = k.get('seg_fwd'), ('sender AND receiver sums of dz1 (backward) in one pass over the rows: hgn_segment_sum_pair' if pair else
                                              'sender sums of dz1 (backward), rows gathered through the sender permutation')
            # (one edge set only: with several sets of different sizes the mean launch time belongs to no byte count)
            if s and len(sets) == 1:
                n_out = 4 if (args.agg == 'pna' and which == 'forward aggregation') else 1
                bytes_launch = 4 * 128 * E_rank + 4 * (N_nodes + 1) + 4 * 128 * N_nodes * n_out + (0 if which == 'forward aggregation' else 4 * E_rank)
                if pair and which != 'forward aggregation':     # rows once, two [N,128] outputs, two row-pointer arrays, one permutation
                    bytes_launch = 4 * 128 * E_rank + 2 * 4 * (N_nodes + 1) + 2 * 4 * 128 * N_nodes + 4 * E_rank
                t = s['ms'] / s['count'] * 1e-3
                ach = bytes_launch / t / 1e9
                agg_traffic, agg_note = None, None
                try:        # PMC bytes of that launch, same stamped record as the dominant kernel's
                    pm = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
                    ent = pm.get('seg_fwd_agg' if which == 'forward aggregation' else ('seg_pair' if pair else 'seg_fwd'))
                    if pm.get('kernel_source_sha') != sha:
                        agg_note = f"profiles/pmc_traffic.json is for kernel sources {pm.get('kernel_source_sha')}, now {sha}: dropped as stale"
                    elif ent and int(ent.get('rows_per_launch', -1)) in (int(N_nodes), int(E_rank)):
                        agg_traffic = ent['traffic_bytes_per_launch']
                        agg_note = f"rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE at commit {pm.get('commit')}: profiles/pmc_traffic.json"
                except Exception as ex:
                    agg_note = f'no PMC record ({type(ex).__name__})'
                res['roofline_aggregation'] = {'kernel': f"{'seg_sum_pair128' if (pair and which != 'forward aggregation') else 'seg_fwd128'} ({which})", 'bound': 'hbm', 'achieved': ach, 'peak': PEAK_HBM_GBS,
                                               'unit': 'GB/s', 'frac': ach / PEAK_HBM_GBS, 'traffic': agg_traffic, 'traffic_source': agg_note,
                                               'bytes_per_launch': bytes_launch, 'ms_per_launch': t * 1e3,
                                               'north_star_target': 'at least 0.40 of the HBM roofline on the scatter-add aggregation'}
        if collective is not None:
            res['collective'] = collective
        if 'graph_build' in wk:
            res['config']['graph_build'] = wk['graph_build']
        if cold is not None:
            res['cold_step'] = cold
        # the headline measurement is complete: hand its device memory back before the children and the CPU leg run (the
        # cylinder shape keeps 25 layers of saved activations for 128 graphs)
        cpu_graph = wk['cpu_graph']() if (not dp and not args.no_cpu_baseline) else None
        cpu_state = {k: v.detach().cpu() for k, v in model.state_dict().items()} if cpu_graph is not None else None
        if not dp:
            import gc
            gstep = step = trainer = model = graph = target = mask = wk = cache = loss = None     # noqa: F841
            gc.collect()
            torch.cuda.empty_cache()
        if not dp and not args.no_secondary:
            res['secondary'] = secondary_configs(args)
        if cpu_graph is not None:
            res['cpu_baseline'] = cpu_baseline(args, cpu_graph, cpu_state)
        print(json.dumps(res))
    if dp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
