"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/hgn_mp.h declares, argument
validation answers with status codes (no compute), the module tree / state_dict contract, loud failure without a GPU,
and the data-parallel host logic over gloo with world_size 2."""
import ctypes as C
import json
import os
import re
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import mgn_oracle as O
from tests import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_match_header():
    from hgn_amd import _lib
    header = ''.join(open(os.path.join(ROOT, 'include', h)).read() for h in ('hgn_mp.h', 'hgn_features.h'))
    declared = set(re.findall(r'^\s*(?:const\s+char\s*\*|int)\s+(hgn_\w+)\s*\(', header, flags=re.M))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.hgn_version() >= 100


def test_abi_struct_sizes_match_header_layout():
    """ctypes mirrors of the C structs: sizes follow the header's field lists under the x86-64 SysV ABI."""
    from hgn_amd import _lib
    assert C.sizeof(_lib.Src) == 48 and C.sizeof(_lib.Add) == 24 and C.sizeof(_lib.Dx) == 48
    assert C.sizeof(_lib.WTask) == 96 + 8                 # + per-call options: products, flags
    assert C.sizeof(_lib.MlpFwd) == 8 + 8 + 8 * 48 + 8 + 2 * 24 + 8 * 6 + 8 + 8 * 2 + 8 * 2 + 8 * 2 + 8 * 4 + 8 * 2 + 8 + 24 + (4 * 8 + 8 + 8 + 8 + 8 + 8) + 8
    assert C.sizeof(_lib.Pack) == 40
    assert C.sizeof(_lib.LnTask) == 8 + 8 + 8 + 8 + 4 + 4                      # hgn_ln_task_t (include/hgn_mp.h)
    assert C.sizeof(_lib.WRed) == 6 * 4 + 8 + 8 + 8 + 8 + 8                    # hgn_wred_task_t
    assert C.sizeof(_lib.MlpBwd) == 8 + 8 + 8 + 8 + 8 * 7 + 8 + 8 * 3 + 8 + 8 * 48 + (8 + 8 + 4 + 16 + 4) + 8 * 4 + 8 * 3 + 8 + 8 * 2 + 8 + 24 + 8


def test_abi_argument_validation_without_gpu():
    from hgn_amd import _lib
    lib = _lib.lib()
    nb = C.c_size_t(0)
    assert lib.hgn_csr_workspace_bytes(1000, 10, C.byref(nb)) == 0 and nb.value > 8000
    assert lib.hgn_csr_workspace_bytes(-1, 10, C.byref(nb)) == -1
    assert lib.hgn_wgrad_workspace_bytes(100000, 4, C.byref(nb)) == 0 and nb.value > 0
    assert lib.hgn_wgrad_workspace_bytes(10, 99, C.byref(nb)) == -1
    a = _lib.MlpFwd()
    a.M = 5
    a.out_w = 500
    assert lib.hgn_mlp_fwd(C.byref(a), None) == -1
    assert b'out_w' in lib.hgn_last_error()
    ops = (C.c_int32 * 1)(7)
    assert lib.hgn_segment_reduce_fwd(None, 128, 128, None, None, 4, ops, 1, None, 128, None, None, None) == -1
    assert b'Invalid operation type' in lib.hgn_last_error()
    # include/hgn_features.h
    assert lib.hgn_cells_to_edges_workspace_bytes(100, 3, C.byref(nb)) == 0 and nb.value > 3 * 300 * 8
    assert lib.hgn_cells_to_edges_workspace_bytes(100, 5, C.byref(nb)) == -1
    assert lib.hgn_col_stats_workspace_bytes(1000, 7, C.byref(nb)) == 0 and nb.value > 0
    assert lib.hgn_col_stats_workspace_bytes(1000, 33, C.byref(nb)) == -1
    assert lib.hgn_rel_edge_features(None, 3, 4, None, 0, 0, 10, None, None, 5, None, 8, None, None) == -1
    assert b'1<=da<=3' in lib.hgn_last_error()
    assert lib.hgn_node_features(None, None, 3, 3, None, 1, None, 0, 40, 1, -1, 5, None, 43, None) == -1
    assert lib.hgn_normalize(None, 5, 0, None, None, None, 1e-8, 0, None, None) == -1
    assert lib.hgn_lincomb3(None, 1.0, None, 1.0, None, 0.0, 5, None, None) == -1
    # deferred gradient sums (hgn_ln_reduce_batch / hgn_slab_reduce_batch): empty batches are fine, bad ones are refused before any launch
    assert lib.hgn_ln_reduce_batch(None, 0, None) == 0 and lib.hgn_slab_reduce_batch(None, 0, None) == 0
    assert lib.hgn_ln_reduce_batch(None, 3, None) == -1 and lib.hgn_slab_reduce_batch(None, 3, None) == -1
    lt = (_lib.LnTask * 2)()
    assert lib.hgn_ln_reduce_batch(lt, 2, None) == -1 and b'bad task' in lib.hgn_last_error()
    assert lib.hgn_ln_reduce_batch(lt, _lib.HGN_MAX_LN_TASK + 1, None) == -1
    buf = (C.c_float * 8)()
    for t in lt:                                      # two tasks with ONE target in a batch: refused (the sums are added without atomics)
        t.ln_ws = C.cast(buf, _lib.c_f32p); t.M = 64; t.d_gamma = C.cast(buf, _lib.c_f32p); t.d_beta = C.cast(buf, _lib.c_f32p)
    assert lib.hgn_ln_reduce_batch(lt, 2, None) == -1 and b'share a target' in lib.hgn_last_error()
    wr = (_lib.WRed * 2)()
    for r in wr:
        r.type = 0; r.K = 128; r.n_out = 128; r.n_chunks = 4; r.dW = C.cast(buf, _lib.c_f32p); r.slab = C.cast(buf, _lib.c_f32p); r.ldw = 128
    assert lib.hgn_slab_reduce_batch(wr, 2, None) == -1 and b'share a target' in lib.hgn_last_error()
    assert lib.hgn_mlp_wgrad_partial(None, 2, 100, None, 0, None, None) == -1
    # precision switch: 3 (default) / 6 (fp32-accurate: two scaled fp16 / three bf16 terms), 1 (bf16), 2 (fp16 forward, mode-3 backward);
    # anything else is refused and changes nothing
    assert lib.hgn_get_matmul_products() == 3
    for n in (1, 2, 6, 3):
        assert lib.hgn_set_matmul_products(n) == 0 and lib.hgn_get_matmul_products() == n
    assert lib.hgn_set_matmul_products(4) == -1 and lib.hgn_get_matmul_products() == 3
    # the shipped library carries no laboratory variants (tools/lab/): their switches are not exported
    raw = C.CDLL(_lib.LIB_PATH)
    assert not any(hasattr(raw, n) for n in ('hgn_set_ws_fwd', 'hgn_set_big_tiles', 'hgn_mlp_fwd_ws_eligible'))
    with pytest.raises(_lib.HgnError):
        _lib.check(-1, 'x')
    with pytest.raises(IndexError):
        _lib.check(-3, 'x')


def test_module_tree_matches_reference_state_dict_keys():
    import hgn_amd
    sets = ['mesh_edges', 'intra_cluster_to_mesh', 'intra_cluster_to_cluster', 'inter_cluster']
    for arch, agg in (('none', 'sum'), ('hyper', 'pna'), ('hetero', 'pna'), ('multiscale', 'sum'), ('repeated', 'max'),
                      ('multi', 'min')):
        use = sets if arch in ('hyper', 'hetero', 'multiscale') else ['mesh_edges']
        shapes = O.param_shapes(arch, agg, 2, use, 5, {n: 7 for n in use}, 8, 3, 128)
        m = hgn_amd.MeshGraphNet(3, 128, 2, agg, 2, arch, use)
        assert all(isinstance(p, torch.nn.parameter.UninitializedParameter) for n, p in m.named_parameters()
                   if n.endswith('linear_0.weight'))
        m.load_state_dict(O.init_state_dict_like(shapes, 0), strict=True)        # reference key names, lazy shapes
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == dict(shapes)
    # golden key names come from the reference itself
    fx = torch.load(os.path.join(ROOT, 'tests', 'golden', 'mgn_hyper_pna_L1_lat128.pt'))
    m = hgn_amd.MeshGraphNet(3, 128, 2, 'pna', 1, 'hyper', fx['edge_sets'])
    m.load_state_dict(O.init_state_dict_like(fx['shapes'], 0), strict=True)


def test_product_path_fails_loudly_without_gpu():
    import hgn_amd
    from hgn_amd import _lib
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    g = synth.grid_graph(nx=4, ny=4)
    m = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges'])
    with pytest.raises(_lib.HgnError, match='no CPU fallback'):
        m(hgn_amd.MultiGraph(g.node_features, g.edge_sets))
    with pytest.raises(_lib.HgnError):
        hgn_amd.unsorted_segment_operation(torch.randn(4, 128), torch.tensor([0, 1, 1, 0]), 2, 'sum')
    with pytest.raises(_lib.HgnError, match='latent_size=128'):
        hgn_amd.MeshGraphNet(3, 64, 2, 'sum', 1, 'none', ['mesh_edges'])(hgn_amd.MultiGraph(g.node_features, g.edge_sets))


def test_no_product_import_of_oracle():
    pkg = os.path.join(ROOT, 'hyper-graph-nets_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('# oracle', ''), f'{f} mentions the oracle'
    # outside the package only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs may import oracle/
    import re
    pat = re.compile(r'^\s*(from\s+oracle\b|import\s+oracle\b)', re.M)
    for dirpath, dirs, files in os.walk(ROOT):
        dirs[:] = [d for d in dirs if d not in ('.git', 'gpurun_out', 'tests', 'oracle', '__pycache__', '_build')]
        for f in files:
            if f.endswith('.py') and pat.search(open(os.path.join(dirpath, f)).read()):
                rel = os.path.relpath(os.path.join(dirpath, f), ROOT)
                assert rel in ('bench.py', '__graft_entry__.py'), f'{rel} imports the oracle'


def test_feature_path_refuses_cpu_tensors():
    """Normaliser / cell-edge / feature kernels are HIP only: a CPU tensor must raise, not fall back to torch math."""
    from hgn_amd import _lib, features, util
    from hgn_amd.normalizer import Normalizer
    nz = Normalizer(5, 't')
    assert 'acc' not in ''.join(nz.state_dict().keys())            # statistics are not buffers (reference semantics)
    if torch.cuda.is_available():
        pytest.skip('CPU-only check')
    with pytest.raises(_lib.HgnError):
        nz(torch.randn(4, 5))
    with pytest.raises(_lib.HgnError):
        util.triangles_to_edges(torch.tensor([[0, 1, 2]]))
    with pytest.raises(_lib.HgnError):
        features.rel_edge_features(torch.randn(3, 3), None, torch.tensor([0]), torch.tensor([1]))


def test_synthetic_edges_and_batcher():
    from hgn_amd import synthetic
    faces = synthetic.grid_triangles(4, 3)
    s, r = synthetic.two_way_edges(faces)
    assert s.dtype == torch.int64 and s.shape == r.shape
    pairs = set(zip(s.tolist(), r.tolist()))
    assert all((b, a) in pairs for a, b in pairs) and len(pairs) == s.shape[0]
    # batching: correct hyper mapping equals the oracle's non-compat mapping
    gs = [synthetic.grid_graph(seed=i, nx=3, ny=3, clusters=2) for i in range(3)]
    a = synthetic.batch(gs)
    b = O.batch_graphs([O.MultiGraph(g.node_features, [O.EdgeSet(*e) for e in g.edge_sets]) for g in gs])
    for ea, eb in zip(a.edge_sets, b.edge_sets):
        assert torch.equal(ea.senders, eb.senders) and torch.equal(ea.receivers, eb.receivers)


@pytest.mark.parametrize('fixture,cls,K', [('feat_flag_hyper_k5', 'KMeansClustering', 5),
                                           ('feat_flag_hyper_k6_spectral', 'SpectralClustering', 6),
                                           ('feat_flag_hyper_k4_gmm', 'GaussianMixtureClustering', 4)])
def test_clustering_host_logic_matches_reference_labels(fixture, cls, K):
    """Cluster labels stay on scikit-learn (host, once per trajectory): same labels, member lists and neighbour pairs as the
    reference produced for the golden frames -- k-means (k_means_clustering.py:27-33), spectral on the edge-length affinity
    (spectral_clustering.py:26-64: what flag.yaml / minimal.yaml / plateCluster.yaml configure) and Gaussian mixture
    (gaussian_mixture.py:24-30).  `random` has no fixture: the reference's RandomClustering.run never sets
    `neigboring_clusters`, so RemoteMessagePassing.create_graph raises at remote_message_passing.py:78."""
    from hgn_amd import rmp, util
    fx = torch.load(os.path.join(ROOT, 'tests', 'golden', fixture + '.pt'), weights_only=False)
    fr, ex, ref = fx['frames'][0], fx['expanded'][0], fx['graphs'][0]
    es, ue = ref['edge_sets'][0], ref['unnormalized_edges']
    g = util.MultiGraphWithPos(node_features=ref['node_features'][0],
                               edge_sets=[util.EdgeSet('mesh_edges', es['features'], es['senders'], es['receivers'])],
                               target_feature=fr['world_pos'], mesh_features=fr['mesh_pos'], model_type='flag',
                               node_dynamic=None,
                               unnormalized_edges=util.EdgeSet(ue['name'], ue['features'], ue['senders'], ue['receivers']),
                               obstacle_nodes=None)
    alg = getattr(rmp, cls)(K, False, 0.1, 0)
    clusters = alg.run(g)
    assert alg._labels == ex['labels']
    assert all(torch.equal(a, b) for a, b in zip(clusters, ex['clusters']))
    want = sorted({(min(a, b), max(a, b)) for a, b in (tuple(t.tolist()) for t in ex['neighbors'])})
    assert [tuple(t.tolist()) for t in alg.neigboring_clusters] == want


# -------------------------------------------------------------------------------------------------------------
# data-parallel host logic on gloo, world_size 2: the model is a stand-in (the CPU oracle wrapped as nn.Module) because
# the product kernels need the GPU; what is tested is sharding, global-mean loss scaling, the single flat all-reduce,
# the flat Adam bookkeeping and normaliser-statistic reduction.
# -------------------------------------------------------------------------------------------------------------
class _OracleNet(torch.nn.Module):
    def __init__(self, sd):
        super().__init__()
        self.keys = list(sd.keys())
        self.ps = torch.nn.ParameterList([torch.nn.Parameter(v.clone()) for v in sd.values()])

    def forward(self, graph):
        sd = dict(zip(self.keys, self.ps))
        return O.mesh_graph_net(sd, graph, 'none', 'sum')


def _dp_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from hgn_amd import parallel, synthetic
    from hgn_amd.normalizer import Normalizer
    torch.set_num_threads(1)
    graphs = [synthetic.grid_graph(seed=10 + i, nx=4, ny=4) for i in range(4)]
    shapes = O.param_shapes('none', 'sum', 1, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 16)
    sd = O.init_state_dict_like(shapes, 3 + rank)                 # different per rank: broadcast must fix it
    model = _OracleNet(sd)
    tr = parallel.DataParallelTrainer(model, lr=1e-2, adam_fn=parallel._torch_adam)
    mine = parallel.shard_indices(4, rank, world)
    g = synthetic.batch([graphs[i] for i in mine])
    gen = torch.Generator().manual_seed(0)
    targets = torch.randn(4, 16, 3, generator=gen)
    masks = torch.ones(4, 16, dtype=torch.bool)
    masks[0, :5] = False                                           # unequal mask counts across ranks
    target = torch.cat([targets[i] for i in mine])
    mask = torch.cat([masks[i] for i in mine])
    losses = [float(tr.step(O.MultiGraph(g.node_features, [O.EdgeSet(*e) for e in g.edge_sets]), target, mask))
              for _ in range(3)]
    nz = Normalizer(3, 'n')
    parallel.attach_normalizer_sync([nz])
    x = targets[mine[0]] * (rank + 1.0)         # the statistics kernels need the GPU: exercise the all-reduce hook itself
    cnt, s1, s2 = nz._reduce_fn(torch.tensor([float(x.shape[0])]), x.sum(0), (x ** 2).sum(0))
    if rank == 0:
        torch.save({'flat': tr.fp.flat.clone(), 'losses': losses, 'acc_sum': s1, 'acc_sq': s2, 'acc_count': cnt}, out)
    flat0 = tr.fp.flat.clone()
    dist.broadcast(flat0, src=0)
    assert torch.equal(flat0, tr.fp.flat), 'replicas diverged'
    dist.destroy_process_group()


def test_data_parallel_equals_single_process_on_concatenated_batch(tmp_path):
    from hgn_amd import parallel, synthetic
    port = 29500 + (os.getpid() % 2000)
    out = str(tmp_path / 'dp.pt')
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    # single process on the whole batch, reference semantics: mean over all NORMAL nodes (flag.py:150-152)
    graphs = [synthetic.grid_graph(seed=10 + i, nx=4, ny=4) for i in range(4)]
    shapes = O.param_shapes('none', 'sum', 1, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 16)
    model = _OracleNet(O.init_state_dict_like(shapes, 3))          # rank 0's weights
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    order = parallel.shard_indices(4, 0, 2) + parallel.shard_indices(4, 1, 2)
    g = synthetic.batch([graphs[i] for i in order])
    gen = torch.Generator().manual_seed(0)
    targets = torch.randn(4, 16, 3, generator=gen)
    masks = torch.ones(4, 16, dtype=torch.bool)
    masks[0, :5] = False
    target = torch.cat([targets[i] for i in order])
    mask = torch.cat([masks[i] for i in order])
    for step in range(3):
        opt.zero_grad()
        outp = model(O.MultiGraph(g.node_features, [O.EdgeSet(*e) for e in g.edge_sets]))
        loss = O.masked_mse(outp, target, mask)
        loss.backward()
        opt.step()
    flat = torch.cat([torch.nn.functional.pad(p.detach().reshape(-1), (0, (-p.numel()) % 4)) for p in model.parameters()])
    torch.testing.assert_close(res['flat'], flat, rtol=2e-5, atol=2e-6)
    # normaliser: both ranks accumulated the union of the two inputs
    both = torch.cat([targets[0] * 1.0, targets[1] * 2.0])
    torch.testing.assert_close(res['acc_sum'], both.sum(0), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(res['acc_sq'], (both ** 2).sum(0), rtol=1e-5, atol=1e-5)
    assert float(res['acc_count']) == 32.0


class _LayeredNet(torch.nn.Module):
    """Stand-in with the module layout the bucket planner looks for (`processor.graphnet_blocks`, MeshGraphNet's): encoder ->
    six residual blocks over a (node rows, edge rows) pair -> decoder.  Rows are independent, so sharding rows over ranks is
    sharding graphs."""

    class Block(torch.nn.Module):
        def __init__(self, d):
            super().__init__()
            self.node, self.edge = torch.nn.Linear(d, d), torch.nn.Linear(d, d)

        def forward(self, x):
            nodes, edges = x
            edges = edges + torch.tanh(self.edge(edges))
            nodes = nodes + torch.tanh(self.node(nodes)) * edges.detach()
            return nodes, edges

    def __init__(self, d=8, blocks=6):
        super().__init__()
        self.encoder = torch.nn.Linear(5, d)
        self.processor = torch.nn.Module()
        self.processor.graphnet_blocks = torch.nn.Sequential(*[self.Block(d) for _ in range(blocks)])
        self.decoder = torch.nn.Linear(d, 3)

    def forward(self, x):
        nodes, edges = self.processor.graphnet_blocks((self.encoder(x), self.encoder(x).detach() * 0.5 + self.encoder(x) * 0.1))
        return self.decoder(nodes) + 0.01 * edges.sum(1, keepdim=True)


def _bucket_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from hgn_amd import parallel
    torch.set_num_threads(1)
    gen = torch.Generator().manual_seed(5)
    x, y = torch.randn(40, 5, generator=gen), torch.randn(40, 3, generator=gen)
    mask = torch.ones(40, dtype=torch.bool); mask[:7] = False        # unequal NORMAL-node counts on the two ranks
    mine = torch.arange(rank, 40, world)
    res = {}
    for buckets in (1, 3):
        torch.manual_seed(100 + rank)                                   # different initial weights per rank: the broadcast fixes it
        model = _LayeredNet()
        tr = parallel.DataParallelTrainer(model, lr=1e-2, adam_fn=parallel._torch_adam, buckets=buckets)
        calls = []
        orig = tr._reduce_from
        tr._reduce_from = lambda start, _o=orig, _c=calls: (_c.append((start, tr._done_upto)), _o(start))[1]
        if rank == 0 and buckets == 3:
            torch.manual_seed(100)
            start_weights = torch.cat([torch.nn.functional.pad(p.detach().reshape(-1), (0, (-p.numel()) % 4)) for p in _LayeredNet().parameters()])
            assert torch.equal(tr.fp.flat, start_weights)
        losses = [float(tr.step(x[mine], y[mine], mask[mine])) for _ in range(3)]
        res[buckets] = {'flat': tr.fp.flat.clone(), 'losses': losses, 'starts': list(tr.bucket_starts), 'calls': calls[-3:] if buckets == 3 else calls[-1:],
                        'numel': tr.fp.grad_ext.numel()}
    if rank == 0:
        torch.save(res, out)
    dist.destroy_process_group()


def test_bucketed_overlapped_all_reduce_equals_one_all_reduce_and_single_process(tmp_path):
    """The flat gradient buffer reduced in three ranges, each launched from an autograd node at its boundary (last layers first,
    while the backward pass goes on; the range of the first parameters -- with the NORMAL-node count in its spare slot -- after
    it): same parameters and losses as ONE all-reduce of the whole buffer, bit for bit on gloo, and as single-process Adam on the
    concatenated batch with the global mean (flag.py:150-152)."""
    port = 23500 + (os.getpid() % 2000)
    out = str(tmp_path / 'bk.pt')
    mp.spawn(_bucket_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    one, three = res[1], res[3]
    assert one['starts'] == [] and len(three['starts']) == 2 and three['starts'] == sorted(three['starts'])
    # the three ranges of a step: [second boundary, end), [first boundary, second), [0, first) -- every element exactly once
    (s0, u0), (s1, u1), (s2, u2) = three['calls']
    assert (s0, u0) == (three['starts'][1], None) and (s1, u1) == (three['starts'][0], three['starts'][1]) and (s2, u2) == (0, three['starts'][0])
    assert torch.equal(one['flat'], three['flat']) and one['losses'] == three['losses']
    gen = torch.Generator().manual_seed(5)
    x, y = torch.randn(40, 5, generator=gen), torch.randn(40, 3, generator=gen)
    mask = torch.ones(40, dtype=torch.bool); mask[:7] = False
    torch.manual_seed(100)
    model = _LayeredNet()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    for _ in range(3):
        opt.zero_grad()
        loss = ((model(x) - y)[mask] ** 2).mean()
        loss.backward()
        opt.step()
    flat = torch.cat([torch.nn.functional.pad(p.detach().reshape(-1), (0, (-p.numel()) % 4)) for p in model.parameters()])
    torch.testing.assert_close(three['flat'], flat, rtol=2e-5, atol=2e-6)


def test_batcher_matches_reference_golden_g6():
    """f1: vectorised batcher; reference_compat mode reproduces MeshSimulator._get_batched's index arithmetic (golden G6,
    generated by the reference itself), the default mode keeps every graph's remote edges inside that graph."""
    from hgn_amd import batching, util
    g6 = torch.load(os.path.join(ROOT, 'tests', 'golden', 'g6_get_batched.pt'))
    for B, fx in g6.items():
        graphs = []
        for gi in fx['in']:
            nf = [torch.zeros(n, 1) for n in gi['n']]
            graphs.append(util.MultiGraph(nf, [util.EdgeSet(nm, torch.zeros(s.shape[0], 1), s, r) for nm, s, r in gi['sets']]))
        compat = batching.batch_graphs(graphs, reference_compat=True)
        assert [x.shape[0] for x in compat.node_features] == fx['n_out']
        for e, (nm, s, r) in zip(compat.edge_sets, fx['out']):
            assert e.name == nm and torch.equal(e.senders, s) and torch.equal(e.receivers, r), (B, nm)
        fixed = batching.batch_graphs(graphs)
        ora = O.batch_graphs([O.MultiGraph(g.node_features, [O.EdgeSet(*e) for e in g.edge_sets]) for g in graphs])
        for a, b in zip(fixed.edge_sets, ora.edge_sets):
            assert torch.equal(a.senders, b.senders) and torch.equal(a.receivers, b.receivers)


def test_batcher_ragged_edge_counts_and_configured_batch_size():
    """Per-graph edge counts may differ inside a batch (the reference concatenates per-graph lists, MeshSimulator.py:186-231);
    a shorter last batch keeps the CONFIGURED batch size in the reference's hyper-id offset (MeshSimulator.py:196)."""
    from hgn_amd import batching, util
    def graph(n_edges, seed):
        gen = torch.Generator().manual_seed(seed)
        s = torch.randint(0, 6, (n_edges,), generator=gen)
        r = torch.randint(0, 6, (n_edges,), generator=gen)
        up_s, up_r = torch.arange(4), 4 + torch.arange(4) % 2
        return util.MultiGraph([torch.randn(4, 2, generator=gen), torch.randn(2, 3, generator=gen)],
                               [util.EdgeSet('mesh_edges', torch.randn(n_edges, 3, generator=gen), s, r),
                                util.EdgeSet('intra_cluster_to_cluster', torch.randn(4, 3, generator=gen), up_s, up_r)])
    graphs = [graph(3, 0), graph(4, 1), graph(0, 2)]
    for compat in (False, True):
        got = batching.batch_graphs(graphs, reference_compat=compat)
        ora = O.batch_graphs([O.MultiGraph(g.node_features, [O.EdgeSet(*e) for e in g.edge_sets]) for g in graphs], reference_compat=compat)
        for a, b in zip(got.edge_sets, ora.edge_sets):
            assert torch.equal(a.senders, b.senders) and torch.equal(a.receivers, b.receivers) and torch.equal(a.features, b.features)
        assert all(torch.equal(a, b) for a, b in zip(got.node_features, ora.node_features))
    # last batch of a trajectory: 2 graphs under a configured batch size of 3 -> hyper threshold 3 * n_mesh = 12 (never reached)
    short = batching.batch_graphs(graphs[:2], reference_compat=True, batch_size=3)
    assert short.edge_sets[1].receivers.tolist() == [4, 5, 4, 5, 8, 9, 8, 9]
    with pytest.raises(ValueError):
        bad = util.MultiGraph([graphs[0].node_features[0], torch.zeros(3, 3)], graphs[0].edge_sets)     # differing n_hyper
        batching.batch_graphs([graphs[0], bad])


def test_random_clustering_formula_and_get_rmp_surface():
    """random_clustering.py:38-39: label = int(rand * K) per node from numpy's global generator; get_rmp.py's three functions
    exist with the reference's names and accept the reference's config keys (incl. every clustering / connector name)."""
    import numpy as np
    from hgn_amd import rmp, util
    g = util.MultiGraphWithPos(None, [util.EdgeSet('mesh_edges', None, torch.tensor([0, 1, 2]), torch.tensor([1, 2, 3]))],
                               torch.zeros(9, 3), None, 'flag', None, None, None)
    np.random.seed(3)
    want = [int(x) for x in np.random.rand(9) * 4]
    np.random.seed(3)
    assert rmp.RandomClustering(4, False, 0.1, 0)._cluster(g) == want
    cfgs = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'configs_model_sections.json')))
    for name, cfg in cfgs.items():
        r = cfg['model']['rmp']
        alg = rmp.get_clustering_algorithm(str(r['clustering']).lower(), cfg['model'])
        conn = rmp.get_connector(str(r['connector']).lower(), cfg['model'])
        assert (alg is None) == (r['clustering'] == 'none') and (conn is None) == (r['connector'] in ('none', 'repeated')), name
    for cl in ('random', 'spectral', 'gmm', 'kmeans', 'k-means', 'none'):
        rmp.get_clustering_algorithm(cl, cfgs['flag']['model'])
    h = rmp.get_clustering_algorithm('hdbscan', cfgs['flag']['model'])           # get_rmp.py:47-69: arguments from the `hdbscan` block
    assert isinstance(h, rmp.HDBSCANClustering) and (h._max_cluster_size, h._min_cluster_size, h._min_samples) == (50, 20, 1)
    with pytest.raises(NotImplementedError):
        rmp.get_clustering_algorithm('dbscan', cfgs['flag']['model'])
    with pytest.raises(NotImplementedError):
        rmp.get_connector('multigraph', cfgs['flag']['model'])                # get_rmp.py:92 rejects it too


def test_shard_indices_partition():
    from hgn_amd import parallel
    for n, w in ((8, 8), (21, 8), (3, 2), (64, 4)):
        parts = [parallel.shard_indices(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))


def test_abi_radius_edges_validation():
    from hgn_amd import _lib
    lib = _lib.lib()
    nb = C.c_size_t(0)
    assert lib.hgn_radius_edges_workspace_bytes(1000, C.byref(nb)) == 0 and nb.value >= 4004
    tot = C.c_int64(0)
    assert lib.hgn_radius_edges_count(None, 3, 5, None, 1, 10, 0.03, 1, 0, None, None, None, C.byref(tot), None, 0, None) == -1
    assert lib.hgn_radius_edges_fill(None, 3, 3, None, 1, 10, 0.03, 1, 0, None, None, None, None, None, None) == -1


def test_intra_cluster_sampling_matches_reference_clusters():
    """Spotter / exemplars / highest-dynamics sampling (host logic with the `random` module): same sampled cluster members,
    in the same order, as the reference produced under random.seed(0) (tests/golden/feat_flag_hyper_k4_sampled.pt)."""
    import random
    from hgn_amd import rmp, util
    fx = torch.load(os.path.join(ROOT, 'tests', 'golden', 'feat_flag_hyper_k4_sampled.pt'), weights_only=False)
    fr, ex, ref = fx['frames'][0], fx['expanded'][0], fx['graphs'][0]
    es = ref['edge_sets'][0]
    g = util.MultiGraphWithPos(node_features=ref['node_features'][0],
                               edge_sets=[util.EdgeSet('mesh_edges', es['features'], es['senders'], es['receivers'])],
                               target_feature=fr['world_pos'], mesh_features=fr['mesh_pos'], model_type='flag',
                               node_dynamic=ref['node_dynamic'], unnormalized_edges=None, obstacle_nodes=None)
    random.seed(0)
    alg = rmp.KMeansClustering(4, True, 0.1, 0)
    clusters = alg.run(g)
    assert [c.tolist() for c in clusters] == [c.tolist() for c in ex['clusters']]
    assert sum(len(c) for c in clusters) < 168                     # a strict subset of the nodes is connected upwards


def test_fused_backward_ring_counts_match_the_emitted_instructions(tmp_path):
    """csrc/fused_bwd.hip retires its weight-ring LDS-DMA with COUNTED waits (s_waitcnt vmcnt(N) in front of every phase barrier of
    the weight-gradient waves): N is the number of vector-memory instructions the wave issues after the DMA of the piece the
    phase reads -- the next piece's DMA (6 instructions) and the operand fetches (8 loads) of the two phases in between.  The
    fetches are ordinary loads emitted by the compiler: if a toolchain merges, splits, duplicates or spills anything in that
    loop, the counts are wrong and the chain reads weights that have not landed.  So the build is checked here, on the
    disassembly: per phase of the weight-gradient loop exactly 6 DMA instructions, 8 loads in the four fetch phases and none
    elsewhere, no store and no scratch access, and the wait immediates are the table of fused_bwd.hip (Keep<>)."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    src = os.path.join(ROOT, 'hyper-graph-nets_amd', 'csrc', 'fused_bwd.hip')
    out = str(tmp_path / 'fb.s')
    r = subprocess.run([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'),
                        '--cuda-device-only', '-S', src, '-o', out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    # the same checker the Makefile runs on the assembly of the object it links (csrc/check_fused_counts.py)
    sys.path.insert(0, os.path.join(ROOT, 'hyper-graph-nets_amd', 'csrc'))
    try:
        import check_fused_counts
    finally:
        sys.path.pop(0)
    check_fused_counts.check(open(out).read())
    # ... and it does reject a stream whose counts are off: one fetch load more in phase 1
    text = open(out).read()
    name = check_fused_counts.KERNEL
    end = text.index('.Lfunc_end', text.index(name + ':'))
    at = text.rindex('global_load_dwordx2', 0, end)            # the last fetch load of the weight-gradient loop
    broken = text[:at] + 'global_load_dwordx2 v[0:1], v0, s[0:1]\n\t' + text[at:]
    with pytest.raises(AssertionError):
        check_fused_counts.check(broken)


def test_bucket_hooks_and_trainer_state_stay_out_of_model_pickles():
    """A model under DataParallelTrainer carries forward hooks at its bucket boundaries; the reference checkpoints with pickle.dump of
    the object that holds the network (MeshSimulator.py:492-493).  The hook object pickles as a detached no-op (it references the
    trainer: process group, streams), and a `_hgn_grad` tag that a reloaded parameter still carries is ignored by the kernels' gradient
    routing because it is no longer that parameter's .grad storage."""
    import pickle
    from hgn_amd import ops, parallel

    class Holder:
        overlap = True

        def _reduce_from(self, start):
            raise AssertionError('a detached hook must never reach a trainer')

    lin = torch.nn.Linear(4, 4)
    lin.register_forward_hook(parallel._BucketHook(Holder(), 128))
    twin = pickle.loads(pickle.dumps(lin))
    hooks = list(twin._forward_hooks.values())
    assert len(hooks) == 1 and isinstance(hooks[0], parallel._DetachedHook)
    x = torch.randn(2, 4, requires_grad=True)
    twin(x).sum().backward()                                   # the detached hook leaves the output alone
    # flat-gradient tags: valid while they ARE the .grad storage, ignored on a reloaded copy
    fp = parallel.FlatParams(lin)
    assert [t is not None for t in ops._grad_targets(list(lin.parameters()))] == [True, True]
    reloaded = pickle.loads(pickle.dumps(lin))
    assert all(hasattr(p, '_hgn_grad') for p in reloaded.parameters())          # the attribute does travel ...
    assert ops._grad_targets(list(reloaded.parameters())) == [None, None]        # ... and is not trusted
    assert fp.left_lazy == []


def test_flat_params_reports_parameters_left_lazy_and_refuses_late_materialisation():
    from hgn_amd import parallel
    net = torch.nn.Sequential(torch.nn.Linear(3, 3), torch.nn.LazyLinear(2))
    with pytest.warns(UserWarning, match='still lazy'):
        fp = parallel.FlatParams(net)
    assert fp.left_lazy == ['1.weight', '1.bias']
    fp.check_outsiders()                                       # still lazy: nothing to complain about
    net(torch.randn(1, 3))                                     # the lazy layer materialises OUTSIDE the flat buffer
    with pytest.raises(RuntimeError, match='outside the flat buffer'):
        fp.check_outsiders()


def test_hdbscan_clustering_host_logic():
    """`clustering: hdbscan` (src/rmp/hdbscan.py:13-105, get_rmp.py:47-69) on scikit-learn's HDBSCAN with the reference's arguments:
    labels equal the fixture (tests/golden/gen_hdbscan_fixture.py: the scikit-learn port is what is pinned -- the wheel the reference
    imports is absent from this image), noise nodes (-1) join no cluster, neighbouring clusters come from mesh edges between two
    different non-noise labels, the obstacle offset of `run(graph, number, b4)` pads the label list like the other algorithms, and
    intra-cluster sampling (wheel internals) raises instead of doing something else."""
    import importlib.util
    from hgn_amd import rmp, util
    spec = importlib.util.spec_from_file_location('gen_hdbscan_fixture', os.path.join(ROOT, 'tests', 'golden', 'gen_hdbscan_fixture.py'))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    fx = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'hdbscan_labels.json')))
    X = torch.from_numpy(gen.cloud(fx['seed']))
    n = X.shape[0]
    snd = torch.arange(n - 1)
    es = util.EdgeSet('mesh_edges', torch.zeros(n - 1, 7), snd, snd + 1)                 # a path: neighbours = label changes along it
    graph = util.MultiGraphWithPos(node_features=[torch.zeros(n, 5)], edge_sets=[es], target_feature=X, mesh_features=X[:, :2],
                                   model_type='flag', node_dynamic=torch.zeros(n), unnormalized_edges=es, obstacle_nodes=None)
    cfg = {'rmp': {'clustering': 'hdbscan', 'connector': 'hyper', 'num_clusters': 10,
                   'intra_cluster_sampling': {'enabled': False, 'alpha': 0.1, 'spotter_threshold': 0}, 'hdbscan': dict(fx['args'], spotter_threshold=0.9)}}
    algo = rmp.get_clustering_algorithm('hdbscan', cfg)
    assert isinstance(algo, rmp.HDBSCANClustering)
    clusters = algo.run(graph)
    labels = fx['labels']
    assert algo._labels == labels
    k = max(labels) + 1
    assert len(clusters) == k == algo._num_clusters and k >= 2
    for c, members in enumerate(clusters):
        assert members.tolist() == [i for i, l in enumerate(labels) if l == c]
    noise = {i for i, l in enumerate(labels) if l < 0}
    assert noise and not (noise & {int(i) for m in clusters for i in m})
    want = sorted({(min(labels[i], labels[i + 1]), max(labels[i], labels[i + 1])) for i in range(n - 1)
                   if labels[i] != labels[i + 1] and labels[i] >= 0 and labels[i + 1] >= 0})
    assert [tuple(p.tolist()) for p in algo.neigboring_clusters] == want
    algo.run(graph, 3, True)
    assert algo._labels == [-1] * 3 + labels
    cfg['rmp']['intra_cluster_sampling']['enabled'] = True
    with pytest.raises(NotImplementedError, match='condensed tree'):
        rmp.get_clustering_algorithm('hdbscan', cfg).run(graph)


def test_two_contexts_with_different_precisions_from_two_threads_through_the_abi():
    """"no global mutable state, re-entrant".  Precision and kernel-selection flags travel IN every argument
    struct (hgn_mlp_fwd_t / hgn_mlp_bwd_t / hgn_wtask_t: products, flags; hgn_linear_*6: a parameter); what a launch depends on
    besides its arguments lives in a per-model ops.Context (deferred weight-gradient queue, pack epoch, workspaces).  Two threads
    drive two contexts with different precisions through the ABI's validation path at the same time (no GPU: every call stops at
    argument validation, AFTER the per-call options were read): each sees its own products / flags, its own error text, and the
    process default is untouched."""
    import threading
    from hgn_amd import _lib, ops
    lib = _lib.lib()
    assert lib.hgn_get_matmul_products() == 3
    results, errors = {}, []

    def drive(name, ctx, want_products, n=300):
        try:
            with ops.using(ctx):
                for i in range(n):
                    c = ops.current()
                    assert c is ctx
                    a = _lib.MlpFwd()
                    c.stamp(a)
                    assert (a.products, a.flags) == (want_products, ctx.flags())
                    a.M = 5
                    a.out_w = 500                                   # invalid on purpose: the call answers with a status code
                    assert lib.hgn_mlp_fwd(C.byref(a), None) == -1 and b'out_w' in lib.hgn_last_error()      # thread-local text
                    a.products = 4                                  # not a mode: refused before anything else
                    assert lib.hgn_mlp_fwd(C.byref(a), None) == -1 and b'products' in lib.hgn_last_error()
                    t = _lib.WTask()
                    c.stamp(t)
                    assert t.products == want_products
                    b = _lib.MlpBwd()
                    c.stamp(b)
                    b.M = 4; b.products = 7
                    assert lib.hgn_mlp_bwd(C.byref(b), None) == -1 and b'products' in lib.hgn_last_error()
                    assert lib.hgn_linear_fwd6(None, 128, 4, None, 2, None, 128, 9, None) == -1
                    c.invalidate_packs()
            results[name] = (ctx.pack_epoch, ctx.products())
        except Exception as ex:            # noqa: BLE001 -- reported by the main thread
            errors.append((name, repr(ex)))

    a_ctx, b_ctx = ops.Context(precision='fp16'), ops.Context(precision='bf16', fp32_mfma=False, general_fwd=True)
    ts = [threading.Thread(target=drive, args=('a', a_ctx, 2)), threading.Thread(target=drive, args=('b', b_ctx, 1))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert results == {'a': (300, 2), 'b': (300, 1)}
    assert ops.current() is ops.default_context() and ops.default_context().products() == 3 and lib.hgn_get_matmul_products() == 3
    assert b_ctx.flags() == _lib.F_GENERAL_FWD and a_ctx.flags() == ops.default_context().flags()
    import hgn_amd
    m1 = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges'])
    m2 = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges'])
    assert m1._hgn_ctx is not m2._hgn_ctx and m1._hgn_ctx.wq is not m2._hgn_ctx.wq
    m2.set_matmul_precision('bf16')
    assert (m1._hgn_ctx.products(), m2._hgn_ctx.products()) == (3, 1)


def test_topology_parts_of_a_hierarchical_graph():
    """topology.EdgeTopology.parts: which node part (0 mesh rows, 1 hyper rows) the senders / the receivers of a set lie in, from the
    lowest / highest index of each side; None when the set is empty or a side straddles the split (the caller then concatenates)."""
    from hgn_amd import topology
    t = object.__new__(topology.EdgeTopology)
    for span, n_mesh, want in (((0, 99, 0, 99), 100, (0, 0)), ((0, 99, 100, 115), 100, (0, 1)), ((100, 115, 100, 115), 100, (1, 1)),
                               ((100, 115, 3, 99), 100, (1, 0)), ((0, 100, 0, 99), 100, None), ((0, 99, 99, 100), 100, None),
                               (None, 100, None)):
        t.span = span
        assert t.parts(n_mesh) == want, (span, want)
