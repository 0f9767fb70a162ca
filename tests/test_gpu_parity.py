"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the reference-generated goldens.

Tolerance (BASELINE.json north_star): 1e-5 relative fp32.  The metric is max|a-b| / max|b| per tensor (helpers.rel_err),
measured against the fp64 oracle where the oracle is the checker, and against the reference's own fp32 outputs for
the golden fixtures (there the reference's own fp32 rounding is part of the difference, hence 2e-5).
"""
import glob
import os

import pytest
import torch

from oracle import mgn_oracle as O
from tests import helpers as H
from tests import synth

pytestmark = pytest.mark.gpu

TOL_OUT = 1e-5
TOL_GRAD = 2e-5


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    import hgn_amd
    from hgn_amd import _lib
    _lib.lib()          # the HIP extension must be the thing that runs: fail loudly if it is not built
    yield


# ---------------------------------------------------------------------------------------------------------------
# a2: segment reductions
# ---------------------------------------------------------------------------------------------------------------
def test_g1_segment_ops_golden():
    import hgn_amd
    g1 = torch.load(os.path.join(H.GOLDEN, 'g1_segment_ops.pt'))
    for key, ref in g1['out'].items():
        op, nm = key.split('_')
        x = (g1['data2'] if nm == '2d' else g1['data1']).clone().cuda().requires_grad_(True)
        for ids in (g1['ids'].cuda(), g1['ids'].clone()):          # ids on the device or left on the host
            x.grad = None
            y = hgn_amd.unsorted_segment_operation(x, ids, g1['num_segments'], op)
            assert y.dtype == x.dtype and tuple(y.shape) == tuple(ref['y'].shape)
            assert H.rel_err(y, ref['y']) <= 2e-6, key
            (y * ref['w'].cuda()).sum().backward()
            assert H.rel_err(x.grad, ref['gx']) <= 2e-6, key
    # the adversarial cases of G1 against the BRUTE-FORCE loops stored with them ('bf': oracle/scatter_loops.py at generation
    # time, independent of the import stand-in and of the oracle): values exact for max / min (selection, not arithmetic),
    # gradient routed to the FIRST of equal values, empty segments 0
    for cname, rec in g1['adversarial'].items():
        for op in ('sum', 'mean', 'max', 'min'):
            x = rec['data'].clone().cuda().requires_grad_(True)
            y = hgn_amd.unsorted_segment_operation(x, rec['ids'].cuda(), rec['num_segments'], op)
            bf, w = rec['bf'][op], rec['ref'][op]['w']
            (y * w.cuda()).sum().backward()
            if op in ('max', 'min'):
                assert torch.equal(y.cpu(), bf['y']), (cname, op)
                assert torch.equal(x.grad.cpu().double(), bf['gx']), (cname, op)
            else:
                assert H.rel_err(y, bf['y']) <= 2e-6 and H.rel_err(x.grad, bf['gx']) <= 2e-6, (cname, op)
    with pytest.raises(Exception, match='Invalid operation type'):
        hgn_amd.unsorted_segment_operation(g1['data2'].cuda(), g1['ids'].cuda(), g1['num_segments'], 'median')
    with pytest.raises(IndexError):
        hgn_amd.unsorted_segment_operation(g1['data2'].cuda(), (g1['ids'] + 100).cuda(), g1['num_segments'], 'sum')


def test_g1_segment_std_golden():
    """The fifth operation unsorted_segment_operation accepts, 'std' (src/util.py:129-130 -> torch_scatter.scatter_std, unbiased;
    unreachable from the reference's configs): HIP kernels (hgn_segment_std_fwd / _bwd) against the reference's own function run
    through the import stand-in AND against the brute-force loops stored next to it (tests/golden/gen_golden_std.py): values to 2e-6,
    gradients to 2e-5 on the rows of segments with at least two elements; a one-element segment has no variance: its gradient is NaN
    here as in the wheel's composite (sqrt'(0) * 0), an empty segment gives 0."""
    import hgn_amd
    fx = torch.load(os.path.join(H.GOLDEN, 'g1_segment_std.pt'))
    assert set(fx) == {'wide', 'one_d', 'narrow_unsorted'}
    for name, rec in fx.items():
        for ids in (rec['ids'].cuda(), rec['ids'].clone()):
            x = rec['data'].clone().cuda().requires_grad_(True)
            y = hgn_amd.unsorted_segment_operation(x, ids, rec['num_segments'], 'std')
            assert y.dtype == x.dtype and tuple(y.shape) == tuple(rec['ref_y'].shape)
            assert H.rel_err(y, rec['ref_y']) <= 2e-6 and H.rel_err(y, rec['bf_y']) <= 2e-6, name
            empty = (rec['count'] == 0).nonzero().flatten()
            assert empty.numel() == 0 or float(y.detach()[empty.cuda()].abs().max()) == 0.0
            (y * rec['w'].cuda()).sum().backward()
            multi = (rec['count'][rec['ids']] >= 2)
            g = x.grad.cpu()
            assert H.rel_err(g[multi], rec['ref_gx'][multi]) <= 2e-5 and H.rel_err(g[multi].double(), rec['bf_gx'][multi]) <= 2e-5, name
            single = ~multi
            if single.any():
                assert torch.isnan(g[single]).all() and torch.isnan(rec['ref_gx'][single]).all(), name


@pytest.mark.parametrize('E,N,D', [(0, 5, 128), (1, 1, 128), (1000, 37, 128), (5000, 4000, 128), (777, 50, 3), (64, 9, 1)])
def test_segment_ops_vs_oracle(E, N, D):
    import hgn_amd
    gen = torch.Generator().manual_seed(E + N)
    ids = torch.randint(0, N, (E,), generator=gen)
    data = torch.randn(E, D, generator=gen)
    if E > 10:
        data[5] = data[2]; ids[5] = ids[2]                       # exact tie inside one segment
    for op in ('sum', 'mean', 'max', 'min'):
        xo = data.clone().double().requires_grad_(True)
        yo = O.segment_reduce(xo, ids, N, op)
        w = torch.randn(yo.shape, generator=gen, dtype=torch.float64)
        (yo * w).sum().backward()
        x = data.clone().cuda().requires_grad_(True)
        y = hgn_amd.unsorted_segment_operation(x, ids.cuda(), N, op)
        (y * w.float().cuda()).sum().backward()
        assert H.rel_err(y, yo) <= 1e-6, op
        if E:
            assert H.rel_err(x.grad, xo.grad) <= 1e-6, op


def test_segment_sum_full_size_properties():
    """BASELINE full size (21 flag graphs: 195k edges x 128): linearity and total-mass conservation, exact ordering
    independence of max/min, against no oracle (too slow in fp64 on the host at this size is fine, but properties
    are size independent)."""
    import hgn_amd
    g = synth.batch([synth.grid_graph(seed=s) for s in range(21)])
    es = g.edge_sets[0]
    N = g.node_features[0].shape[0]
    E = es.receivers.shape[0]
    gen = torch.Generator().manual_seed(0)
    a = torch.randn(E, 128, generator=gen).cuda()
    b = torch.randn(E, 128, generator=gen).cuda()
    ids = es.receivers.cuda()
    sa = hgn_amd.unsorted_segment_operation(a, ids, N, 'sum')
    sb = hgn_amd.unsorted_segment_operation(b, ids, N, 'sum')
    sab = hgn_amd.unsorted_segment_operation(2.0 * a - b, ids, N, 'sum')
    assert H.rel_err(sab, 2.0 * sa - sb) <= 1e-5
    assert H.rel_err(sa.double().sum(0), a.double().sum(0)) <= 1e-6
    perm = torch.randperm(E, generator=gen).cuda()
    mx1 = hgn_amd.unsorted_segment_operation(a, ids, N, 'max')
    mx2 = hgn_amd.unsorted_segment_operation(a[perm], ids[perm], N, 'max')
    assert torch.equal(mx1, mx2)
    cnt = torch.bincount(ids, minlength=N).clamp(min=1).unsqueeze(1)
    mean = hgn_amd.unsorted_segment_operation(a, ids, N, 'mean')
    assert H.rel_err(mean * cnt, sa) <= 1e-5


# ---------------------------------------------------------------------------------------------------------------
# a1 / a3 / a5: fused MLP kernels at operator level
# ---------------------------------------------------------------------------------------------------------------
def _mlp_sd(in_w, out_w, ln, seed):
    shapes = O.collections.OrderedDict()
    base = 'm.0.layers.' if ln else 'm.layers.'
    for i, (a, b) in enumerate(((in_w, 128), (128, 128), (128, out_w))):
        shapes[f'{base}linear_{i}.weight'] = (b, a)
        shapes[f'{base}linear_{i}.bias'] = (b,)
    if ln:
        shapes['m.1.weight'] = (out_w,)
        shapes['m.1.bias'] = (out_w,)
    return O.init_state_dict_like(shapes, seed)


def _weights(sd, ln):
    from hgn_amd import ops
    base = 'm.0.layers.' if ln else 'm.layers.'
    ts = [sd[f'{base}linear_{i}.{p}'].cuda().requires_grad_(True) for i in range(3) for p in ('weight', 'bias')]
    if ln:
        ts += [sd['m.1.weight'].cuda().requires_grad_(True), sd['m.1.bias'].cuda().requires_grad_(True)]
    return ops.MLPWeights(*ts), ts


@pytest.mark.parametrize('widths', [[128], [128, 128], [7]], ids=['latent', 'node2src', 'encoder7'])
def test_latency_form_forward_equals_the_throughput_kernels_bit_for_bit(widths):
    """Launches of at most 256 tiles (one workgroup per CU: a rollout step, every small case of this file) take the latency form of
    the forward -- loader waves streaming the packed weights through an LDS ring, csrc/mlp6_device.h: lat_loader / gemm6_lat --,
    bigger ones the 64-row (<= 98 303 rows) or 128-row workgroups with staged weights.  Rows are independent, the products and
    their order per accumulator are the same in all three: the first rows of a big launch must equal a small launch on those
    rows BIT FOR BIT, saved activations and ReLU sign words included (what the backward kernels read)."""
    from hgn_amd import ops
    gen = torch.Generator().manual_seed(11 + len(widths))
    sd = _mlp_sd(sum(widths), 128, True, seed=5)
    w, _ = _weights(sd, True)
    small, mid, big = 1000, 20000, 99000                       # 16 / 313 / 774 (x 128 rows) tiles
    srcs = [torch.randn(big, wd, generator=gen).cuda() for wd in widths]
    res = 0 if widths[0] == 128 else -1
    outs = {}
    for M in (small, mid, big):
        use = [x[:M].clone().requires_grad_(True) for x in srcs]
        y = ops.fused_mlp(use, w, [None] * len(use), res)
        saved = [(t, t.shape[0] // M) for t in y.grad_fn.saves if torch.is_tensor(t) and t.dim() >= 1 and t.shape[0] >= M and t.shape[0] % M == 0]
        outs[M] = (y.detach(), saved)
    for M in (mid, big):
        assert torch.equal(outs[M][0][:small], outs[small][0]), M
        assert len(outs[M][1]) == len(outs[small][1]) and len(outs[small][1]) >= 4     # z1, z2, x-hat, rstd / sign words
        for (a, ka), (b, kb) in zip(outs[M][1], outs[small][1]):
            assert ka == kb and torch.equal(a[:small * ka], b[:small * kb]), (M, a.shape)


@pytest.mark.parametrize('agg', [('sum',), ('sum', 'mean', 'max', 'min'), None], ids=['sum_in_kernel', 'pna', 'no_aggregate'])
def test_training_edge_forward_kernel_equals_the_general_kernel_bit_for_bit(agg):
    """Training edge blocks of >= 98 304 rows run mlp6_fwd_edge_kernel (csrc/mlp6.hip): the general 128-row kernel with everything it
    decides at run time decided at launch (32-bit row offsets from scalar bases, no per-row store tests except in the launch's last
    tile, three-instruction ReLU + sign words).  Same products, same order, same row sums: outputs, in-kernel segment sums, every
    saved tensor the backward reads (z1, z2, x-hat, 1/sigma, sign words) and therefore every gradient must equal the general
    kernel's BIT FOR BIT -- on a row count that is not a multiple of the tile (the last workgroup takes the tested store path)."""
    from hgn_amd import ops, topology, modules
    import hgn_amd
    g = synth.batch([synth.grid_graph(seed=i % 3, nx=40, ny=40) for i in range(11)])
    es = g.edge_sets[0]
    N, E = g.node_features[0].shape[0], es.senders.shape[0]
    assert E >= 98304 and E % 128 != 0
    topo = topology.EdgeTopology(es.senders, es.receivers, N, torch.device('cuda'))
    torch.manual_seed(3)
    m = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges']).cuda()
    m(hgn_amd.MultiGraph([x.cuda() for x in g.node_features], [hgn_amd.EdgeSet(es.name, es.features.cuda(), es.senders, es.receivers)]))
    w = modules.weights_of(m.processor.graphnet_blocks[0].edge_models['mesh_edges'], 384)
    h0, e0 = torch.randn(N, 128, device='cuda'), torch.randn(E, 128, device='cuda')
    wsum = torch.randn(E, 128, device='cuda')

    def run():
        h, e = h0.clone().requires_grad_(True), e0.clone().requires_grad_(True)
        for p_ in w.tensors():
            p_.grad = None
        out = ops.edge_block(h, e, topo, w, agg)
        y, a_ = out if agg is not None else (out, None)
        saves = [t.clone() for t in y.grad_fn.saves if torch.is_tensor(t)]
        loss = (y * wsum).sum() + (a_.sum() if a_ is not None else 0.0)
        loss.backward()
        return [y.detach(), a_.detach() if a_ is not None else None] + saves + [h.grad, e.grad] + [p_.grad.clone() for p_ in w.tensors()]

    fast = run()
    with ops.using(ops.Context(general_fwd=True)):              # hgn_mlp_fwd_t.flags: HGN_F_GENERAL_FWD for these calls only
        general = run()
    assert len(fast) == len(general) and len(fast) >= 2 + 5 + 2 + 8
    for i, (a_, b_) in enumerate(zip(fast, general)):
        assert (a_ is None) == (b_ is None), i
        if a_ is not None:
            assert a_.shape == b_.shape and torch.equal(a_, b_), (i, a_.shape)


@pytest.mark.parametrize('agg', ['sum', 'pna'])
def test_inference_forward_of_the_whole_model_equals_the_training_forward_bit_for_bit(agg):
    """Without gradients a stack of plain GraphNet blocks runs its small launches in the column-split form, and every node kernel also
    forms the NEXT block's pre-projection (and zero-fills its aggregate buffer) in the same launch (hgn_mlp_fwd_t.post_*,
    modules.Processor.forward); with gradients every block launches its own pre-projection and saves activations.  Same
    arithmetic in the same order: the network outputs are equal bit for bit."""
    import hgn_amd
    graph = synth.grid_graph(seed=5, nx=13, ny=9)
    shapes = O.param_shapes('none', agg, 4, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=21)
    model = H.hip_model('none', agg, 4, ['mesh_edges'], sd)
    G = hgn_amd.MultiGraph([x.cuda() for x in graph.node_features],
                           [hgn_amd.EdgeSet(e.name, e.features.cuda(), e.senders.cuda(), e.receivers.cuda()) for e in graph.edge_sets])
    y_train = model(G).detach()
    with torch.no_grad():
        y_inf = model(G)
    assert torch.equal(y_inf, y_train)
    out_o, _, _, _ = H.oracle_run(sd, graph, 'none', agg, torch.zeros(13 * 9, 3), torch.ones(13 * 9, dtype=torch.bool))
    assert H.rel_err(y_inf, out_o) <= TOL_OUT


def test_pre_projection_forms_agree_bit_for_bit():
    """hgn_linear_fwd6 (h -> [h W1s^T | h W1r^T], the node-level half of the split first edge layer) has three forms by row count:
    column-split latency form (<= 4 096 rows), row-per-wave latency form (<= 256 tiles), staged weights.  Same products, same
    order: the first rows of a bigger launch equal a smaller launch on those rows bit for bit."""
    import ctypes as C
    from hgn_amd import ops, _lib
    sd = _mlp_sd(384, 128, True, seed=9)
    w, _ = _weights(sd, True)
    pk = ops.packs_of(w)
    L = _lib.lib()
    pb = (C.c_void_p * 2)(pk.data_ptr(), pk.data_ptr() + _lib.PACK_BLOCK_BYTES)
    h = torch.randn(20000, 128, generator=torch.Generator().manual_seed(4)).cuda()
    outs = {}
    for N in (1000, 4096, 9000, 20000):
        P = torch.empty(N, 256, device='cuda')
        _lib.check(L.hgn_linear_fwd6(h.data_ptr(), 128, N, pb, 2, P.data_ptr(), 256, 0, _lib.stream_ptr()), 'hgn_linear_fwd6')
        outs[N] = P
    for N in (4096, 9000, 20000):
        assert torch.equal(outs[N][:1000], outs[1000]), N
    want = h.double() @ sd['m.0.layers.linear_0.weight'].double().cuda()[:, :128].t()
    assert H.rel_err(outs[20000][:, :128], want) <= 2e-6


@pytest.mark.parametrize('accumulate', [0, 1])
def test_pre_projection_backward_forms_agree_bit_for_bit(accumulate):
    """hgn_linear_bwd6a (dh (+)= dP_s W1s + dP_r W1r) by row count: column-split latency form (<= 4 096 rows), row-per-wave latency
    form, staged weights -- also when the accumulators start from the rows of dh (row scale capped by what they hold), and with
    gradient rows 1e-20 apart in magnitude."""
    import ctypes as C
    from hgn_amd import ops, _lib
    sd = _mlp_sd(384, 128, True, seed=9)
    w, _ = _weights(sd, True)
    pk_t = ops.packs_of(w, transposed=True)
    L = _lib.lib()
    pb = (C.c_void_p * 2)(pk_t.data_ptr(), pk_t.data_ptr() + _lib.PACK_BLOCK_BYTES)
    gen = torch.Generator().manual_seed(6)
    g = torch.randn(20000, 256, generator=gen)
    g[5:9] *= 1e-20; g[11] = 0
    g = g.cuda()
    base = torch.randn(20000, 128, generator=gen).cuda()
    outs = {}
    for N in (1000, 4096, 9000, 20000):
        dx = base[:N].clone()
        _lib.check(L.hgn_linear_bwd6a(g.data_ptr(), 256, N, pb, 2, dx.data_ptr(), 128, accumulate, 0, _lib.stream_ptr()), 'hgn_linear_bwd6a')
        outs[N] = dx
    for N in (4096, 9000, 20000):
        assert torch.equal(outs[N][:1000], outs[1000]), N
    W = sd['m.0.layers.linear_0.weight'].double().cuda()
    want = g[:, :128].double() @ W[:, :128] + g[:, 128:].double() @ W[:, 128:256] + (base.double() if accumulate else 0)
    assert H.rel_err(outs[20000], want) <= 2e-6


@pytest.mark.parametrize('M', [1, 17, 333, 1600, 4096, 4100])
@pytest.mark.parametrize('case', ['encoder7', 'encoder_idx', 'node2src', 'node_pna', 'latent_res'])
def test_inference_forward_column_split_form_equals_training_forward_bit_for_bit(M, case):
    """Launches of at most 4 096 rows take the column-split latency form (csrc/mlp6.hip: mlp6_fwd_cs_kernel -- four waves share 16
    rows, each owning 32 output columns, operand vectors exchanged through LDS, LayerNorm on the gathered tile), with gradients
    (activations saved) and without: the outputs must be equal bit for bit (4 100 rows: both sides run the row-per-wave form).
    Column-split against row-per-wave: test_training_forward_column_split_form_equals_the_row_per_wave_kernels_bit_for_bit."""
    from hgn_amd import ops
    gen = torch.Generator().manual_seed(M * 3 + len(case))
    residual, idx = -1, None
    if case == 'encoder7':
        widths = [7]
    elif case == 'encoder_idx':
        widths = [128]
        idx = torch.randint(0, 50, (M,), generator=gen)
    elif case == 'node2src':
        widths, residual = [128, 128], 0
    elif case == 'node_pna':
        widths, residual = [128, 512], 0
    else:
        widths, residual = [128], 0
    sd = _mlp_sd(sum(widths), 128, True, seed=3)
    w, _ = _weights(sd, True)
    rows0 = 50 if idx is not None else M
    srcs = [torch.randn(rows0 if i == 0 else M, wd, generator=gen).cuda() for i, wd in enumerate(widths)]
    idxs = [idx.cuda().int() if idx is not None else None] + [None] * (len(srcs) - 1)
    if idx is not None:
        residual = -1
    y_train = ops.fused_mlp([x.clone().requires_grad_(True) for x in srcs], w, idxs, residual).detach()
    with torch.no_grad():
        y_inf = ops.fused_mlp(srcs, w, idxs, residual)
    assert torch.equal(y_inf, y_train)


@pytest.mark.parametrize('M', [1, 17, 333, 1600, 4096])
@pytest.mark.parametrize('case', ['encoder7', 'node2src', 'node_pna', 'latent_res'])
def test_training_forward_column_split_form_equals_the_row_per_wave_kernels_bit_for_bit(M, case):
    """Training launches of at most 4 096 rows take the column-split form too (one graph per step: a node update is 1 600 rows): every
    wave stores its own 32 columns of z1 / z2 / xhat and its byte of the ReLU sign words.  The same M rows as the head of a launch of
    M + 4 100 rows run the row-per-wave kernels: outputs, every saved array and the data gradients must agree bit for bit."""
    from hgn_amd import ops
    gen = torch.Generator().manual_seed(M * 5 + len(case))
    residual = -1
    if case == 'encoder7':
        widths = [7]
    elif case == 'node2src':
        widths, residual = [128, 128], 0
    elif case == 'node_pna':
        widths, residual = [128, 512], 0
    else:
        widths, residual = [128], 0
    sd = _mlp_sd(sum(widths), 128, True, seed=3)
    w, _ = _weights(sd, True)
    big = M + 4100
    srcs_big = [torch.randn(big, wd, generator=gen).cuda() for wd in widths]
    d_big = torch.randn(big, 128, generator=gen).cuda()

    def run(rows):
        xs = [x[:rows].clone().requires_grad_(True) for x in srcs_big]
        y = ops.fused_mlp(xs, w, [None] * len(xs), residual)
        saves = [t for t in y.grad_fn.saves]
        y.backward(d_big[:rows].clone())
        return y.detach(), saves, [x.grad for x in xs]
    y_s, sv_s, dx_s = run(M)
    y_b, sv_b, dx_b = run(big)
    assert torch.equal(y_s, y_b[:M])
    for name, a_, b_ in zip(('z1', 'z2', 'xhat', 'rstd', 'relu bits'), sv_s, sv_b):
        assert torch.equal(a_, b_[:M]), name
    for a_, b_ in zip(dx_s, dx_b):
        assert torch.equal(a_, b_[:M])


@pytest.mark.parametrize('M', [1, 31, 128, 333])
@pytest.mark.parametrize('case', ['encoder7', 'encoder_idx', 'node2src', 'node_pna', 'decoder3', 'latent_res'])
def test_fused_mlp_vs_oracle(M, case):
    from hgn_amd import ops
    gen = torch.Generator().manual_seed(M * 7 + len(case))
    ln, out_w, residual, idx = True, 128, -1, None
    if case == 'encoder7':
        widths = [7]
    elif case == 'encoder_idx':
        widths = [8]
        idx = torch.randperm(M, generator=gen)
    elif case == 'node2src':
        widths, residual = [128, 128], 0
    elif case == 'node_pna':
        widths, residual = [128, 1024], 0
    elif case == 'decoder3':
        widths, ln, out_w = [128], False, 3
    else:
        widths, residual = [128], 0
    sd = _mlp_sd(sum(widths), out_w, ln, seed=M)
    srcs = [torch.randn(M, wd, generator=gen) for wd in widths]
    # oracle (fp64)
    sdo = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    so = [s.double().requires_grad_(True) for s in srcs]
    xin = [so[0][idx] if idx is not None else so[0]] + so[1:]
    yo = O.mlp(sdo, 'm', torch.cat(xin, -1), layer_norm=ln)
    if residual >= 0:
        yo = yo + so[residual]
    w_out = torch.randn(yo.shape, generator=gen, dtype=torch.float64)
    (yo * w_out).sum().backward()
    # HIP
    w, wts = _weights(sd, ln)
    sh = [s.cuda().requires_grad_(True) for s in srcs]
    if case == 'node_pna':                 # second source as a row-slice view of a wider, taller buffer
        big = torch.zeros(M + 5, 1024, device='cuda')
        big[:M] = srcs[1].cuda()
        big.requires_grad_(True)
        use = [sh[0], big[:M]]
    else:
        use = sh
    idxs = [idx.cuda().int() if idx is not None else None] + [None] * (len(use) - 1)
    y = ops.fused_mlp(use, w, idxs, residual)
    (y * w_out.float().cuda()).sum().backward()
    assert H.rel_err(y, yo) <= TOL_OUT
    names = [f"{'m.0.layers.' if ln else 'm.layers.'}linear_{i}.{p}" for i in range(3) for p in ('weight', 'bias')]
    if ln:
        names += ['m.1.weight', 'm.1.bias']
    for n, t in zip(names, wts):
        assert H.rel_err(t.grad, sdo[n].grad) <= TOL_GRAD, n
    assert H.rel_err(sh[0].grad, so[0].grad) <= TOL_GRAD
    if case == 'node_pna':
        assert H.rel_err(big.grad[:M], so[1].grad) <= TOL_GRAD
        assert float(big.grad[M:].abs().max()) == 0.0
    elif len(sh) > 1:
        assert H.rel_err(sh[1].grad, so[1].grad) <= TOL_GRAD


@pytest.mark.parametrize('nx,ny', [(2, 2), (7, 5), (40, 40)])
def test_edge_block_vs_oracle(nx, ny):
    from hgn_amd import ops, topology
    g = synth.grid_graph(seed=nx, nx=nx, ny=ny)
    es = g.edge_sets[0]
    N, E = nx * ny, es.senders.shape[0]
    gen = torch.Generator().manual_seed(3)
    h = torch.randn(N, 128, generator=gen)
    e = torch.randn(E, 128, generator=gen)
    sd = _mlp_sd(384, 128, True, seed=nx)
    sdo = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    ho, eo = h.double().requires_grad_(True), e.double().requires_grad_(True)
    yo = O.update_edge_features(sdo, 'm', [ho], O.EdgeSet('x', eo, es.senders, es.receivers))
    w_out = torch.randn(yo.shape, generator=gen, dtype=torch.float64)
    (yo * w_out).sum().backward()
    topo = topology.EdgeTopology(es.senders, es.receivers, N, torch.device('cuda'))
    perm = topo.r.perm.long()
    # CSR invariants
    assert torch.equal(topo.rcv.long().cpu(), es.receivers[perm.cpu()])
    assert torch.equal(topo.snd.long().cpu(), es.senders[perm.cpu()])
    assert bool((topo.rcv[1:] >= topo.rcv[:-1]).all())
    w, wts = _weights(sd, True)
    hh = h.cuda().requires_grad_(True)
    ee = e.cuda()[perm].requires_grad_(True)
    y = ops.edge_block(hh, ee, topo, w)
    (y * w_out.float().cuda()[perm]).sum().backward()
    assert H.rel_err(y, yo[perm.cpu()]) <= TOL_OUT
    assert H.rel_err(hh.grad, ho.grad) <= TOL_GRAD
    assert H.rel_err(ee.grad, eo.grad[perm.cpu()]) <= TOL_GRAD
    names = [f'm.0.layers.linear_{i}.{p}' for i in range(3) for p in ('weight', 'bias')] + ['m.1.weight', 'm.1.bias']
    for n, t in zip(names, wts):
        assert H.rel_err(t.grad, sdo[n].grad) <= TOL_GRAD, n


@pytest.mark.parametrize('with_base', [True, False])
@pytest.mark.parametrize('N,max_deg,seed', [(60, 70, 0), (1237, 9, 1), (3, 5, 2)])
def test_segment_reduce_bwd_sorted_equals_the_edge_parallel_kernel_bit_for_bit(N, max_deg, seed, with_base):
    """include/hgn_mp.h: hgn_segment_reduce_bwd_sorted -- the aggregation backward of the four pna aggregates (graphnet.py:50-70 under autograd)
    for rows in receiver order, one half-wave per RECEIVER: the same bits as hgn_segment_reduce_bwd (one half-wave per row), with and without
    the d(e') base, ragged and empty segments, ties among the winners."""
    import ctypes as C
    from hgn_amd import _lib, topology
    gen = torch.Generator().manual_seed(seed)
    deg = torch.randint(0, max_deg + 1, (N,), generator=gen)
    receivers = torch.repeat_interleave(torch.arange(N), deg)
    E = receivers.shape[0]
    senders = torch.randint(0, N, (E,), generator=gen)
    topo = topology.EdgeTopology(senders, receivers, N, torch.device('cuda'))
    data = torch.randint(-3, 4, (E, 128), generator=gen).float().cuda()              # many ties
    L, st = _lib.lib(), _lib.stream_ptr()
    ops = (C.c_int32 * 4)(0, 1, 2, 3)
    agg = torch.empty(N, 512, device='cuda')
    amax = torch.empty(N, 128, dtype=torch.int32, device='cuda'); amin = torch.empty_like(amax)
    _lib.check(L.hgn_segment_reduce_fwd(data.data_ptr(), 128, 128, None, topo.r.rowptr.data_ptr(), N, ops, 4, agg.data_ptr(), 512,
                                        amax.data_ptr(), amin.data_ptr(), st), 'fwd')
    d_agg = torch.randn(N, 512, generator=gen).cuda()
    base = torch.randn(E, 128, generator=gen).cuda() if with_base else None
    ref = torch.full((E, 128), float('nan'), device='cuda'); got = torch.full((E, 128), float('nan'), device='cuda')
    bp = base.data_ptr() if base is not None else None
    _lib.check(L.hgn_segment_reduce_bwd(d_agg.data_ptr(), 512, 128, None, topo.rcv.data_ptr(), topo.r.rowptr.data_ptr(), E, ops, 4,
                                        amax.data_ptr(), amin.data_ptr(), bp, ref.data_ptr(), 128, st), 'edge-parallel')
    _lib.check(L.hgn_segment_reduce_bwd_sorted(d_agg.data_ptr(), 512, topo.r.rowptr.data_ptr(), N, ops, 4, amax.data_ptr(), amin.data_ptr(), bp,
                                               got.data_ptr(), 128, st), 'receiver-parallel')
    assert torch.equal(got, ref)
    assert L.hgn_segment_reduce_bwd_sorted(d_agg.data_ptr(), 512, topo.r.rowptr.data_ptr(), N, ops, 4, None, amin.data_ptr(), bp,
                                           got.data_ptr(), 128, st) != 0                     # max without its arg rows: refused


@pytest.mark.parametrize('N,max_deg,seed', [(50, 70, 0), (1237, 9, 1), (3, 5, 2), (20000, 12, 3)])
def test_segment_sum_pair_equals_two_segment_reduce_launches_bit_for_bit(N, max_deg, seed):
    """include/hgn_mp.h: hgn_segment_sum_pair -- the receiver sums and the sender sums of dz1 (graphnet.py:22-32 backward) in one pass over
    the rows: same additions in the same order as two hgn_segment_reduce_fwd launches, ragged and empty segments, rows past a multiple
    of the workgroup size."""
    import ctypes as C
    from hgn_amd import _lib, topology
    gen = torch.Generator().manual_seed(seed)
    deg = torch.randint(0, max_deg + 1, (N,), generator=gen)
    receivers = torch.repeat_interleave(torch.arange(N), deg)
    E = receivers.shape[0]
    receivers = receivers[torch.randperm(E, generator=gen)]
    senders = torch.randint(0, N, (E,), generator=gen)
    topo = topology.EdgeTopology(senders, receivers, N, torch.device('cuda'))
    data = torch.randn(E, 128, generator=gen).cuda()
    L, st = _lib.lib(), _lib.stream_ptr()
    ops = (C.c_int32 * 1)(0)
    ref = torch.empty(N, 256, device='cuda')
    _lib.check(L.hgn_segment_reduce_fwd(data.data_ptr(), 128, 128, topo.s.perm.data_ptr(), topo.s.rowptr.data_ptr(), N, ops, 1,
                                        ref.data_ptr(), 256, None, None, st), 'senders')
    _lib.check(L.hgn_segment_reduce_fwd(data.data_ptr(), 128, 128, None, topo.r.rowptr.data_ptr(), N, ops, 1,
                                        ref.data_ptr() + 512, 256, None, None, st), 'receivers')
    got = torch.full((N, 256), float('nan'), device='cuda')
    _lib.check(L.hgn_segment_sum_pair(data.data_ptr(), 128, topo.r.rowptr.data_ptr(), topo.s.perm.data_ptr(), topo.s.rowptr.data_ptr(), N,
                                      got.data_ptr() + 512, 256, got.data_ptr(), 256, st), 'pair')
    assert torch.equal(got, ref)
    # against the definition
    srt_rcv = topo.rcv.cpu().long()
    want = torch.zeros(N, 128, dtype=torch.float64).index_add_(0, srt_rcv, data.cpu().double())
    assert H.rel_err(got[:, 128:], want) <= 1e-6
    assert L.hgn_segment_sum_pair(data.data_ptr(), 100, topo.r.rowptr.data_ptr(), topo.s.perm.data_ptr(), topo.s.rowptr.data_ptr(), N,
                                  got.data_ptr() + 512, 256, got.data_ptr(), 256, st) != 0            # rows narrower than 128: refused


@pytest.mark.parametrize('kind', ['mesh_to_hyper', 'hyper_to_mesh', 'hyper_to_hyper', 'mesh_to_mesh'])
@pytest.mark.parametrize('agg', [('sum',), ('sum', 'mean', 'max', 'min')])
def test_edge_block_by_node_part_equals_the_concatenated_form_bit_for_bit(kind, agg):
    """Hierarchical graphs keep mesh rows and hyper rows in two tensors (modules.GraphNet._edge): an edge set whose senders lie in
    one part and whose receivers lie in one part hands over those parts only (ops.edge_block: parts, h_r).  Same bits as the one
    concatenated tensor the reference indexes (graphnet.py:25-26, hypergraphnet.py:21-54): outputs, the receiver part's rows of the
    aggregate (the other rows of the full aggregate are zero), every gradient -- and the untouched part gets NO gradient where the
    concatenated form hands back zeros."""
    from hgn_amd import ops, topology
    gen = torch.Generator().manual_seed(11)
    n_mesh, n_hyper, E = 700, 37, 5000
    N = n_mesh + n_hyper
    lo = {'m': (0, n_mesh), 'h': (n_mesh, N)}
    sp, rp = {'mesh_to_hyper': 'mh', 'hyper_to_mesh': 'hm', 'hyper_to_hyper': 'hh', 'mesh_to_mesh': 'mm'}[kind]
    senders = torch.randint(*lo[sp], (E,), generator=gen)
    receivers = torch.randint(*lo[rp], (E,), generator=gen)
    topo = topology.EdgeTopology(senders, receivers, N, torch.device('cuda'))
    ps, pr = topo.parts(n_mesh)
    assert (ps, pr) == ('mh'.index(sp), 'mh'.index(rp))
    sd = _mlp_sd(384, 128, True, seed=3)
    w, wts = _weights(sd, True)
    hm0, hh0 = torch.randn(n_mesh, 128, generator=gen).cuda(), torch.randn(n_hyper, 128, generator=gen).cuda()
    e0 = torch.randn(E, 128, generator=gen).cuda()
    k = len(agg)
    w_y, w_a = torch.randn(E, 128, generator=gen).cuda(), torch.randn(N, k * 128, generator=gen).cuda()
    offs = (0, n_mesh)

    def run(by_part):
        for t in wts:
            t.grad = None
        hm, hh, e = hm0.clone().requires_grad_(True), hh0.clone().requires_grad_(True), e0.clone().requires_grad_(True)
        nodes = (hm, hh)
        if by_part:
            y, a = ops.edge_block(nodes[ps], e, topo, w, agg, parts=(offs[ps], offs[pr]), h_r=None if ps == pr else nodes[pr])
            assert a.shape[0] == nodes[pr].shape[0]
            wa = w_a[offs[pr]:offs[pr] + a.shape[0]]
        else:
            y, a = ops.edge_block(torch.cat(nodes), e, topo, w, agg)
            wa = w_a
        ((y * w_y).sum() + (a * wa).sum()).backward()
        return y.detach(), a.detach(), hm.grad, hh.grad, e.grad, [t.grad.clone() for t in wts]

    y1, a1, gm1, gh1, ge1, gw1 = run(True)
    y0, a0, gm0, gh0, ge0, gw0 = run(False)
    assert torch.equal(y1, y0)
    r0 = offs[pr]
    assert torch.equal(a1, a0[r0:r0 + a1.shape[0]])
    rest = torch.ones(N, dtype=torch.bool); rest[r0:r0 + a1.shape[0]] = False
    assert not bool(a0[rest.cuda()].any())
    assert torch.equal(ge1, ge0)
    for part, (g1, g0) in enumerate(((gm1, gm0), (gh1, gh0))):
        if part in (ps, pr):
            assert torch.equal(g1, g0)
        else:
            assert g1 is None and not bool(g0.any())
    # (dW1's node-row blocks are sums over the rows of the launch in chunks whose size follows the row count: same terms, other
    #  partial sums -- every other parameter gradient is formed over the same E edge rows)
    assert H.rel_err(gw1[0], gw0[0]) <= 1e-6 and torch.equal(gw1[0][:, 256:], gw0[0][:, 256:])
    for a, b in zip(gw1[1:], gw0[1:]):
        assert torch.equal(a, b)


def test_fused_mlp_with_left_out_input_blocks_equals_zero_sources():
    """ops.fused_mlp(cols=...): an input block that is zero for every row of the launch (the aggregate of an edge set that arrives in
    the other node part, heterographnet.py:17-33) is left out -- same output and source gradients as feeding zeros; its W1 columns
    get a zero gradient."""
    from hgn_amd import ops
    gen = torch.Generator().manual_seed(5)
    M = 333
    sd = _mlp_sd(128 + 512 + 128 + 512, 128, True, seed=9)
    w, wts = _weights(sd, True)
    x0 = [torch.randn(M, 128, generator=gen).cuda(), torch.randn(M, 128, generator=gen).cuda(), torch.randn(M, 512, generator=gen).cuda()]
    w_o = torch.randn(M, 128, generator=gen).cuda()

    def run(skip):
        for t in wts:
            t.grad = None
        h, a1, a3 = [t.clone().requires_grad_(True) for t in x0]
        if skip:      # columns: h 0..128 | (zeros) 128..640 | a1 640..768 | a3 768..1280
            out = ops.fused_mlp([h, a1, a3], w, residual=0, cols=(0, 640, 768))
        else:
            out = ops.fused_mlp([h, torch.zeros(M, 512, device='cuda'), a1, a3], w, residual=0)
        (out * w_o).sum().backward()
        return out.detach(), h.grad, a1.grad, a3.grad, [t.grad.clone() for t in wts]

    o1, gh1, ga1, gb1, gw1 = run(True)
    o0, gh0, ga0, gb0, gw0 = run(False)
    assert torch.equal(o1, o0) and torch.equal(gh1, gh0) and torch.equal(ga1, ga0) and torch.equal(gb1, gb0)
    for a, b in zip(gw1, gw0):
        assert torch.equal(a, b)
    assert not bool(gw1[0][:, 128:640].any())
    with pytest.raises(Exception):
        ops.fused_mlp([x0[0], x0[1]], w, residual=0, cols=(0, 64))          # overlapping the source before


@pytest.mark.parametrize('N,max_deg,seed', [(50, 65, 0), (700, 20, 1), (3, 65, 2), (1200, 9, 3), (300, 17, 4), (5, 3, 5), (200000, 12, 6)])
def test_edge_block_fused_segment_sums(N, max_deg, seed):
    """The `sum` aggregation of e' and the receiver half of the pre-projection gradient come out of the edge kernels themselves
    (include/hgn_mp.h: seg_out / seg_dz1).  Ragged degrees 0..max_deg (segments crossing 64-row tile ends, empty segments, a
    partial last tile): equal to the separate segment-reduce path within fp32 summation-order noise, and bit-reproducible."""
    from hgn_amd import ops, topology
    gen = torch.Generator().manual_seed(seed)
    deg = torch.randint(0, max_deg + 1, (N,), generator=gen)
    deg[torch.randint(0, N, (1,), generator=gen)] = max_deg
    receivers = torch.repeat_interleave(torch.arange(N), deg)
    E = receivers.shape[0]
    shuffle = torch.randperm(E, generator=gen)
    receivers = receivers[shuffle]
    senders = torch.randint(0, N, (E,), generator=gen)
    topo = topology.EdgeTopology(senders, receivers, N, torch.device('cuda'))
    assert topo.r.max_rows == max_deg
    sd = _mlp_sd(384, 128, True, seed=seed)
    w, wts = _weights(sd, True)
    h0 = torch.randn(N, 128, generator=gen).cuda()
    e0 = torch.randn(E, 128, generator=gen).cuda()
    w_y = torch.randn(E, 128, generator=gen).cuda()
    w_a = torch.randn(N, 128, generator=gen).cuda()

    def run():
        for t in wts:
            t.grad = None
        h, e = h0.clone().requires_grad_(True), e0.clone().requires_grad_(True)
        y, agg = ops.edge_block(h, e, topo, w, ('sum',))
        ((y * w_y).sum() + (agg * w_a).sum()).backward()
        return [y.detach(), agg.detach(), h.grad, e.grad] + [t.grad.clone() for t in wts]

    fused = run()
    again = run()
    for a, b in zip(fused, again):
        assert torch.equal(a, b)                                  # order-independent atomics: bit-reproducible
    keep = ops._FUSED_SEG_MAX_ROWS
    ops._FUSED_SEG_MAX_ROWS = -1                                  # the separate segment-reduce launches
    try:
        plain = run()
    finally:
        ops._FUSED_SEG_MAX_ROWS = keep
    assert torch.equal(fused[0], plain[0])                        # e' itself does not depend on where it is summed
    ref = torch.zeros(N, 128, dtype=torch.float64, device='cuda').index_add_(0, topo.rcv.long(), fused[0].double())
    assert H.rel_err(fused[1], ref) <= 2e-6 and H.rel_err(plain[1], ref) <= 2e-6
    for a, b in zip(fused[2:], plain[2:]):
        assert H.rel_err(a, b) <= 5e-6


# ---------------------------------------------------------------------------------------------------------------
# a4 + whole model: goldens generated by the reference, and oracle parity for every block type
# ---------------------------------------------------------------------------------------------------------------
LAT128 = sorted(glob.glob(os.path.join(H.GOLDEN, 'mgn_*_lat128.pt')))


@pytest.mark.parametrize('path', LAT128, ids=[os.path.basename(p)[4:-3] for p in LAT128])
def test_model_vs_reference_golden(path):
    fx = torch.load(path)
    sd = O.init_state_dict_like(fx['shapes'], fx['seed'])
    graph = H.graph_from_fixture(fx)
    order = list(fx['set_order']) + list(fx['set_order_hyper'])
    model = H.hip_model(fx['arch'], fx['agg'], fx['steps'], fx['edge_sets'], sd, set_order=order)
    out, loss, grads, in_grads = H.hip_run(model, graph, fx['target'], fx['mask'])
    assert H.rel_err(out, fx['out']) <= 2e-5
    assert H.rel_err(loss, fx['loss']) <= 2e-5
    # The golden gradients are the reference's own fp32 numbers; through 15 layers those carry up to ~3e-3 relative
    # rounding noise on the earliest weights (measured against fp64).  So the HIP gradients are required to be at
    # least as close to exact (fp64 oracle) arithmetic as the reference's are, with 5e-5 as the floor.
    _, _, grads64, ing64 = H.oracle_run(sd, graph, fx['arch'], fx['agg'], fx['target'], fx['mask'], set_order=order)
    dg, d64 = H.digest(grads, fx['seed']), H.digest(grads64, fx['seed'])
    for k, ref in fx['grad_digest'].items():
        floor = 5e-5 * float(ref['l2']) * (grads[k].numel() ** 0.5) + 1e-9
        ref_noise = float((ref['proj'] - d64[k]['proj']).abs().max())
        ours = float((dg[k]['proj'] - d64[k]['proj']).abs().max())
        assert ours <= max(floor, 1.5 * ref_noise), (k, ours, ref_noise)
        assert float((dg[k]['proj'] - ref['proj']).abs().max()) <= max(floor, 2.5 * ref_noise), k
        assert abs(float(dg[k]['l2'] - ref['l2'])) <= 1e-4 * float(ref['l2']) + 2.5 * abs(float(ref['l2'] - d64[k]['l2'])) + 1e-9, k
    for x, gref, g64 in zip(in_grads['node'], fx['in_grads']['node'], ing64['node']):
        assert H.rel_err(x, g64) <= max(5e-5, 1.5 * H.rel_err(gref, g64))


CASES = [
    ('none', 'sum', 2, ['mesh_edges'], dict(nx=9, ny=7)),
    ('none', 'pna', 2, ['mesh_edges', 'balance'], dict(nx=9, ny=7, balance=13)),
    ('none', 'max', 1, ['mesh_edges'], dict(nx=9, ny=7)),
    ('none', 'min', 1, ['mesh_edges'], dict(nx=9, ny=7)),
    ('none', 'mean', 1, ['mesh_edges'], dict(nx=9, ny=7)),
    ('multi', 'sum', 1, ['mesh_edges'], dict(nx=9, ny=7)),
    ('repeated', 'sum', 2, ['mesh_edges'], dict(nx=9, ny=7)),
    ('hyper', 'pna', 2, ['mesh_edges', 'intra_cluster_to_mesh', 'intra_cluster_to_cluster', 'inter_cluster'],
     dict(nx=12, ny=8, clusters=5)),
    ('hyper', 'sum', 1, ['mesh_edges', 'intra_cluster_to_mesh', 'intra_cluster_to_cluster', 'inter_cluster', 'world_edges'],
     dict(nx=12, ny=8, clusters=5, world=19)),
    ('hetero', 'pna', 2, ['mesh_edges', 'intra_cluster_to_mesh', 'intra_cluster_to_cluster', 'inter_cluster', 'world_edges'],
     dict(nx=12, ny=8, clusters=5, world=19)),
    ('multiscale', 'sum', 1, ['mesh_edges', 'intra_cluster_to_mesh', 'intra_cluster_to_cluster', 'inter_cluster'],
     dict(nx=12, ny=8, clusters=5)),
]


_KinkMargin, _TieMargin = H.KinkMargin, H.TieMargin      # (instance conditioning probes: tests/helpers.py)


@pytest.mark.parametrize('arch,agg,steps,sets,gkw', CASES, ids=[f'{c[0]}-{c[1]}-L{c[2]}-S{len(c[3])}' for c in CASES])
@pytest.mark.parametrize('index_device', ['cuda', 'cpu'])
def test_model_vs_oracle(arch, agg, steps, sets, gkw, index_device):
    if index_device == 'cpu' and arch not in ('hyper', 'none'):
        pytest.skip('host-resident indices are exercised on two architectures')
    graph = synth.grid_graph(seed=5, **gkw)
    edge_in = {e.name: e.features.shape[1] for e in graph.edge_sets}
    hyper_in = graph.node_features[1].shape[1] if len(graph.node_features) > 1 else 0
    nsn = None
    if arch == 'hetero':
        nsn = {'node_model_cross': len(sets), 'hyper_node_model_cross': len(sets)}
    shapes = O.param_shapes(arch, agg, steps, sets, graph.node_features[0].shape[1], edge_in, hyper_in, 3, 128, nsn)
    N = graph.node_features[0].shape[0]
    gen = torch.Generator().manual_seed(1)
    target = torch.randn(N, 3, generator=gen)
    mask = torch.ones(N, dtype=torch.bool); mask[:3] = False
    order = ['mesh_edges', 'world_edges', 'inter_cluster', 'inter_cluster_world']
    # max/min route a gradient to ONE arg element: when the two best candidates of some segment differ by less than
    # fp32 rounding, which one wins is implementation dependent (the fp32 and fp64 oracles themselves then disagree
    # by ~1e-2).  Such ill-conditioned instances say nothing about parity: pick the first well-conditioned seed.
    for wseed in range(11, 40):
        sd = O.init_state_dict_like(shapes, seed=wseed)
        with _TieMargin() as tm:
            out_o, loss_o, grads_o, ing_o = H.oracle_run(sd, graph, arch, agg, target, mask, set_order=order)
        if agg not in ('pna', 'max', 'min') or tm.worst > 2e-6:     # every max/min winner leads by more than fp32 rounding
            break
    # (how selective the choice was goes into the parity report: seeds rejected before the one used, and its margin)
    H._REPORT.append({'test': f'test_model_vs_oracle[{arch}-{agg}-L{steps}-S{len(sets)}-{index_device}]', 'what': 'instance selection',
                      'first_seed_tried': 11, 'seed_used': wseed, 'seeds_rejected': wseed - 11,
                      'criterion': 'smallest lead of a max / min winner over its runner-up > 2e-6 of the aggregate\'s scale (fp64 oracle)',
                      'margin_of_seed_used': tm.worst if agg in ('pna', 'max', 'min') else None})
    model = H.hip_model(arch, agg, steps, sets, sd, set_order=order)
    out, loss, grads, ing = H.hip_run(model, graph, target, mask, index_device=index_device)
    assert H.rel_err(out, out_o) <= TOL_OUT
    assert H.rel_err(loss, loss_o) <= TOL_OUT
    worst = max((H.rel_err(grads[k], grads_o[k]), k) for k in grads_o if float(grads_o[k].abs().max()) > 0)
    assert worst[0] <= TOL_GRAD, worst
    if index_device == 'cuda':           # element-wise figures next to the norm-wise ones, beside the reference's own fp32
        out32, _, grads32, _ = H.oracle_run(sd, graph, arch, agg, target, mask, set_order=order, dtype=torch.float32)
        tid = f'test_model_vs_oracle[{arch}-{agg}-L{steps}-S{len(sets)}]'
        H.report(tid, 'output', out, out_o, out32)
        wn, we = H.worst_grad(grads, grads_o)
        rn, re_ = H.worst_grad(grads32, grads_o)
        H._REPORT.append({'test': tid, 'what': 'param grads (worst tensor)', 'norm': wn, 'elem': we, 'ref_fp32_norm': rn, 'ref_fp32_elem': re_})
    for k in grads_o:
        if float(grads_o[k].abs().max()) == 0:
            assert float(grads[k].abs().max()) == 0, k
    for a, b in zip(ing['node'], ing_o['node']):
        assert H.rel_err(a, b) <= TOL_GRAD
    for name, b in ing_o['edge'].items():
        if b is not None and ing['edge'][name] is not None:
            assert H.rel_err(ing['edge'][name], b) <= TOL_GRAD, name


def test_flag_L15_sum_vs_oracle_fp64():
    """The headline configuration (architecture none, 15 MP layers, latent 128, sum) on a 12x12 flag-shaped mesh against
    the fp64 oracle: the 1e-5 relative target through 15 residual + LayerNorm layers."""
    graph = synth.grid_graph(seed=2, nx=12, ny=12)
    shapes = O.param_shapes('none', 'sum', 15, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 128)
    N = 144
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(4))
    mask = torch.ones(N, dtype=torch.bool); mask[:3] = False
    # 3.6 million ReLU inputs: an instance where one of them sits within fp32 rounding of zero has a gradient that jumps with
    # the last bit of any implementation (the fp32 and fp64 oracles disagree there too) -- take the first seed without one
    for wseed in range(3, 40):
        sd = O.init_state_dict_like(shapes, seed=wseed)
        with _KinkMargin() as km:
            out_o, loss_o, grads_o, _ = H.oracle_run(sd, graph, 'none', 'sum', target, mask)
        if km.worst > 3e-7:
            break
    H._REPORT.append({'test': 'test_flag_L15_sum_vs_oracle_fp64', 'what': 'instance selection', 'first_seed_tried': 3, 'seed_used': wseed,
                      'seeds_rejected': wseed - 3, 'criterion': 'smallest |ReLU input| of the fp64 oracle run > 3e-7',
                      'margin_of_seed_used': km.worst})
    model = H.hip_model('none', 'sum', 15, ['mesh_edges'], sd)
    out, loss, grads, _, gates, winners = H.hip_run_logged(model, graph, target, mask)
    assert H.rel_err(out, out_o) <= TOL_OUT
    assert H.rel_err(loss, loss_o) <= TOL_OUT
    worst = max((H.rel_err(grads[k], grads_o[k]), k) for k in grads_o)
    # The arithmetic: against the fp64 oracle run with the HIP forward's ReLU gates (tests/helpers.py: GateTransfer) every gradient
    # tensor agrees to 1e-5.  Without the transfer the bound is 5e-5 -- unless a gate of this instance sits within fp32 rounding of
    # its kink after all (the seed search above looks at the fp64 run's margin, 3e-7; a forward that is 3e-7 off at layer 10 can still
    # draw one): then at most three gates may differ, each at |pre-activation| <= 1e-5 (the bound of the headline test), and the transferred figure is the statement.
    _, _, grads_g, gt, _ = H.oracle_run_with_hip_decisions(sd, graph, 'none', 'sum', target, mask, gates, winners)
    gn, _ = H.worst_grad(grads, grads_g)
    H._REPORT.append({'test': 'test_flag_L15_sum_vs_oracle_fp64', 'what': 'param grads (worst tensor)', 'norm': worst[0], 'tensor': worst[1],
                      'norm_with_hip_gates': gn, 'gates_differing_from_fp64': gt.flipped, 'max_abs_preactivation_at_flip': gt.max_abs_at_flip})
    assert gn <= 1e-5, gn
    assert worst[0] <= 5e-5 or (0 < gt.flipped <= 3 and gt.max_abs_at_flip <= 1e-5), (worst, gt.flipped, gt.max_abs_at_flip)


@pytest.mark.parametrize('agg', ['sum', 'pna'])
def test_headline_graph_40x40_L15_vs_oracle_fp64(agg):
    """The headline workload itself against the oracle: ONE 40x40 flag_simple-shape graph (1 600 nodes, 9 282 directed edges),
    architecture none, 15 MP layers, latent 128 -- BASELINE.json configs[1] -- HIP vs the fp64 oracle at 1e-5 on the outputs
    (norm-wise), the element-wise figure reported beside it and bounded by the reference's own fp32 arithmetic (the oracle run
    in fp32); gradients at the L=15 tolerance.  Then the 128-graph batch of the benchmark: rows of graph k equal the
    single-graph result (graphs do not interact; the batch is what bench.py times)."""
    import hgn_amd
    graph = synth.grid_graph(seed=0)
    shapes = O.param_shapes('none', agg, 15, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 128)
    N = 1600
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(4))
    mask = torch.ones(N, dtype=torch.bool); mask[:3] = False
    sd = O.init_state_dict_like(shapes, seed=3)
    out_o, loss_o, grads_o, _ = H.oracle_run(sd, graph, 'none', agg, target, mask)
    out32, _, grads32, _ = H.oracle_run(sd, graph, 'none', agg, target, mask, dtype=torch.float32)
    model = H.hip_model('none', agg, 15, ['mesh_edges'], sd)
    out, loss, grads, _, gates, winners = H.hip_run_logged(model, graph, target, mask)
    tid = f'test_headline_graph_40x40_L15_vs_oracle_fp64[{agg}]'
    r = H.report(tid, 'output', out, out_o, out32)
    assert r['norm'] <= TOL_OUT, r                                  # 1e-5 relative, on the tensor's scale
    assert r['elem'] <= max(1e-5, 2.0 * r['ref_fp32_elem']), r     # element-wise: no worse than 2x the reference's own fp32
    assert H.rel_err(loss, loss_o) <= TOL_OUT
    wn, we = H.worst_grad(grads, grads_o)
    rn, re_ = H.worst_grad(grads32, grads_o)
    H._REPORT.append({'test': tid, 'what': 'param grads (worst tensor)', 'norm': wn, 'elem': we, 'ref_fp32_norm': rn, 'ref_fp32_elem': re_})
    # Gradients at this size: 42 M ReLU inputs (and, pna, 6 M max/min winners) -- on every seed a few sit within fp32 rounding
    # of the kink / a tie, and ONE gate that falls the other way than in fp64 moves every gradient upstream of it by 1e-4..1e-3
    # of its scale.  That is a property of fp32 arithmetic, the reference's included: measured with tests/diag_grad_err_headline.py
    # the reference's own fp32 (the oracle in fp32) is 3.3e-4 / 6.3e-4 (sum, seeds 3 / 4) and 2.7e-4 / 3.9e-4 (pna) from fp64, the
    # HIP path 1.1e-3 / 3.3e-4 and 1.4e-3 / 3.4e-4 -- who draws the flip nearer the output differs per seed.  Flip-free instances
    # exist only at smaller sizes and are held to 5e-5 there (test_flag_L15_sum_vs_oracle_fp64, test_model_vs_oracle); here the
    # bound is the kink-noise level, and both figures go into the parity report.
    assert wn <= 3e-3, (wn, rn)
    # ... and the arithmetic itself: the fp64 oracle with the HIP forward's DISCRETE decisions transferred (tests/helpers.py) -- its
    # ReLU gates and, for pna (the reference's default aggregator, configs/flag.yaml:32), the winners of every max / min.  They
    # differ from fp64's own only where a pre-activation is at fp32 rounding level of 0 / two candidates at rounding level of each
    # other; with them fixed, the gradients of the headline workload meet the north star's 1e-5.
    out_g, _, grads_g, gt, wt = H.oracle_run_with_hip_decisions(sd, graph, 'none', agg, target, mask, gates, winners)
    gn, ge = H.worst_grad(grads, grads_g)
    H._REPORT.append({'test': tid, 'what': 'param grads vs fp64 oracle with the HIP gates / winners (worst tensor)', 'norm': gn, 'elem': ge,
                      'gates': gt.total, 'gates_differing_from_fp64': gt.flipped, 'max_abs_preactivation_at_flip': gt.max_abs_at_flip,
                      'winners': wt.total, 'winners_differing_from_fp64': wt.flipped, 'max_relative_gap_at_flip': wt.max_gap_at_flip})
    print('decision transfer', gt.total, gt.flipped, gt.max_abs_at_flip, wt.total, wt.flipped, wt.max_gap_at_flip, gn, ge)
    assert gt.total == 2 * 128 * (15 * (1600 + 9282) + 2 * 1600 + 9282)
    assert gt.flipped <= 200 and gt.max_abs_at_flip <= 1e-5, (gt.flipped, gt.max_abs_at_flip)
    if agg == 'pna':
        assert wt.total == 15 * 2 * 1600 * 128                  # every node of the grid receives edges: one max and one min winner per feature
        assert wt.flipped <= 200 and wt.max_gap_at_flip <= 1e-5, (wt.flipped, wt.max_gap_at_flip)
    assert H.rel_err(out, out_g) <= TOL_OUT
    assert gn <= 1e-5, gn                               # sum: measured 6.1e-7 (7 of 45 M gates differ from fp64's, each at |z| < 1e-6)
    # ---- the benchmark batch: 128 graphs, graph k's rows == the single-graph result ---------------------------------
    graphs = [graph] + [synth.grid_graph(seed=s) for s in (1, 2, 3)]
    members = [graphs[(i * 7) % 4] if i != 77 else graph for i in range(128)]
    big = synth.batch(members)
    with torch.no_grad():
        ob = model(hgn_amd.MultiGraph([x.cuda() for x in big.node_features],
                                      [hgn_amd.EdgeSet(e.name, e.features.cuda(), e.senders.cuda(), e.receivers.cuda())
                                       for e in big.edge_sets]))
    for k in (0, 77, 124):
        if members[k] is graph:
            assert H.rel_err(ob[k * N:(k + 1) * N], out) <= 2e-6, k           # same arithmetic per row, other tiling
            assert H.rel_err(ob[k * N:(k + 1) * N], out_o) <= TOL_OUT, k


def test_edge_order_invariance_and_batch_independence_full_size():
    """Size-independent properties at the benchmark size (8 flag_simple-shape graphs, 74k edges, L=3):
    (i) shuffling the edge list does not change the output beyond rounding; (ii) graphs in a batch do not interact:
    graph 0's rows equal the single-graph result."""
    import hgn_amd
    graphs = [synth.grid_graph(seed=s) for s in range(8)]
    big = synth.batch(graphs)
    shapes = O.param_shapes('none', 'sum', 3, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=1)
    model = H.hip_model('none', 'sum', 3, ['mesh_edges'], sd)

    def run(g):
        with torch.no_grad():
            return model(hgn_amd.MultiGraph([x.cuda() for x in g.node_features],
                                            [hgn_amd.EdgeSet(e.name, e.features.cuda(), e.senders.cuda(), e.receivers.cuda())
                                             for e in g.edge_sets]))
    out = run(big)
    e = big.edge_sets[0]
    p = torch.randperm(e.senders.shape[0], generator=torch.Generator().manual_seed(0))
    shuffled = synth.MultiGraph(big.node_features, [synth.EdgeSet(e.name, e.features[p], e.senders[p], e.receivers[p])])
    assert H.rel_err(run(shuffled), out) <= TOL_OUT
    n0 = graphs[0].node_features[0].shape[0]
    assert H.rel_err(out[:n0], run(graphs[0])) <= TOL_OUT


def test_state_dict_keys_and_lazy_api():
    """API fidelity of the drop-in boundary: lazy parameters before the first forward, reference key names after."""
    import hgn_amd
    torch.manual_seed(0)
    m = hgn_amd.MeshGraphNet(3, 128, 2, 'pna', 2, 'hyper',
                             ['mesh_edges', 'intra_cluster_to_mesh', 'intra_cluster_to_cluster', 'inter_cluster']).to('cuda')
    params_before = list(m.parameters())
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)              # reference: MeshSimulator.py:110, before any forward
    g = synth.grid_graph(seed=0, nx=8, ny=8, clusters=4)
    gg = hgn_amd.MultiGraph([x.cuda() for x in g.node_features],
                            [hgn_amd.EdgeSet(e.name, e.features.cuda(), e.senders, e.receivers) for e in g.edge_sets])
    out = m(gg)
    assert out.shape == (64, 3) and out.is_cuda
    out.square().mean().backward()
    opt.step()
    assert all(a is b for a, b in zip(params_before, m.parameters()))
    keys = set(m.state_dict().keys())
    expect = set(O.param_shapes('hyper', 'pna', 2, ['mesh_edges', 'intra_cluster_to_mesh', 'intra_cluster_to_cluster',
                                                    'inter_cluster'], 5, {n: 7 for n in
                                                                          ['mesh_edges', 'intra_cluster_to_mesh',
                                                                           'intra_cluster_to_cluster', 'inter_cluster']},
                                8, 3, 128).keys())
    assert keys == expect
    import pickle
    m2 = pickle.loads(pickle.dumps(m))
    with torch.no_grad():
        assert torch.equal(m2(gg), m(gg))


def test_one_launch_pack_table_equals_per_mlp_packs(monkeypatch):
    """A trainer's step refreshes every packed weight image with ONE launch over a descriptor table in device memory
    (include/hgn_mp.h: hgn_pack_bf16x3_table, ops.PackPlan: recorded during the first step, used from the second on) instead of one
    launch per MLP and form.  Same images: three training steps give the same parameters bit for bit with and without the plan, the
    plan covers every (weights, form) pair the step uses, and a moved precision rebuilds the table."""
    from hgn_amd import ops, parallel
    graph = synth.grid_graph(seed=3, nx=9, ny=8, clusters=4)
    sets = [e.name for e in graph.edge_sets]
    shapes = O.param_shapes('hyper', 'pna', 2, sets, 5, {n: 7 for n in sets}, 8, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=4)
    N = 72
    import hgn_amd
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(1)).cuda()
    mask = torch.ones(N, dtype=torch.bool).cuda()
    g = hgn_amd.MultiGraph([x.cuda() for x in graph.node_features],
                           [hgn_amd.EdgeSet(e.name, e.features.cuda(), e.senders.cuda(), e.receivers.cuda()) for e in graph.edge_sets])

    def run(with_plan):
        if not with_plan:
            monkeypatch.setattr(ops, 'begin_step_packs', lambda c: None)
        m = H.hip_model('hyper', 'pna', 2, sets, sd)
        tr = parallel.DataParallelTrainer(m, lr=1e-3)
        losses = [tr.step(g, target, mask).item() for _ in range(3)]
        monkeypatch.undo()
        return tr, losses

    tr1, l1 = run(True)
    tr0, l0 = run(False)
    assert tr0.ctx.pack_plan is None
    plan = tr1.ctx.pack_plan
    assert plan is not None and plan.table is not None and len(plan.items) >= 2 * 4 * 2      # 2 blocks x (>= 4 MLPs) x 2 forms
    assert l1 == l0 and torch.equal(tr1.fp.flat, tr0.fp.flat)
    # every image the step asks for is the plan's, fresh for the current epoch
    rec = []
    tr1.ctx.pack_recorder = rec
    ops.begin_step_packs(tr1.ctx)
    with ops.using(tr1.ctx):
        tr1.model(g).sum().backward()
    tr1.ctx.pack_recorder = None
    have = {(w.w1.data_ptr(), tr) for w, tr in plan.items}
    assert rec and all((w.w1.data_ptr(), bool(tr)) in have for w, tr in rec)
    table, ptr, gen = plan.table, plan.table.data_ptr(), plan.generation
    before = bytes(plan.table.cpu().numpy().tobytes())
    tr1.ctx.set_matmul_precision('bf16')                       # other images: same table layout, but the descriptors' form flags change ->
    ops.begin_step_packs(tr1.ctx)                              # rewritten IN PLACE (a captured step has the table's address baked in)
    assert plan.table is table and plan.table.data_ptr() == ptr and plan.generation == gen + 1
    assert bytes(plan.table.cpu().numpy().tobytes()) != before


def test_trainer_flat_gradients_and_adam_match_torch():
    """Flat-buffer training path (kernels accumulate straight into the flat gradient, fused HIP Adam) against plain
    autograd + torch.optim.Adam on the same model / batch: gradients after one step and weights after three."""
    import copy
    import hgn_amd
    from hgn_amd import parallel
    graph = synth.grid_graph(seed=7, nx=10, ny=9, clusters=4)
    sets = [e.name for e in graph.edge_sets]
    shapes = O.param_shapes('multiscale', 'pna', 2, sets, 5, {n: 7 for n in sets}, 8, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=5)
    N = 90
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(2)).cuda()
    mask = torch.ones(N, dtype=torch.bool); mask[:3] = False
    mask = mask.cuda()
    g = hgn_amd.MultiGraph([x.cuda() for x in graph.node_features],
                           [hgn_amd.EdgeSet(e.name, e.features.cuda(), e.senders.cuda(), e.receivers.cuda()) for e in graph.edge_sets])
    ref = H.hip_model('multiscale', 'pna', 2, sets, sd)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    flat = H.hip_model('multiscale', 'pna', 2, sets, sd)
    tr = parallel.DataParallelTrainer(flat, lr=1e-3, wgrad_stream=True)      # weight gradients on the second stream
    for step in range(3):
        opt.zero_grad()
        out = ref(g)
        loss = torch.nn.functional.mse_loss(target[mask], out[mask])
        loss.backward()
        if step == 0:
            g_ref = {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for k, p in ref.named_parameters()}
        opt.step()
        l2 = tr.step(g, target, mask)
        if step == 0:
            for k, p in flat.named_parameters():
                # (the two paths sum the rows of a weight / bias gradient in different orders: fp32 rounding of a sum over rows)
                assert H.rel_err(p.grad, g_ref[k]) <= 2e-6 or float(g_ref[k].abs().max()) == 0, k
        assert abs(float(l2) - float(loss.detach())) <= 1e-5 * abs(float(loss.detach()))
    # Adam divides by sqrt(v): where a gradient is ~0 the update direction is rounding-sensitive, so weights are compared
    # on the scale of the updates they received (3 steps x lr): 0.5 % of that.  The two optimisers round differently, so
    # from the second step on the models differ in the last bits and a pna max/min winner may change in one of them; that
    # moves the gradients of a handful of rows, hence "all but 2 % of the entries (at least four: the inter-cluster models see 8 edge rows)" rather than "all", and a hard cap for the rest.
    for (k, p), (_, q) in zip(ref.named_parameters(), flat.named_parameters()):
        d = (q - p).abs()
        assert int((d > 0.005 * 3 * 1e-3).sum()) <= max(4, d.numel() // 50), k
        assert float(d.max()) <= 2 * 3 * 1e-3, k


@pytest.mark.parametrize('arch,agg', [('none', 'sum'), ('hetero', 'pna'), ('repeated', 'sum')])
def test_deferred_gradient_sums_equal_the_per_call_reductions_bit_for_bit(arch, agg):
    """Flat-buffer training: the LayerNorm-affine gradients of every MLP are summed by ONE launch at the end of the backward pass
    (ops._ln_defer, hgn_ln_reduce_batch: each backward call leaves its per-workgroup slabs in a workspace of its own) instead of a
    reduction launch per call, and so are the chunk slabs of the weight-gradient launches (ops._wred_deferrable,
    hgn_mlp_wgrad_partial / hgn_edge_bwd_fused_partial -> hgn_slab_reduce_batch; sums that share a target -- `repeated` applies one
    MLP twice -- go into separate launches).  Same slabs, same fixed-order sums: the whole flat gradient must be bit-identical
    with and without the deferral, eagerly and under HIP-graph capture (loss and weights after two steps)."""
    import hgn_amd
    from hgn_amd import parallel, ops, graphs as hg
    graph = synth.grid_graph(seed=11, nx=12, ny=9, clusters=3 if arch == 'hetero' else 0)
    sets = [e.name for e in graph.edge_sets]
    shapes = O.param_shapes(arch, agg, 3, sets, 5, {e.name: e.features.shape[1] for e in graph.edge_sets},
                            graph.node_features[1].shape[1] if len(graph.node_features) > 1 else 0, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=8)
    N = graph.node_features[0].shape[0]
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(3)).cuda()
    mask = torch.ones(N, dtype=torch.bool).cuda()
    g = hgn_amd.MultiGraph([x.cuda() for x in graph.node_features],
                           [hgn_amd.EdgeSet(e.name, e.features.cuda(), e.senders.cuda(), e.receivers.cuda()) for e in graph.edge_sets])

    def run(defer, captured):
        old = (ops._DEFER_LN, ops._DEFER_WRED)
        ops._DEFER_LN = ops._DEFER_WRED = defer
        try:
            tr = parallel.DataParallelTrainer(H.hip_model(arch, agg, 3, sets, sd), lr=1e-3, device_step=captured)
            step = hg.GraphedTrainStep(tr, g, target, mask, warmup=1) if captured else (lambda: tr.step(g, target, mask))
            losses = [float(step()) for _ in range(2)]
            torch.cuda.synchronize()
            return losses, tr.fp.grad.clone(), tr.fp.flat.clone()
        finally:
            ops._DEFER_LN, ops._DEFER_WRED = old
    base = run(False, False)
    for defer, captured in ((True, False), (True, True)):
        got = run(defer, captured)
        if not captured:
            assert got[0] == base[0]
            assert torch.equal(got[1], base[1]), 'flat gradient'
            assert torch.equal(got[2], base[2]), 'weights after two steps'
        else:                                   # (the captured step warms up once more: compare with its own eager twin instead)
            twin = run(False, True)
            assert got[0] == twin[0] and torch.equal(got[1], twin[1]) and torch.equal(got[2], twin[2])


def test_hip_graph_forward_and_train_step_replay():
    """f4: the forward (rollout) and the whole training step captured into a HIP graph replay bit-identically to the eager
    path on new inputs (fixed topology)."""
    import hgn_amd
    from hgn_amd import graphs, parallel
    g0 = synth.grid_graph(seed=1, nx=12, ny=10)
    g1 = synth.grid_graph(seed=2, nx=12, ny=10)
    shapes = O.param_shapes('none', 'sum', 3, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=1)

    def dev(g):
        return hgn_amd.MultiGraph([x.cuda() for x in g.node_features],
                                  [hgn_amd.EdgeSet(e.name, e.features.cuda(), e.senders.cuda(), e.receivers.cuda()) for e in g.edge_sets])
    model = H.hip_model('none', 'sum', 3, ['mesh_edges'], sd)
    G0, G1 = dev(g0), dev(g1)
    gf = graphs.GraphedForward(model, G0)
    with torch.no_grad():
        ref0, ref1 = model(G0).clone(), model(G1).clone()
    out0 = gf(G0.node_features, {'mesh_edges': G0.edge_sets[0].features}).clone()
    out1 = gf(G1.node_features, {'mesh_edges': G1.edge_sets[0].features}).clone()
    assert torch.equal(out0, ref0) and torch.equal(out1, ref1)
    # The packed weight images are not part of the captured forward: weights updated between two rollouts (in-place torch update ->
    # version counter; our Adam kernels -> pack epoch) must reach the next replay.
    from hgn_amd import ops as _ops
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.03)
        ref2 = model(G1).clone()
    out2 = gf(G1.node_features, {'mesh_edges': G1.edge_sets[0].features}).clone()
    assert torch.equal(out2, ref2) and not torch.equal(out2, ref1)
    with torch.no_grad():
        for p in model.parameters():
            p.data.copy_(p.data / 1.03)             # behind the version counters, as a fused optimiser kernel would ...
        _ops.invalidate_packs()                      # ... which then announces it
        ref3 = model(G1).clone()
    out3 = gf(G1.node_features, {'mesh_edges': G1.edge_sets[0].features}).clone()
    assert torch.equal(out3, ref3)
    # training step
    N = 120
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(0)).cuda()
    mask = torch.ones(N, dtype=torch.bool).cuda()
    # (the side stream of `captured` selects the two-launch edge backward; `eager` gets the same arithmetic: the fused kernel
    # sums the weight gradients' rows in another order, 2e-6 per step)
    from hgn_amd import ops
    ops.set_fused_edge_backward(False)
    try:
        eager = parallel.DataParallelTrainer(H.hip_model('none', 'sum', 3, ['mesh_edges'], sd), lr=1e-3)
        captured = parallel.DataParallelTrainer(H.hip_model('none', 'sum', 3, ['mesh_edges'], sd), lr=1e-3, device_step=True,
                                                wgrad_stream=True)
        for _ in range(3):                                   # GraphedTrainStep warms up with 3 eager steps + 1 captured
            eager.step(G0, target, mask)
        gs = graphs.GraphedTrainStep(captured, G0, target, mask, warmup=3)
        eager.step(G0, target, mask)                          # the step executed during capture? no: capture does not execute
        l_e = [float(eager.step(G1, target, mask)) for _ in range(2)]
        gs()                                                  # replay #1 on G0 (matches eager's 4th step)
        l_g = [float(gs(G1.node_features, {'mesh_edges': G1.edge_sets[0].features})) for _ in range(2)]
    finally:
        ops.set_fused_edge_backward(None)
    assert int(captured.t_dev) == 6
    for a, b in zip(l_e, l_g):
        assert abs(a - b) <= 1e-6 * abs(a)
    assert H.rel_err(captured.fp.flat, eager.fp.flat) <= 1e-6
    # an EAGER forward after graphed training must see the replayed Adam update in its packed weight images too
    with torch.no_grad():
        o_g, o_e = captured.model(G1), eager.model(G1)
    assert H.rel_err(o_g, o_e) <= 2e-6


def test_graphed_shard_step_matches_trainer_step():
    """Data-parallel step with forward+backward replayed from a HIP graph and the (here absent) collectives + Adam eager:
    same parameters and losses as DataParallelTrainer.step, which scales the loss before the backward instead of after."""
    import hgn_amd
    from hgn_amd import graphs, parallel
    g0 = synth.grid_graph(seed=3, nx=12, ny=10)
    shapes = O.param_shapes('none', 'sum', 3, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=2)
    G0 = hgn_amd.MultiGraph([x.cuda() for x in g0.node_features],
                            [hgn_amd.EdgeSet(e.name, e.features.cuda(), e.senders.cuda(), e.receivers.cuda()) for e in g0.edge_sets])
    N = 120
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(0)).cuda()
    mask = torch.ones(N, dtype=torch.bool).cuda()
    mask[:7] = False
    eager = parallel.DataParallelTrainer(H.hip_model('none', 'sum', 3, ['mesh_edges'], sd), lr=1e-3)
    shard = parallel.DataParallelTrainer(H.hip_model('none', 'sum', 3, ['mesh_edges'], sd), lr=1e-3, device_step=True)
    gs = graphs.GraphedShardStep(shard, G0, target, mask, warmup=1)       # warm-up touches gradients only, not parameters
    l_e, l_g = [float(eager.step(G0, target, mask))], [float(gs())]
    assert abs(l_e[0] - l_g[0]) <= 2e-6 * abs(l_e[0])                  # same weights: only the place of the scaling differs
    assert H.rel_err(shard.fp.grad, eager.fp.grad) <= 1e-5               # gradient of the global mean, scaled after vs before
    l_e += [float(eager.step(G0, target, mask)) for _ in range(3)]
    l_g += [float(gs()) for _ in range(3)]
    for a, b in zip(l_e, l_g):        # later steps: Adam turns rounding-level gradient entries into +-lr moves, losses stay close
        assert abs(a - b) <= 5e-5 * abs(a), (a, b)
    assert int(shard.t_dev) == 4


def test_captured_step_survives_another_models_storage_move_and_an_eager_step_in_between():
    """A captured training step bakes in the device address of its PackPlan's descriptor table and of the packed images.  Moving
    ANOTHER model (`.to()` bumps the process-wide storage epoch), then running an eager step of the captured trainer (which finds
    the plan's signature moved and re-prepares it), must not free or replace what the graph replays against: the table is
    rewritten in place and only when its bytes changed (here: not at all), the graph is not captured again, and the replays
    continue the eager trajectory.  Re-homing the trainer's OWN parameters does change the descriptors: the next call captures again."""
    import hgn_amd
    from hgn_amd import graphs, parallel
    g0 = synth.grid_graph(seed=4, nx=12, ny=10)
    shapes = O.param_shapes('none', 'sum', 2, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=3)
    G0 = hgn_amd.MultiGraph([x.cuda() for x in g0.node_features],
                            [hgn_amd.EdgeSet(e.name, e.features.cuda(), e.senders.cuda(), e.receivers.cuda()) for e in g0.edge_sets])
    N = 120
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(0)).cuda()
    mask = torch.ones(N, dtype=torch.bool).cuda()
    mask[:5] = False
    ref = parallel.DataParallelTrainer(H.hip_model('none', 'sum', 2, ['mesh_edges'], sd), lr=1e-3, device_step=True)
    tr = parallel.DataParallelTrainer(H.hip_model('none', 'sum', 2, ['mesh_edges'], sd), lr=1e-3, device_step=True)
    gs = graphs.GraphedShardStep(tr, G0, target, mask, warmup=2)
    l_ref = [float(ref.step(G0, target, mask)) for _ in range(5)]
    l = [float(gs())]
    plan = tr.ctx.pack_plan
    assert plan is not None and plan.table is not None
    table_ptr, gen = plan.table.data_ptr(), plan.generation
    other = H.hip_model('none', 'sum', 1, ['mesh_edges'], O.init_state_dict_like(
        O.param_shapes('none', 'sum', 1, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 128), seed=5))
    other.to('cuda')                                      # MeshGraphNet._apply: bumps the storage epoch of the whole process
    other.float()
    l.append(float(tr.step(G0, target, mask)))            # eager step of the captured trainer: PackPlan.prepare sees a moved signature
    assert plan.table.data_ptr() == table_ptr and plan.generation == gen and not plan.retired
    torch.cuda.empty_cache()                              # anything dropped would be unmapped now
    l += [float(gs()) for _ in range(3)]
    assert gs.captures == 1
    for a, b in zip(l_ref, l):
        assert abs(a - b) <= 5e-5 * abs(a), (l_ref, l)
    # the trainer's own parameters re-homed: descriptors change in place, the step is captured again and stays correct
    ref2 = [float(ref.step(G0, target, mask)) for _ in range(2)]
    tr.model.to('cuda')
    tr.fp = parallel.FlatParams(tr.model)                 # new flat buffers: every weight address moves
    assert tr.fp.flat.data_ptr() != 0
    # (the optimiser state lives in the old trainer; only the forward / backward addresses matter here)
    before = gs.captures
    gs()
    assert gs.captures == before + 1 and plan.table.data_ptr() == table_ptr


@pytest.mark.parametrize('mode', ['fp32', 'fp32-bf16x3'])
def test_split_products_cover_the_fp32_range(mode):
    """Product mode 3 (the default: two fp16 terms per operand, three MFMAs per product) lives on powers of two that put every operand
    into fp16's five exponent bits: per row for activations and gradients, per packed block for weights, per 32-row block for the
    operands of a weight gradient (csrc/mlp6_device.h: Prod<3>).  The scales must make the arithmetic independent of magnitude: an edge
    block + node update, forward and backward, against fp64 autograd with rows whose magnitudes lie e^(+-12) apart, inputs at 1e-12
    and 1e+12, weights x 1e-4 and x 300, loss scales 1e-18 .. 1e+12, an all-zero row, and a whole zero tile of 64 rows -- every
    output and gradient tensor within 5e-6 of fp64 norm-wise (2e-6 away from the extremes), nothing non-finite.  The range: one
    scale is 2^e with |e| <= 120 (a normal fp32 factor), so rows and blocks whose largest element is >= 2^-105 (2.5e-32) keep
    full precision; two scales add up to |e| <= 240 and come off the accumulators in two factors.  The regimes at 1e-12 put
    gradient rows at 6e-18 (sum of exponents 127: the two-factor path) and at 1e-28 .. 1e-30 (one exponent at 115); tensors
    whose TRUE rms is below 1e-34 (the first layers' weight gradients there, 1e-40: fp32 denormals) are only required to be
    finite.  Below 2^-105 a row loses one bit per octave (measured at 1e-33: 4e-5).  The six-product bf16 mode, which needs no
    scales, runs as the second parameter on the same inputs -- except the regime with gradients around 1e-30, where its third
    bf16 term falls into fp32's denormal range (5e-4 off, measured at 1e-32): there the scaled mode is the more accurate one.
    The regime with edge rows at 1e-30 beside node rows at 1 is the overflow case of the scaled accumulators: the edge operand's
    scale (2^115 and the block's on top) would push the bias + node projections the accumulators start from beyond fp32, were it
    not capped by what they hold (csrc/split_bf16.h: SCALE_EASY / acc_room)."""
    from hgn_amd import ops, topology, modules
    import hgn_amd
    g = synth.grid_graph(seed=3, nx=20, ny=20)
    es = g.edge_sets[0]
    N, E = g.node_features[0].shape[0], es.senders.shape[0]
    topo = topology.EdgeTopology(es.senders.cuda(), es.receivers.cuda(), N, torch.device('cuda'))
    worst = 0.0
    for h_scale, e_scale, w_scale, spread, gscale in ((1.0, 1.0, 1.0, 4.0, 1.0), (1e-12, 1e-12, 1.0, 0.0, 1e-8), (1e-12, 1e-12, 1.0, 0.0, 1e-18), (1.0, 1e-30, 1.0, 0.0, 1.0), (1e12, 1e10, 1.0, 0.0, 1e12),
                                                     (1.0, 1.0, 300.0, 2.0, 1e-6), (1.0, 1.0, 1e-4, 0.0, 1e6)):
        if mode == 'fp32-bf16x3' and gscale < 1e-10:
            continue
        torch.manual_seed(0)
        m = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges']).cuda()
        with torch.no_grad():
            m(hgn_amd.MultiGraph([g.node_features[0].cuda()], [hgn_amd.EdgeSet(es.name, es.features.cuda(), es.senders.cuda(), es.receivers.cuda())]))
            for p_ in m.parameters():
                if p_.dim() == 2:
                    p_.mul_(w_scale)
        blk = m.processor.graphnet_blocks[0]
        we = modules.weights_of(blk.edge_models['mesh_edges'], 384)
        wn = modules.weights_of(blk.node_model_cross, 256)
        gen = torch.Generator().manual_seed(1)
        h0 = (torch.randn(N, 128, generator=gen) * h_scale).cuda()
        rs = torch.exp(torch.randn(E, 1, generator=gen) * spread)
        e0 = (torch.randn(E, 128, generator=gen) * e_scale * rs).cuda()
        e0[5] = 0                                              # an all-zero row
        e0[128:192] = 0                                        # a whole 64-row tile of zeros
        rowsc = torch.exp(torch.randn(E, 1, generator=gen) * spread).cuda()      # gradient rows of very different magnitude
        # The loss is LINEAR in the outputs (fixed random cotangents): with |out|^2 and inputs at 1e-12 the outputs are pure LayerNorm
        # rows, whose squared norm does not depend on the MLP at all -- the true gradient into the MLP is then the residue of a
        # cancellation (measured: 1e-3 relative in ANY fp32 arithmetic), which says nothing about the products.
        r_hn = torch.randn(N, 128, generator=gen).cuda()
        r_y = torch.randn(E, 128, generator=gen).cuda()

        def run():
            h = h0.clone().requires_grad_(True); e = e0.clone().requires_grad_(True)
            for p_ in blk.parameters():
                p_.grad = None
            y, agg = ops.edge_block(h, e, topo, we, ('sum',))
            hn = ops.fused_mlp([h, agg], wn, None, 0)
            (((hn * r_hn).sum() + (y * rowsc * r_y).sum()) * gscale).backward()
            return [y.detach(), hn.detach(), h.grad, e.grad] + [p_.grad.clone() for p_ in blk.parameters()]
        snd, rcv = topo.snd.long(), topo.rcv.long()
        h = h0.double().requires_grad_(True); e = e0.double().requires_grad_(True)
        ps = [p_.detach().double().requires_grad_(True) for p_ in blk.parameters()]
        P = dict(zip([n for n, _ in blk.named_parameters()], ps))

        def mlp(x, pre):
            z = torch.relu(x @ P[pre + '.0.layers.linear_0.weight'].T + P[pre + '.0.layers.linear_0.bias'])
            z = torch.relu(z @ P[pre + '.0.layers.linear_1.weight'].T + P[pre + '.0.layers.linear_1.bias'])
            z = z @ P[pre + '.0.layers.linear_2.weight'].T + P[pre + '.0.layers.linear_2.bias']
            return torch.nn.functional.layer_norm(z, (128,), P[pre + '.1.weight'], P[pre + '.1.bias'], 1e-5)
        y = e + mlp(torch.cat([h[snd], h[rcv], e], 1), 'edge_models.mesh_edges')
        agg = torch.zeros(N, 128, dtype=torch.float64, device='cuda').index_add(0, rcv, y)
        hn = h + mlp(torch.cat([h, agg], 1), 'node_model_cross')
        (((hn * r_hn.double()).sum() + (y * rowsc.double() * r_y.double()).sum()) * gscale).backward()
        want = [y.detach(), hn.detach(), h.grad, e.grad] + [p_.grad for p_ in ps]
        with ops.using(ops.Context(precision=mode)):
            got = run()
        names = ['y', 'hn', 'dh', 'de'] + ['d ' + n for n, _ in blk.named_parameters()]
        errs = {}
        for n_, a_, b_ in zip(names, got, want):
            assert bool(torch.isfinite(a_).all()), (mode, h_scale, w_scale, gscale, n_)
            if float(b_.square().mean().sqrt()) < 1e-34:      # (fp32 itself holds such a tensor in denormals only: finite is all there is to ask)
                continue
            errs[n_] = H.rel_err(a_, b_)
        worst = max(worst, max(errs.values()))
        bound = 2e-6 if (h_scale == 1.0 and w_scale == 1.0) else 5e-6
        bad = {n_: f'{v:.2e}' for n_, v in errs.items() if v > bound}
        print((h_scale, e_scale, w_scale, spread, gscale), {n_: f'{v:.2e}' for n_, v in errs.items()})      # (shown by pytest when the test fails)
        assert not bad, (mode, h_scale, e_scale, w_scale, spread, gscale, bad)
    H._REPORT.append({'test': f'test_split_products_cover_the_fp32_range[{mode}]', 'what': 'worst tensor over seven magnitude regimes', 'norm': worst})


def test_split_bf16_products_are_fp32_accurate():
    """The default kernels evaluate each fp32 product as six bf16 MFMAs on 3-way bf16 splits (csrc/mlp6.hip): against fp64 they
    must be as accurate as the plain fp32-MFMA kernels (HGN_FP32_MFMA=1) on the same inputs -- forward, data gradients and
    weight gradients of an edge block and of a two-source node MLP."""
    import os
    from hgn_amd import ops, topology, modules
    import hgn_amd
    g = synth.grid_graph(seed=3, nx=20, ny=20)
    es = g.edge_sets[0]
    N, E = g.node_features[0].shape[0], es.senders.shape[0]
    topo = topology.EdgeTopology(es.senders.cuda(), es.receivers.cuda(), N, torch.device('cuda'))
    torch.manual_seed(0)
    m = hgn_amd.MeshGraphNet(3, 128, 2, 'sum', 1, 'none', ['mesh_edges']).cuda()
    with torch.no_grad():
        m(hgn_amd.MultiGraph([g.node_features[0].cuda()], [hgn_amd.EdgeSet(es.name, es.features.cuda(), es.senders.cuda(), es.receivers.cuda())]))
    blk = m.processor.graphnet_blocks[0]
    we = modules.weights_of(blk.edge_models['mesh_edges'], 384)
    wn = modules.weights_of(blk.node_model_cross, 256)
    h0 = torch.randn(N, 128, generator=torch.Generator().manual_seed(1)).cuda()
    e0 = torch.randn(E, 128, generator=torch.Generator().manual_seed(2)).cuda()

    def run():
        h = h0.clone().requires_grad_(True); e = e0.clone().requires_grad_(True)
        for p in list(blk.parameters()):
            p.grad = None
        y, agg = ops.edge_block(h, e, topo, we, ('sum',))
        hn = ops.fused_mlp([h, agg], wn, None, 0)
        (hn.square().sum() + y.square().sum()).backward()
        return [y.detach(), hn.detach(), h.grad, e.grad] + [p.grad.clone() for p in blk.parameters()]

    def ref64():
        snd, rcv = topo.snd.long(), topo.rcv.long()
        h = h0.double().requires_grad_(True); e = e0.double().requires_grad_(True)
        ps = [p.detach().double().requires_grad_(True) for p in blk.parameters()]
        names = [n for n, _ in blk.named_parameters()]
        P = dict(zip(names, ps))

        def mlp(x, pre):
            z = torch.relu(x @ P[pre + '.0.layers.linear_0.weight'].T + P[pre + '.0.layers.linear_0.bias'])
            z = torch.relu(z @ P[pre + '.0.layers.linear_1.weight'].T + P[pre + '.0.layers.linear_1.bias'])
            z = z @ P[pre + '.0.layers.linear_2.weight'].T + P[pre + '.0.layers.linear_2.bias']
            return torch.nn.functional.layer_norm(z, (128,), P[pre + '.1.weight'], P[pre + '.1.bias'], 1e-5)
        y = e + mlp(torch.cat([h[snd], h[rcv], e], 1), 'edge_models.mesh_edges')
        agg = torch.zeros(N, 128, dtype=torch.float64, device='cuda').index_add(0, rcv, y)
        hn = h + mlp(torch.cat([h, agg], 1), 'node_model_cross')
        (hn.square().sum() + y.square().sum()).backward()
        return [y.detach(), hn.detach(), h.grad, e.grad] + [p.grad for p in ps]
    r64 = ref64()
    split = run()
    os.environ['HGN_FP32_MFMA'] = '1'                      # read by the library at every launch
    try:
        plain = run()
    finally:
        del os.environ['HGN_FP32_MFMA']
    for a, b, c in zip(split, plain, r64):
        e6, e32 = H.rel_err(a, c), H.rel_err(b, c)
        assert e6 <= max(1.5 * e32, 2e-6), (e6, e32)
    # the opt-in reduced-precision mode (one bf16 MFMA per product, BASELINE.json configs[4]): really bf16-grade, still sane,
    # and switching back restores the fp32-accurate results bit for bit
    assert hgn_amd.get_matmul_precision() == 'fp32'
    hgn_amd.set_matmul_precision('bf16')
    try:
        low = run()
    finally:
        hgn_amd.set_matmul_precision('fp32')
    # (i) it computes exactly what it claims: the forward equals an fp64 evaluation whose matrix-product operands are rounded
    #     to bf16 (fp32 accumulation, LayerNorm and residual in fp32)
    def rb(x):
        return x.detach().float().bfloat16().double()
    with torch.no_grad():
        P = {n: p.detach().double() for n, p in blk.named_parameters()}

        def mlp_bf16(x, pre):
            z = torch.relu(rb(x) @ rb(P[pre + '.0.layers.linear_0.weight']).T + P[pre + '.0.layers.linear_0.bias'])
            z = torch.relu(rb(z) @ rb(P[pre + '.0.layers.linear_1.weight']).T + P[pre + '.0.layers.linear_1.bias'])
            z = rb(z) @ rb(P[pre + '.0.layers.linear_2.weight']).T + P[pre + '.0.layers.linear_2.bias']
            return torch.nn.functional.layer_norm(z, (128,), P[pre + '.1.weight'], P[pre + '.1.bias'], 1e-5)
        snd, rcv = topo.snd.long(), topo.rcv.long()
        h, e = h0.double(), e0.double()
        y_e = e + mlp_bf16(torch.cat([h[snd], h[rcv], e], 1), 'edge_models.mesh_edges')
        agg_e = torch.zeros(N, 128, dtype=torch.float64, device='cuda').index_add(0, rcv, low[0].double())
        hn_e = h + mlp_bf16(torch.cat([h, agg_e], 1), 'node_model_cross')
    #     mean error: a value that sits on a bf16 rounding boundary may round the other way under fp32 vs fp64 accumulation
    #     (a 4e-3 step for that one operand), so single entries differ by up to ~1e-3 of the output range
    for got, ref in ((low[0], y_e), (low[1], hn_e)):
        mean_err = float((got.double() - ref).abs().mean() / ref.abs().mean())
        assert mean_err <= 2e-5 and H.rel_err(got, ref) <= 3e-3, (mean_err, H.rel_err(got, ref))
    # (ii) it is bf16-grade, not fp32-grade, against the exact result; gradients stay aligned with the exact ones (ReLU gates
    #      near zero flip under the rounding, so their max-norm error is several percent)
    errs = [H.rel_err(a, c) for a, c in zip(low, r64)]
    assert 2e-5 <= errs[0] <= 1e-2 and 2e-5 <= errs[1] <= 1e-2 and max(errs) <= 0.25, errs
    for a, c in zip(low[2:], r64[2:]):
        cos = torch.nn.functional.cosine_similarity(a.double().flatten(), c.double().flatten(), dim=0)
        assert float(cos) >= 0.99, float(cos)
    again = run()
    for a, b in zip(split, again):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        hgn_amd.set_matmul_precision('fp8')


# -------------------------------------------------------------------------------------------------------------
# empty edge sets through the whole model, forward AND backward (plate `world_edges` with no obstacle in range,
# plate.py:84-110: the normal case for most frames).  Both gradient paths: per-parameter tensors and the flat buffer.
# -------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('arch,agg', [('none', 'sum'), ('none', 'pna'), ('hetero', 'pna')])
def test_empty_edge_set_forward_backward_vs_oracle(arch, agg):
    import hgn_amd
    from hgn_amd import parallel
    g = synth.grid_graph(seed=5, nx=7, ny=6, clusters=3 if arch == 'hetero' else 0)
    empty = synth.EdgeSet('world_edges', torch.zeros(0, 4), torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long))
    g = synth.MultiGraph(g.node_features, [g.edge_sets[0], empty] + list(g.edge_sets[1:]))
    sets = [e.name for e in g.edge_sets]
    shapes = O.param_shapes(arch, agg, 2, sets, 5, {e.name: e.features.shape[1] for e in g.edge_sets},
                            g.node_features[1].shape[1] if len(g.node_features) > 1 else 0, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=4)
    N = g.node_features[0].shape[0]
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(1))
    mask = torch.ones(N, dtype=torch.bool); mask[:3] = False
    out_o, loss_o, grads_o, ig_o = H.oracle_run(sd, g, arch, agg, target, mask)
    model = H.hip_model(arch, agg, 2, sets, sd)
    out, loss, grads, ig = H.hip_run(model, g, target, mask)
    assert H.rel_err(out, out_o) <= 1e-5
    for k in grads_o:
        if 'world_edges' in k:                          # no rows: exactly zero, like autograd's
            assert float(grads[k].abs().max()) == 0.0 and float(grads_o[k].abs().max()) == 0.0, k
        elif float(grads_o[k].abs().max()) > 0:
            assert H.rel_err(grads[k], grads_o[k]) <= 2e-5, k
    assert ig['edge']['world_edges'].shape == (0, 4)
    # flat-gradient trainer path (accumulating targets are left untouched by the empty set)
    tr = parallel.DataParallelTrainer(H.hip_model(arch, agg, 2, sets, sd), lr=1e-3)
    G = hgn_amd.MultiGraph([x.cuda() for x in g.node_features],
                           [hgn_amd.EdgeSet(e.name, e.features.cuda(), e.senders.cuda(), e.receivers.cuda()) for e in g.edge_sets])
    l = tr.step(G, target.cuda(), mask.cuda())
    assert abs(float(l) - float(loss_o)) <= 1e-5 * abs(float(loss_o))
    for (k, p) in tr.model.named_parameters():
        if 'world_edges' in k:
            assert float(p.grad.abs().max()) == 0.0, k
        elif float(grads_o[k].abs().max()) > 0:
            assert H.rel_err(p.grad, grads_o[k]) <= 2e-5, k


@pytest.mark.parametrize('arch', ['none', 'hetero', 'hyper', 'multiscale'])
def test_node_latent_gradients_are_shared_not_added_by_autograd(arch):
    """ops.share_grad: the edge blocks add their share of d(h) into the tensor the node update returned (hgn_linear_bwd6a) instead of
    handing autograd a tensor of their own to add -- in the plain stack and, when every edge set reads one node part per side, in the
    two-part schedules (hetero / hyper / multiscale).  Counted on a two-block model; the gradients are compared with the fp64 oracle
    (a shared tensor that the engine replaced by a sum of its own would lose every later contribution: 37 % off when it happened).
    With an EMPTY edge set in the graph (its parts are unknown: the node rows are concatenated for it) nobody shares in the two-part
    schedules -- test_empty_edge_set_forward_backward_vs_oracle[hetero-pna] covers the values."""
    from hgn_amd import ops
    g = synth.grid_graph(seed=9, nx=8, ny=7, clusters=0 if arch == 'none' else 4)
    sets = [e.name for e in g.edge_sets]
    shapes = O.param_shapes(arch, 'pna', 2, sets, 5, {e.name: e.features.shape[1] for e in g.edge_sets},
                            g.node_features[1].shape[1] if len(g.node_features) > 1 else 0, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=6)
    N = g.node_features[0].shape[0]
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(2))
    mask = torch.ones(N, dtype=torch.bool)
    out_o, loss_o, grads_o, _ = H.oracle_run(sd, g, arch, 'pna', target, mask)
    model = H.hip_model(arch, 'pna', 2, sets, sd)
    before = dict(ops.share_stats)
    out, loss, grads, _ = H.hip_run(model, g, target, mask)
    acc, own = ops.share_stats['accumulated'] - before['accumulated'], ops.share_stats['own'] - before['own']
    # every edge block of the processor reports for its node operand(s); only the encoder's outputs (read by the first block) and the
    # operands nobody vouches for may come back as tensors of their own
    assert acc > 0 and acc >= own, (arch, acc, own)
    assert H.rel_err(out, out_o) <= 1e-5
    for k in grads_o:
        if float(grads_o[k].abs().max()) > 0:
            assert H.rel_err(grads[k], grads_o[k]) <= 2e-5, k
    H._REPORT.append({'test': f'test_node_latent_gradients_are_shared_not_added_by_autograd[{arch}]', 'accumulated': acc, 'own_tensor': own})


def test_batcher_on_device_golden_g6_and_ragged_sets():
    """f1 on the device: the vectorised batcher fed CUDA index tensors reproduces golden G6 (reference_compat) bit for bit
    (MeshSimulator.py:159-234), and batches graphs whose per-graph edge counts differ (plate world edges / balance edges)
    like the oracle batcher; the batched ragged graph then runs through the HIP model like the concatenation it is."""
    import hgn_amd
    from hgn_amd import batching, util
    g6 = torch.load(os.path.join(H.GOLDEN, 'g6_get_batched.pt'))
    for B, fx in g6.items():
        graphs = []
        for gi in fx['in']:
            nf = [torch.zeros(n, 1, device='cuda') for n in gi['n']]
            graphs.append(util.MultiGraph(nf, [util.EdgeSet(nm, torch.zeros(s.shape[0], 1, device='cuda'), s.cuda(), r.cuda())
                                               for nm, s, r in gi['sets']]))
        compat = batching.batch_graphs(graphs, reference_compat=True)
        assert [x.shape[0] for x in compat.node_features] == fx['n_out']
        for e, (nm, s, r) in zip(compat.edge_sets, fx['out']):
            assert e.senders.is_cuda and e.name == nm
            assert torch.equal(e.senders.cpu(), s) and torch.equal(e.receivers.cpu(), r), (B, nm)
    # ragged: 3 graphs of one mesh with 4 / 0 / 7 world edges
    gs = []
    for i, nw in enumerate((4, 0, 7)):
        g = synth.grid_graph(seed=20 + i, nx=6, ny=5, world=nw)
        if nw == 0:
            g = synth.MultiGraph(g.node_features, list(g.edge_sets) + [synth.EdgeSet(
                'world_edges', torch.zeros(0, 4), torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long))])
        gs.append(g)
    ora = O.batch_graphs([O.MultiGraph(g.node_features, [O.EdgeSet(*e) for e in g.edge_sets]) for g in gs])
    dev = [util.MultiGraph([x.cuda() for x in g.node_features],
                           [util.EdgeSet(e.name, e.features.cuda(), e.senders.cuda(), e.receivers.cuda()) for e in g.edge_sets]) for g in gs]
    got = batching.batch_graphs(dev)
    for a, b in zip(got.edge_sets, ora.edge_sets):
        assert torch.equal(a.senders.cpu(), b.senders) and torch.equal(a.receivers.cpu(), b.receivers)
        assert torch.equal(a.features.cpu(), b.features)
    sets = [e.name for e in gs[0].edge_sets]
    shapes = O.param_shapes('none', 'sum', 2, sets, 5, {'mesh_edges': 7, 'world_edges': 4}, 0, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=6)
    N = ora.node_features[0].shape[0]
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(2))
    mask = torch.ones(N, dtype=torch.bool)
    out_o, _, grads_o, _ = H.oracle_run(sd, ora, 'none', 'sum', target, mask)
    out, _, grads, _ = H.hip_run(H.hip_model('none', 'sum', 2, sets, sd), got, target, mask)
    assert H.rel_err(out, out_o) <= 1e-5
    assert max(H.rel_err(grads[k], grads_o[k]) for k in grads_o if float(grads_o[k].abs().max()) > 0) <= 2e-5


def test_topology_cache_by_content_and_producer_key_and_graphed_step_cache():
    """The reference's loop builds FRESH index tensors for every batch (MeshSimulator.py:136,159-234): equal index content must
    find the topology built before (no second pair of radix sorts), different content must not; a union of graphs sharing one
    index-tensor pair is found by its producer key without a fingerprint pass; and the captured step keyed on those topologies
    (graphs.GraphedStepCache) trains exactly like eager steps on the same fresh batches."""
    import hgn_amd
    from hgn_amd import batching, graphs, parallel, topology, util
    g = synth.grid_graph(seed=8, nx=9, ny=7)
    e = g.edge_sets[0]
    s, r = e.senders.cuda(), e.receivers.cuda()
    topology.clear_cache()
    base = dict(topology.stats)
    t0 = topology.edge_topology(s, r, 63, s.device)
    t1 = topology.edge_topology(s, r, 63, s.device)                       # same objects
    t2 = topology.edge_topology(s.clone(), r.clone(), 63, s.device)       # fresh objects, equal content
    t3 = topology.edge_topology(e.senders.clone(), e.receivers.clone(), 63, s.device)     # host-resident ids, equal content
    assert t1 is t0 and t2 is t0 and t3 is t0
    r2 = r.clone(); r2[5] = (r2[5] + 1) % 63
    t4 = topology.edge_topology(s.clone(), r2, 63, s.device)              # one id differs
    assert t4 is not t0 and not torch.equal(t4.r.rowptr, t0.r.rowptr)
    assert topology.edge_topology(r.clone(), s.clone(), 63, s.device) is not t0          # roles swapped: another topology
    d = {k: topology.stats[k] - base[k] for k in base}
    assert d == {'object_hits': 1, 'key_hits': 0, 'content_hits': 2, 'builds': 3}, d
    assert t0.r.max_rows == int((t0.r.rowptr[1:] - t0.r.rowptr[:-1]).max())
    with pytest.raises(IndexError):
        topology.edge_topology(s.clone() + 100, r.clone(), 63, s.device)
    # producer key: three frames of one mesh handed out with the SAME index tensors (what the system models do)
    def frame(seed):
        f = synth.grid_graph(seed=seed, nx=9, ny=7)
        return util.MultiGraph([x.cuda() for x in f.node_features], [util.EdgeSet('mesh_edges', f.edge_sets[0].features.cuda(), s, r)])
    b1 = batching.batch_graphs([frame(1), frame(2), frame(3)])
    b2 = batching.batch_graphs([frame(4), frame(5), frame(6)])
    before = dict(topology.stats)
    ta = topology.edge_topology(b1.edge_sets[0].senders, b1.edge_sets[0].receivers, 189, s.device)
    tb = topology.edge_topology(b2.edge_sets[0].senders, b2.edge_sets[0].receivers, 189, s.device)
    assert ta is tb and topology.stats['key_hits'] - before['key_hits'] == 1 and topology.stats['builds'] - before['builds'] == 1
    # captured step in the real loop: fresh tensors every step
    shapes = O.param_shapes('none', 'sum', 2, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=9)
    target = torch.randn(189, 3, generator=torch.Generator().manual_seed(0)).cuda()
    masks = [torch.ones(189, dtype=torch.bool).cuda() for _ in range(3)]
    masks[1][:40] = False
    batches = [batching.batch_graphs([frame(10 + 3 * i + j) for j in range(3)]) for i in range(3)]
    eager = parallel.DataParallelTrainer(H.hip_model('none', 'sum', 2, ['mesh_edges'], sd), lr=1e-3)
    cached = parallel.DataParallelTrainer(H.hip_model('none', 'sum', 2, ['mesh_edges'], sd), lr=1e-3, device_step=True)
    with torch.no_grad():
        cached.model(batches[0])
    cache = graphs.GraphedStepCache(cached)
    for b, m in zip(batches, masks):
        le, lc = eager.step(b, target, m), cache.step(b, target, m)
        assert abs(float(le) - float(lc)) <= 5e-5 * abs(float(le))
    assert cache.captures == 1
    assert H.rel_err(cached.fp.flat, eager.fp.flat) <= 1e-5


# -------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4] shape: cylinder_flow frame -> CylinderModel features (3-dim mesh-edge features, cylinder.py:85-87)
# -> HyperGraphNets, 25 MP layers, + the graph balancer's `balance` edge set, reduced-precision edge / node MLPs.
# The reference cannot run this combination (cylinder + any clustering or balancer fails in its
# normalisers), so the remote and balance sets are constructed here on the frame's positions, and parity is HIP vs the fp64
# oracle: at block level (L = 1) at the usual tolerances, at full depth on the outputs, and the reduced-precision mode against
# its own, separately stated tolerance.
# -------------------------------------------------------------------------------------------------------------
def _config4_graph(nx, ny, K, n_balance, seed):
    from hgn_amd import system_model
    params = {'size': 3, 'aggregation': 'pna', 'message_passing_steps': 1,
              'rmp': {'clustering': 'none', 'connector': 'none', 'num_clusters': K, 'hyper_noise': 'none', 'hyper_node_features': True,
                      'frequency': 1, 'fully_connect': False,
                      'intra_cluster_sampling': {'enabled': False, 'alpha': 0.1, 'spotter_threshold': 0}},
              'graph_balancer': {'algorithm': 'none', 'frequency': 1}}
    fr = synth.cylinder_frame(seed=seed, nx=nx, ny=ny)
    cm = system_model.CylinderModel(params)
    g = cm.build_graph({k: v.cuda() for k, v in fr.items()}, True)            # HIP feature kernels (csrc/features.hip)
    nodes = g.node_features[0].detach().cpu()
    me = g.edge_sets[0]
    assert me.features.shape[1] == 3                                            # cylinder: rel mesh pos (2) + norm
    mesh = synth.EdgeSet('mesh_edges', me.features.detach().cpu(), me.senders.cpu(), me.receivers.cpu())
    return synth.cylinder_remote_sets(nodes, fr['mesh_pos'], mesh, K, n_balance, seed)


@pytest.mark.parametrize('steps', [1, 25])
def test_config4_shape_cylinder_hyper_L25_balance_vs_oracle(steps):
    import hgn_amd
    graph = _config4_graph(nx=24, ny=16, K=12, n_balance=60, seed=21)
    sets = [e.name for e in graph.edge_sets]
    shapes = O.param_shapes('hyper', 'pna', steps, sets, graph.node_features[0].shape[1],
                            {e.name: e.features.shape[1] for e in graph.edge_sets}, graph.node_features[1].shape[1], 3, 128)
    N = graph.node_features[0].shape[0]
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(1))
    mask = torch.ones(N, dtype=torch.bool); mask[:16] = False
    order = ['mesh_edges', 'world_edges', 'inter_cluster', 'inter_cluster_world']
    sd = O.init_state_dict_like(shapes, seed=31)
    out_o, loss_o, grads_o, _ = H.oracle_run(sd, graph, 'hyper', 'pna', target, mask, set_order=order)
    out32, _, grads32, _ = H.oracle_run(sd, graph, 'hyper', 'pna', target, mask, set_order=order, dtype=torch.float32)
    model = H.hip_model('hyper', 'pna', steps, sets, sd, set_order=order)
    out, loss, grads, _, gates, winners = H.hip_run_logged(model, graph, target, mask)
    tid = f'test_config4_shape_cylinder_hyper_L25_balance_vs_oracle[{steps}]'
    r = H.report(tid, 'output (fp32-accurate mode)', out, out_o, out32)
    assert r['norm'] <= TOL_OUT, r
    assert H.rel_err(loss, loss_o) <= TOL_OUT
    # the balance set is encoded and then DROPPED by the hyper block (hypergraphnet.py:54): its processor
    # models receive no gradient, its encoder model does not reach the output either
    for k in grads_o:
        if 'balance' in k:
            assert float(grads_o[k].abs().max()) == 0 and float(grads[k].abs().max()) == 0, k
    wn, we = H.worst_grad(grads, grads_o)
    rn, re_ = H.worst_grad(grads32, grads_o)
    H._REPORT.append({'test': tid, 'what': 'param grads (worst tensor)', 'norm': wn, 'elem': we, 'ref_fp32_norm': rn, 'ref_fp32_elem': re_})
    if steps == 1:
        assert wn <= TOL_GRAD, (wn, rn)                    # block level: the usual gradient tolerance
    else:
        # 25 layers, pna winners / ReLU gates at fp32 rounding (see the 40x40 test): the kink-noise level, or -- where the
        # reference's own fp32 arithmetic sits further from fp64 (measured here: 5.2e-3 for BOTH, the same gate falls the same
        # way in the HIP path and in the reference's fp32) -- twice the reference's distance
        assert wn <= max(3e-3, 2.0 * rn), (wn, rn)
    # the arithmetic itself: the fp64 oracle with the HIP forward's ReLU gates and max / min winners (tests/helpers.py) -- 1e-5 at
    # both depths (at L = 25 the un-transferred figure above is dominated by ONE decision that falls the other way)
    out_g, _, grads_g, gt, wt = H.oracle_run_with_hip_decisions(sd, graph, 'hyper', 'pna', target, mask, gates, winners, set_order=order)
    gn, ge = H.worst_grad(grads, grads_g)
    H._REPORT.append({'test': tid, 'what': 'param grads vs fp64 oracle with the HIP gates / winners (worst tensor)', 'norm': gn, 'elem': ge,
                      'gates': gt.total, 'gates_differing_from_fp64': gt.flipped, 'max_abs_preactivation_at_flip': gt.max_abs_at_flip,
                      'winners': wt.total, 'winners_differing_from_fp64': wt.flipped, 'max_relative_gap_at_flip': wt.max_gap_at_flip})
    print('decision transfer', gt.total, gt.flipped, gt.max_abs_at_flip, wt.total, wt.flipped, wt.max_gap_at_flip, gn, ge)
    assert gt.max_abs_at_flip <= 1e-5 and wt.max_gap_at_flip <= 1e-5, (gt.max_abs_at_flip, wt.max_gap_at_flip)
    assert H.rel_err(out, out_g) <= TOL_OUT
    assert gn <= 1e-5, gn
    # ---- reduced-precision mode (configs[4] "fp16 MFMA edge-MLP"; here ONE bf16 MFMA per product: fp32 range, so the backward
    # needs no loss scaling).  Separately stated tolerance: operands carry 8 significant bits, so 2^-9 = 2e-3 per product;
    # through 3 products x (4 edge sets + 4 node updates) x `steps` layers of residual + LayerNorm the outputs stay within 5e-2
    # of the fp64 result on the tensor's scale (block level: 2e-2; measured 9e-3), the loss within 5e-2.  Not a parity claim.
    hgn_amd.set_matmul_precision('bf16')
    try:
        out_b, loss_b, _, _ = H.hip_run(model, graph, target, mask)
    finally:
        hgn_amd.set_matmul_precision('fp32')
    rb = H.report(tid, 'output (reduced precision: one bf16 product)', out_b, out_o)
    assert rb['norm'] <= (2e-2 if steps == 1 else 5e-2), rb
    assert H.rel_err(loss_b, loss_o) <= 5e-2
    # fp16 forward products (what configs[4] names): 11 significant bits per forward operand instead of 8 -> the outputs are ~8x
    # closer to the fp64 result than in the bf16 mode (bound: a quarter of the bf16 bound, and better than the bf16 run of the same
    # inputs).  The BACKWARD of this mode runs the two-term fp16 products of mode 3 on per-row / per-block scaled operands
    # (csrc/host.cpp: bwd_products): the gradients are the exact derivatives of the reduced-precision forward up to fp32 rounding, so
    # their distance from the fp64 gradients is the forward's rounding carried through the chain rule -- STATED TOLERANCE of the
    # mode: worst parameter-gradient tensor <= 2e-2 at one layer, <= 5e-2 at 25 layers (norm-wise, tensor by tensor).
    hgn_amd.set_matmul_precision('fp16')
    try:
        assert hgn_amd.get_matmul_precision() == 'fp16'
        out_h, loss_h, grads_h, _, gates_h, winners_h = H.hip_run_logged(model, graph, target, mask)
    finally:
        hgn_amd.set_matmul_precision('fp32')
    rh = H.report(tid, 'output (reduced precision: fp16 forward products)', out_h, out_o)
    assert rh['norm'] <= (5e-3 if steps == 1 else 1.25e-2), rh
    assert rh['norm'] < rb['norm'], (rh, rb)
    assert H.rel_err(loss_h, loss_o) <= 2e-2
    # Gradients of the mode.  A forward that is 1e-3 off moves DISCRETE decisions: of the ~25 M ReLU gates and ~5 M pna max / min
    # winners of this instance a few thousand fall the other way than in fp64, and each of them re-routes a gradient (the worst
    # tensors are the small ones fed by a handful of hyper rows) -- that part is a property of any 11-bit forward and is reported,
    # not bounded.  What the kernels answer for is the arithmetic: against the fp64 oracle run WITH this forward's gates and winners
    # (tests/helpers.py) the worst parameter-gradient tensor stays within the mode's stated tolerance.
    wh, _ = H.worst_grad(grads_h, grads_o)
    _, _, grads_hg, gth, wth = H.oracle_run_with_hip_decisions(sd, graph, 'hyper', 'pna', target, mask, gates_h, winners_h, set_order=order)
    whg, _ = H.worst_grad(grads_h, grads_hg)
    H._REPORT.append({'test': tid, 'what': 'param grads (worst tensor), fp16 forward / two-term fp16 backward', 'norm': wh,
                      'norm_with_this_forwards_gates_and_winners': whg, 'gates_differing_from_fp64': gth.flipped,
                      'winners_differing_from_fp64': wth.flipped})
    print('fp16 mode gradients', steps, wh, whg, gth.flipped, wth.flipped)
    assert whg <= (2e-2 if steps == 1 else 5e-2), (whg, wh)
    assert wh < 1.0, wh
    out_again, _, _, _ = H.hip_run(model, graph, target, mask)
    assert torch.equal(out_again, out)                     # switching back restores the fp32-accurate results bit for bit


@pytest.mark.parametrize('agg,nx,ny', [('sum', 7, 5), ('sum', 40, 40), ('sum', 120, 100), ('pna', 23, 17), ('max', 9, 9)])
def test_fused_edge_backward_equals_two_launch_backward(agg, nx, ny):
    """hgn_edge_bwd_fused (data gradients + weight gradients of an edge block in one persistent kernel, dz3 / dz2 never written)
    against the two-launch path it replaces (hgn_mlp_bwd + hgn_mlp_wgrad) on the same inputs: same products, other summation
    order over rows -> 2e-6, and the fused path against the fp64 oracle at the usual tolerances.  `pna` / `max` blocks reach the
    fused kernel through a streaming pre-pass that forms d(e') + the scattered aggregation backward (ops.EdgeBlockFn.backward).
    Sizes: fewer tiles than workgroups, the 146-tile benchmark graph, 1 100 tiles (several per persistent workgroup: the
    loop-carried prefetch), a ragged last tile."""
    import hgn_amd
    from hgn_amd import ops
    graph = synth.grid_graph(seed=3, nx=nx, ny=ny)
    shapes = O.param_shapes('none', agg, 2, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=12)
    N = nx * ny
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(3))
    mask = torch.ones(N, dtype=torch.bool); mask[:3] = False
    model = H.hip_model('none', agg, 2, ['mesh_edges'], sd)
    res = {}
    for fused in (True, False):
        ops.set_fused_edge_backward(fused)
        try:
            ops.prof_reset(); ops.prof_enable(True)
            res[fused] = H.hip_run(model, graph, target, mask)
            k = ops.prof_collect()
        finally:
            ops.prof_enable(False)
            ops.set_fused_edge_backward(None)
        # several aggregates per edge set (pna) / arg-routed ones (max): the gradient reaching e' is formed by hgn_segment_reduce_bwd
        # (d(e') as `base`) and handed to the fused kernel as its d_out -- the same values the two-launch kernel adds in registers
        assert ('edge_bwd_fused' in k) == fused and ('mlp_bwd_edge' in k) == (not fused), sorted(k)
        assert ('seg_bwd' in k) == (fused and agg != 'sum'), sorted(k)
    (out_f, loss_f, g_f, ig_f), (out_u, loss_u, g_u, ig_u) = res[True], res[False]
    assert torch.equal(out_f, out_u)
    for kname in g_u:
        if float(g_u[kname].abs().max()) > 0:
            assert H.rel_err(g_f[kname], g_u[kname]) <= 2e-6, kname
        else:
            assert float(g_f[kname].abs().max()) == 0, kname
    assert H.rel_err(ig_f['node'][0], ig_u['node'][0]) <= 2e-6
    assert H.rel_err(ig_f['edge']['mesh_edges'], ig_u['edge']['mesh_edges']) <= 2e-6
    # bit-reproducible (fixed-order reductions over per-workgroup partials), and within the parity tolerances of the oracle
    ops.set_fused_edge_backward(True)
    try:
        again = H.hip_run(model, graph, target, mask)
    finally:
        ops.set_fused_edge_backward(None)
    assert all(torch.equal(again[2][kname], g_f[kname]) for kname in g_f)
    if agg == 'sum' and nx * ny <= 400:      # (max / min instances need a tie-free seed: covered by test_model_vs_oracle's search)
        out_o, _, g_o, _ = H.oracle_run(sd, graph, 'none', agg, target, mask)
        assert H.rel_err(out_f, out_o) <= TOL_OUT
        assert max(H.rel_err(g_f[kname], g_o[kname]) for kname in g_o if float(g_o[kname].abs().max()) > 0) <= TOL_GRAD


def test_deferred_node_level_weight_gradients_match_immediate_launches():
    """Flat-gradient training queues the node-level weight-gradient tasks (4 per node MLP + 2 per edge block's pre-projection)
    and launches them 16 at a time, the rest when the autograd engine finishes the backward pass: same gradients as launching
    each list on the spot (other chunking of the row sums: 1e-6), in a third of the launches, and complete when backward()
    returns (also inside a captured HIP graph: test_hip_graph_forward_and_train_step_replay runs with the default)."""
    import hgn_amd
    from hgn_amd import ops, parallel
    graph = synth.grid_graph(seed=9, nx=20, ny=15, clusters=4)
    sets = [e.name for e in graph.edge_sets]
    shapes = O.param_shapes('hyper', 'sum', 3, sets, 5, {n: 7 for n in sets}, 8, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=3)
    G = hgn_amd.MultiGraph([x.cuda() for x in graph.node_features],
                           [hgn_amd.EdgeSet(e.name, e.features.cuda(), e.senders.cuda(), e.receivers.cuda()) for e in graph.edge_sets])
    N = 300
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(1)).cuda()
    mask = torch.ones(N, dtype=torch.bool).cuda()
    grads, launches = {}, {}
    for defer in (True, False):
        old = ops._DEFER_NODE_WGRAD
        ops._DEFER_NODE_WGRAD = defer
        try:
            tr = parallel.DataParallelTrainer(H.hip_model('hyper', 'sum', 3, sets, sd), lr=0.0)
            tr.step(G, target, mask)                         # lr = 0: parameters stay, the flat gradient is what we look at
            ops.prof_reset(); ops.prof_enable(True)
            tr.step(G, target, mask)
            k = ops.prof_collect()
        finally:
            ops.prof_enable(False)
            ops._DEFER_NODE_WGRAD = old
        assert not tr.ctx.wq                                 # nothing left behind after backward()
        grads[defer], launches[defer] = tr.fp.grad.clone(), k['wgrad_node']['count']
    assert launches[True] * 2 <= launches[False], launches
    assert H.rel_err(grads[True], grads[False]) <= 1e-6
    _, _, g_o, _ = H.oracle_run(sd, graph, 'hyper', 'sum', target.cpu(), mask.cpu())
    for (kname, p), off in zip(tr.model.named_parameters(), tr.fp.offsets):
        if float(g_o[kname].abs().max()) > 0:
            assert H.rel_err(grads[True][off:off + p.numel()].view(p.shape), g_o[kname]) <= TOL_GRAD, kname


@pytest.mark.parametrize('arch', ['repeated', 'multiscale'])
def test_shared_weight_gradient_targets_through_the_flat_gradient_trainer(arch):
    """A block that applies ONE MLP twice at one row count (RepeatedGraphNet: node and edge model `repetitions` times,
    repeatedgraphnet.py:18-22; MultiScaleGraphNet: node_model_cross and the mesh_edges model twice, multiscalegraphnet.py:20-63)
    hands the deferred queue two accumulating tasks with the same dW / db target.  The reduction adds without atomics, one grid
    slice per task, so the two must not share a launch: gradients through DataParallelTrainer equal the undeferred launches,
    are the same on every run, and match the fp64 oracle."""
    import hgn_amd
    from hgn_amd import ops, parallel
    graph = synth.grid_graph(seed=5, nx=16, ny=12, clusters=4 if arch == 'multiscale' else 0)
    sets = [e.name for e in graph.edge_sets]
    hyper_w = 8 if arch == 'multiscale' else 0
    shapes = O.param_shapes(arch, 'sum', 2, sets, 5, {n: 7 for n in sets}, hyper_w, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=2)
    G = hgn_amd.MultiGraph([x.cuda() for x in graph.node_features],
                           [hgn_amd.EdgeSet(e.name, e.features.cuda(), e.senders.cuda(), e.receivers.cuda()) for e in graph.edge_sets])
    N = graph.node_features[0].shape[0]
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(1))
    mask = torch.ones(N, dtype=torch.bool)
    runs = []
    for defer in (True, True, False):
        old = ops._DEFER_NODE_WGRAD
        ops._DEFER_NODE_WGRAD = defer
        try:
            tr = parallel.DataParallelTrainer(H.hip_model(arch, 'sum', 2, sets, sd), lr=0.0)
            tr.step(G, target.cuda(), mask.cuda())
            tr.step(G, target.cuda(), mask.cuda())
        finally:
            ops._DEFER_NODE_WGRAD = old
        assert not tr.ctx.wq
        runs.append(tr.fp.grad.clone())
    assert torch.equal(runs[0], runs[1])
    assert H.rel_err(runs[0], runs[2]) <= 1e-6
    _, _, g_o, _ = H.oracle_run(sd, graph, arch, 'sum', target, mask)
    n_norm = float(mask.sum()) * 3
    for (kname, p), off in zip(tr.model.named_parameters(), tr.fp.offsets):
        if float(g_o[kname].abs().max()) > 0:
            assert H.rel_err(runs[0][off:off + p.numel()].view(p.shape), g_o[kname]) <= TOL_GRAD, kname


def test_failed_backward_leaves_no_stale_weight_gradient_tasks():
    """A backward pass that raises after node-level tasks were queued drops the engine's final callback: the queued tasks (raw
    pointers into that step's buffers) must never be launched into the next step's gradients, and the next backward must arm
    its own callback -- its gradients are complete when backward() returns."""
    import hgn_amd
    from hgn_amd import ops, parallel
    graph = synth.grid_graph(seed=9, nx=12, ny=9)
    sets = ['mesh_edges']
    shapes = O.param_shapes('none', 'sum', 2, sets, 5, {'mesh_edges': 7}, 0, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=3)
    G = hgn_amd.MultiGraph([x.cuda() for x in graph.node_features],
                           [hgn_amd.EdgeSet(e.name, e.features.cuda(), e.senders.cuda(), e.receivers.cuda()) for e in graph.edge_sets])
    N = graph.node_features[0].shape[0]
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(1)).cuda()
    mask = torch.ones(N, dtype=torch.bool).cuda()
    tr = parallel.DataParallelTrainer(H.hip_model('none', 'sum', 2, sets, sd), lr=0.0)
    tr.step(G, target, mask)
    good = tr.fp.grad.clone()

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError('boom')

    # the encoder runs last in the backward pass: by then the processor's node-level tasks are queued
    tr.fp.zero_grad()
    feats = [Boom.apply(x.requires_grad_(True)) for x in G.node_features]
    out = tr.model(hgn_amd.MultiGraph(feats, G.edge_sets))
    with pytest.raises(RuntimeError, match='boom'):
        ((out - target) * mask.unsqueeze(1)).square().sum().backward()
    assert sum(len(q[0]) for q in tr.ctx.wq.values()) > 0        # the failed run left its tasks behind
    for x in G.node_features:
        x.requires_grad_(False)
    tr.step(G, target, mask)
    assert not tr.ctx.wq
    torch.cuda.synchronize()
    assert torch.equal(tr.fp.grad, good)


def test_two_models_with_different_precisions_interleaved_equal_their_solo_runs():
    """Two models in one process, one fp32-accurate and one in the reduced 'bf16' mode, trained step by step IN TURN: every launch
    carries its model's own precision and flags in its argument struct and its own ops.Context holds queue / packs / workspaces, so
    each model's losses and parameters equal, bit for bit, those of the same model trained alone."""
    import hgn_amd
    from hgn_amd import parallel
    graph = synth.grid_graph(seed=5, nx=14, ny=11, clusters=3)
    sets = [e.name for e in graph.edge_sets]
    shapes = O.param_shapes('hyper', 'pna', 2, sets, 5, {n: 7 for n in sets}, 8, 3, 128)
    G = hgn_amd.MultiGraph([x.cuda() for x in graph.node_features],
                           [hgn_amd.EdgeSet(e.name, e.features.cuda(), e.senders.cuda(), e.receivers.cuda()) for e in graph.edge_sets])
    N = graph.node_features[0].shape[0]
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(2)).cuda()
    mask = torch.ones(N, dtype=torch.bool).cuda()

    def make(seed, mode):
        m = H.hip_model('hyper', 'pna', 2, sets, O.init_state_dict_like(shapes, seed=seed))
        m.set_matmul_precision(mode)
        return parallel.DataParallelTrainer(m, lr=1e-3)

    def solo(seed, mode, steps=3):
        tr = make(seed, mode)
        losses = [tr.step(G, target, mask).clone() for _ in range(steps)]
        return losses, tr.fp.flat.clone()

    solo_a, solo_b = solo(1, None), solo(2, 'bf16')
    ta, tb = make(1, None), make(2, 'bf16')
    la, lb = [], []
    for _ in range(3):
        la.append(ta.step(G, target, mask).clone())
        lb.append(tb.step(G, target, mask).clone())
    assert all(torch.equal(x, y) for x, y in zip(la, solo_a[0])) and torch.equal(ta.fp.flat, solo_a[1])
    assert all(torch.equal(x, y) for x, y in zip(lb, solo_b[0])) and torch.equal(tb.fp.flat, solo_b[1])
    assert not torch.equal(solo_a[0][0], solo_b[0][0])
    assert hgn_amd.get_matmul_precision() == 'fp32'            # the process default was never touched
    # ... and the reduced mode really was in effect for model b only: a's result equals the fp32-accurate run of the same weights
    tc = make(2, None)
    assert not torch.equal(tc.step(G, target, mask), solo_b[0][0])
