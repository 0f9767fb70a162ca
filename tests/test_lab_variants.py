"""Laboratory kernel variants (tools/lab/): measured-and-rejected forms of the edge kernels that the SHIPPED library does not
contain -- two sub-tiles per wave everywhere (HGN_TILE128), the previous weight-gradient kernel (HGN_WGRAD_RESPLIT), 12-wave
workgroups on 192-row tiles (HGN_BIG_TILES), the weight-stationary edge forward (tools/lab/ws_fwd.hip).  These tests keep them
honest (same function as the product kernels) and run only when a laboratory build is present:

    tools/lab/build_lab.sh && HGN_LIB=tools/_build/libhgn_mp_lab.so python -m pytest tests/test_lab_variants.py -m gpu
"""
import ctypes as C
import os

import pytest
import torch

from oracle import mgn_oracle as O
from tests import helpers as H
from tests import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB_LIB = os.environ.get('HGN_LIB', '')


def _is_lab(path):
    try:
        return bool(path) and hasattr(C.CDLL(path), 'hgn_set_ws_fwd')
    except OSError:
        return False


pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not _is_lab(LAB_LIB), reason='needs HGN_LIB=<laboratory build> (tools/lab/build_lab.sh)')]
TOL_OUT, TOL_GRAD = 1e-5, 2e-5


def _set_ws(on):
    from hgn_amd import _lib
    L = _lib.lib()
    L.hgn_set_ws_fwd.argtypes = [C.c_int]
    _lib.check(L.hgn_set_ws_fwd(1 if on else 0), 'hgn_set_ws_fwd')



_VARIANT_SNIPPET = r"""
import sys, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {pkg!r})
from hgn_amd import ops, topology
from tests import synth
from tests.test_gpu_parity import _mlp_sd, _weights
g = synth.grid_graph(seed=5, nx=30, ny=17)
es = g.edge_sets[0]
N, E = 30 * 17, es.senders.shape[0]
topo = topology.EdgeTopology(es.senders, es.receivers, N, torch.device('cuda'))
w, wts = _weights(_mlp_sd(384, 128, True, seed=5), True)
wn, wnts = _weights(_mlp_sd(256, 128, True, seed=6), True)
gen = torch.Generator().manual_seed(9)
h = torch.randn(N, 128, generator=gen).cuda().requires_grad_(True)
e = torch.randn(E, 128, generator=gen).cuda().requires_grad_(True)
y, agg = ops.edge_block(h, e, topo, w, ('sum',))
hn = ops.fused_mlp([h, agg], wn, None, 0)
(hn.square().sum() + y.square().sum()).backward()
torch.save([t.detach().cpu() for t in [y, agg, hn, h.grad, e.grad] + [p.grad for p in wts + wnts]], {out!r})
"""


def test_kernel_variants_behind_environment_switches(tmp_path):
    """The diagnostic kernel variants kept in the library (HGN_TILE128: two sub-tiles per wave; HGN_WGRAD_RESPLIT: the previous
    weight-gradient kernel; HGN_BIG_TILES: 12-wave workgroups on 192-row tiles) compute the same function as the defaults: one child process per setting (the switches are read
    once per process), edge block + node MLP forward / backward compared with the default build of the same inputs."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, 'hyper-graph-nets_amd')
    outs = {}
    for name, env in (('default', {}), ('tile128', {'HGN_TILE128': '1'}), ('resplit', {'HGN_WGRAD_RESPLIT': '1'}), ('big', {'HGN_BIG_TILES': '1', 'HGN_BIG_MIN_ROWS': '1'})):
        out = str(tmp_path / (name + '.pt'))
        code = _VARIANT_SNIPPET.format(root=root, pkg=pkg, out=out)
        r = subprocess.run([sys.executable, '-c', code], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = torch.load(out)
    for a, b in zip(outs['default'], outs['tile128']):
        assert H.rel_err(a, b) <= 1e-6
    for a, b in zip(outs['default'][:5], outs['tile128'][:5]):
        assert torch.equal(a, b)                               # same arithmetic per row, only the tiling differs
    for a, b in zip(outs['default'], outs['resplit']):
        assert H.rel_err(a, b) <= 1e-6
    for a, b in zip(outs['default'], outs['big']):         # 12-wave workgroups on 192-row tiles (forward): the same bits
        assert torch.equal(a, b)


@pytest.mark.parametrize('nx,ny,agg', [(7, 5, 'sum'), (40, 40, 'sum'), (120, 100, 'sum'), (23, 17, 'pna')])
def test_weight_stationary_edge_forward_equals_staged_forward(nx, ny, agg):
    """csrc/ws_fwd.hip (opt-in: weights of the three edge-MLP layers resident in registers, activations through LDS) against the
    staged-weights kernel it can replace, through the same autograd function: outputs, aggregates and every gradient (the
    backward reads the activations / sign words the forward saved) to fp32 rounding -- the first layer adds its bias and gathered
    pre-projections after the products instead of before, so not bit for bit -- and against the fp64 oracle at the usual
    tolerances.  Sizes: fewer tiles than workgroups, more, and a ragged last tile; pna: no in-kernel segment sums."""
    import hgn_amd
    from hgn_amd import ops
    graph = synth.grid_graph(seed=4, nx=nx, ny=ny)
    shapes = O.param_shapes('none', agg, 2, ['mesh_edges'], 5, {'mesh_edges': 7}, 0, 3, 128)
    sd = O.init_state_dict_like(shapes, seed=13)
    N = nx * ny
    target = torch.randn(N, 3, generator=torch.Generator().manual_seed(4))
    mask = torch.ones(N, dtype=torch.bool); mask[:2] = False
    model = H.hip_model('none', agg, 2, ['mesh_edges'], sd)
    res = {}
    for ws in (True, False):
        _set_ws(ws)
        try:
            res[ws] = H.hip_run(model, graph, target, mask)
        finally:
            _set_ws(False)
    (out_w, loss_w, g_w, ig_w), (out_s, loss_s, g_s, ig_s) = res[True], res[False]
    assert H.rel_err(out_w, out_s) <= 2e-6
    for kname in g_s:
        if float(g_s[kname].abs().max()) > 0:
            # (a ReLU sign flips where |z| is at rounding level -- both results are valid -- and a flipped unit changes its
            #  row's gradient by O(1): a handful of rows in 72 000 move a weight gradient by up to 1e-4 of its norm)
            assert H.rel_err(g_w[kname], g_s[kname]) <= 2e-4, kname
    out_o, _, g_o, _ = H.oracle_run(sd, graph, 'none', agg, target, mask) if agg == 'sum' and N <= 400 else (None, None, None, None)
    if out_o is not None:
        assert H.rel_err(out_w, out_o) <= TOL_OUT
        assert max(H.rel_err(g_w[kname], g_o[kname]) for kname in g_o if float(g_o[kname].abs().max()) > 0) <= TOL_GRAD
