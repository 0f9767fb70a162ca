"""ctypes binding of libhgn_mp.so (C ABI: include/hgn_mp.h, include/hgn_features.h).

The library is the product path: if it is missing or a symbol is absent this module raises -- there is no
PyTorch/CPU fallback anywhere in the package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HGN_LIB: laboratory / diagnostic builds of the same C-ABI (tools/lab/build_lab.sh, tools/build_ablations.sh); never set in production
LIB_PATH = os.environ.get('HGN_LIB') or os.path.join(_HERE, 'libhgn_mp.so')

HGN_MAX_SRC = 8
HGN_MAX_ADD = 2
HGN_MAX_WTASK = 16
PACK_BLOCK_BYTES = 98304
HGN_MAX_PACK = 32
NUM_KERNEL_IDS = 15
OP_CODES = {'sum': 0, 'mean': 1, 'max': 2, 'min': 3}
F_FP32_MFMA, F_GENERAL_FWD, F_TILE64_FWD, F_DEFER_LN = 1, 2, 4, 8          # hgn_mlp_fwd_t.flags / hgn_mlp_bwd_t.flags / hgn_wtask_t.flags
KERNEL_NAMES = ['mlp_fwd_edge', 'mlp_fwd', 'mlp_bwd_edge', 'mlp_bwd', 'wgrad', 'seg_fwd', 'seg_bwd', 'linear_fwd',
                'linear_bwd', 'adam', 'csr', 'wgrad_node', 'seg_fwd_agg', 'features', 'edge_bwd_fused']

c_f32p = C.c_void_p      # device pointers travel as integers (tensor.data_ptr())
c_i32p = C.c_void_p


class Src(C.Structure):
    _fields_ = [('x', c_f32p), ('ld', C.c_int64), ('K', C.c_int32), ('idx', c_i32p), ('W', c_f32p), ('Wpk', C.c_void_p)]


class Add(C.Structure):
    _fields_ = [('P', c_f32p), ('ld', C.c_int64), ('idx', c_i32p)]


class MlpFwd(C.Structure):
    _fields_ = [('M', C.c_int64), ('n_src', C.c_int32), ('src', Src * HGN_MAX_SRC), ('n_add', C.c_int32),
                ('add', Add * HGN_MAX_ADD), ('ldw1', C.c_int64), ('b1', c_f32p), ('W2', c_f32p), ('b2', c_f32p),
                ('W3', c_f32p), ('b3', c_f32p), ('out_w', C.c_int32), ('ln_g', c_f32p), ('ln_b', c_f32p),
                ('res', c_f32p), ('ld_res', C.c_int64), ('out', c_f32p), ('ld_out', C.c_int64), ('z1', c_f32p),
                ('z2', c_f32p), ('xhat', c_f32p), ('rstd', c_f32p), ('W2pk', C.c_void_p), ('W3pk', C.c_void_p),
                ('relu_bits', C.c_void_p), ('seg_out', c_f32p), ('ld_seg_out', C.c_int64), ('seg_ids', c_i32p),
                ('post_pk', C.c_void_p * 4), ('n_post', C.c_int32), ('post_out', c_f32p), ('ld_post', C.c_int64),
                ('post_zero', c_f32p), ('ld_post_zero', C.c_int64), ('products', C.c_int32), ('flags', C.c_int32)]


class Dx(C.Structure):
    _fields_ = [('W', c_f32p), ('K', C.c_int32), ('dx', c_f32p), ('ld', C.c_int64), ('residual', C.c_int32),
                ('Wpk_t', C.c_void_p)]


class MlpBwd(C.Structure):
    _fields_ = [('M', C.c_int64), ('d_out', c_f32p), ('ld_dout', C.c_int64), ('out_w', C.c_int32), ('ln_g', c_f32p),
                ('xhat', c_f32p), ('rstd', c_f32p), ('z2', c_f32p), ('z1', c_f32p), ('W3', c_f32p), ('W2', c_f32p),
                ('ldw1', C.c_int64), ('dz3', c_f32p), ('dz2', c_f32p), ('dz1', c_f32p), ('n_dx', C.c_int32),
                ('dx', Dx * HGN_MAX_SRC),
                ('agg_dout', c_f32p), ('ld_agg', C.c_int64), ('n_agg_ops', C.c_int32), ('agg_ops', C.c_int32 * 4),
                ('agg_seg', c_i32p), ('agg_rowptr', c_i32p), ('agg_argmax', c_i32p), ('agg_argmin', c_i32p),
                ('d_gamma', c_f32p), ('d_beta', c_f32p), ('ln_ws', c_f32p), ('ln_accumulate', C.c_int32),
                ('W3pk_t', C.c_void_p), ('W2pk_t', C.c_void_p), ('relu_bits', C.c_void_p),
                ('seg_dz1', c_f32p), ('ld_seg_dz1', C.c_int64), ('seg_ids', c_i32p), ('products', C.c_int32), ('flags', C.c_int32)]


class WTask(C.Structure):
    _fields_ = [('type', C.c_int32), ('A', c_f32p), ('lda', C.c_int64), ('K', C.c_int32), ('idxA', c_i32p),
                ('G', c_f32p), ('ldg', C.c_int64), ('n_out', C.c_int32), ('dW', c_f32p), ('ldw', C.c_int64),
                ('db', c_f32p), ('accumulate', C.c_int32), ('products', C.c_int32), ('flags', C.c_int32)]


class WFuse(C.Structure):
    _fields_ = [('z2', c_f32p), ('z1', c_f32p), ('dW3', c_f32p), ('db3', c_f32p), ('dW2', c_f32p), ('db2', c_f32p),
                ('accumulate', C.c_int32)]


class LnTask(C.Structure):
    _fields_ = [('ln_ws', c_f32p), ('M', C.c_int64), ('d_gamma', c_f32p), ('d_beta', c_f32p), ('accumulate', C.c_int32),
                ('reserved', C.c_int32)]


HGN_MAX_LN_TASK = 48


class WRed(C.Structure):
    _fields_ = [('type', C.c_int32), ('K', C.c_int32), ('n_out', C.c_int32), ('accumulate', C.c_int32), ('n_chunks', C.c_int32),
                ('reserved', C.c_int32), ('dW', c_f32p), ('ldw', C.c_int64), ('db', c_f32p), ('slab', c_f32p), ('chunk_stride', C.c_int64)]


HGN_MAX_WRED = 48


class Pack(C.Structure):
    _fields_ = [('W', c_f32p), ('ldw', C.c_int64), ('n_out', C.c_int32), ('n_in', C.c_int32), ('transposed', C.c_int32),
                ('out', C.c_void_p)]


_SIGS = {
    'hgn_last_error': (C.c_char_p, []),
    'hgn_version': (C.c_int, []),
    'hgn_csr_workspace_bytes': (C.c_int, [C.c_int64, C.c_int64, C.POINTER(C.c_size_t)]),
    'hgn_csr_build': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_size_t, C.POINTER(C.c_int32), C.c_void_p]),
    'hgn_index_fingerprint': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]),
    'hgn_narrow_gather_i64': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'hgn_segment_reduce_fwd': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                         C.POINTER(C.c_int32), C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    'hgn_segment_reduce_bwd': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.POINTER(C.c_int32), C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'hgn_segment_reduce_bwd_sorted': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'hgn_segment_sum_pair': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                       C.c_int64, C.c_void_p]),
    'hgn_segment_std_fwd': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_void_p]),
    'hgn_segment_std_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    'hgn_mlp_fwd': (C.c_int, [C.POINTER(MlpFwd), C.c_void_p]),
    'hgn_pack_bf16x3': (C.c_int, [C.POINTER(Pack), C.c_int, C.c_void_p]),
    'hgn_pack_bf16x3_table': (C.c_int, [C.POINTER(Pack), C.c_void_p, C.c_int, C.c_void_p]),
    'hgn_set_matmul_products': (C.c_int, [C.c_int]),
    'hgn_get_matmul_products': (C.c_int, []),
    'hgn_mlp_fwd6_eligible': (C.c_int, [C.POINTER(MlpFwd)]),
    'hgn_mlp_fwd_post_eligible': (C.c_int, [C.POINTER(MlpFwd)]),
    'hgn_linear_fwd6': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int64,
                                  C.c_int, C.c_void_p]),
    'hgn_linear_fwd6z': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int64,
                                   C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    'hgn_mlp_bwd6_eligible': (C.c_int, [C.POINTER(MlpBwd)]),
    'hgn_linear_bwd6': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int64,
                                  C.c_int, C.c_void_p]),
    'hgn_linear_bwd6a': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int64,
                                   C.c_int, C.c_int, C.c_void_p]),
    'hgn_mlp_bwd_ln_workspace_bytes': (C.c_int, [C.c_int64, C.POINTER(C.c_size_t)]),
    'hgn_mlp_bwd': (C.c_int, [C.POINTER(MlpBwd), C.c_void_p]),
    'hgn_edge_bwd_fused_workspace_bytes': (C.c_int, [C.c_int64, C.POINTER(C.c_size_t)]),
    'hgn_edge_bwd_fused_eligible': (C.c_int, [C.POINTER(MlpBwd)]),
    'hgn_edge_bwd_fused': (C.c_int, [C.POINTER(MlpBwd), C.POINTER(WFuse), C.c_void_p, C.c_size_t, C.c_void_p]),
    'hgn_ln_reduce_batch': (C.c_int, [C.POINTER(LnTask), C.c_int, C.c_void_p]),
    'hgn_mlp_wgrad_partial': (C.c_int, [C.POINTER(WTask), C.c_int, C.c_int64, C.c_void_p, C.c_size_t, C.POINTER(WRed), C.c_void_p]),
    'hgn_edge_bwd_fused_partial': (C.c_int, [C.POINTER(MlpBwd), C.POINTER(WFuse), C.c_void_p, C.c_size_t, C.POINTER(WRed), C.c_void_p]),
    'hgn_slab_reduce_batch': (C.c_int, [C.POINTER(WRed), C.c_int, C.c_void_p]),
    'hgn_wgrad_workspace_bytes': (C.c_int, [C.c_int64, C.c_int, C.POINTER(C.c_size_t)]),
    'hgn_mlp_wgrad': (C.c_int, [C.POINTER(WTask), C.c_int, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    'hgn_linear_fwd': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.c_int, C.c_int64,
                                 C.c_void_p, C.c_int64, C.c_void_p]),
    'hgn_linear_bwd': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.c_int, C.c_int64,
                                 C.c_void_p, C.c_int64, C.c_void_p]),
    'hgn_adam_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                C.c_float, C.c_float, C.c_int32, C.c_float, C.c_void_p]),
    'hgn_adam_step_dev': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                    C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_void_p]),
    # include/hgn_features.h
    'hgn_cells_to_edges_workspace_bytes': (C.c_int, [C.c_int64, C.c_int, C.POINTER(C.c_size_t)]),
    'hgn_cells_to_edges': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64),
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    'hgn_rel_edge_features': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int64,
                                        C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                        C.c_void_p]),
    'hgn_node_features': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    'hgn_col_stats_workspace_bytes': (C.c_int, [C.c_int64, C.c_int, C.POINTER(C.c_size_t)]),
    'hgn_col_stats': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'hgn_normalizer_update': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_float, C.c_void_p]),
    'hgn_normalize': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                C.c_int, C.c_void_p, C.c_void_p]),
    'hgn_radius_edges_workspace_bytes': (C.c_int, [C.c_int64, C.POINTER(C.c_size_t)]),
    'hgn_radius_edges_count': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_float,
                                         C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64),
                                         C.c_void_p, C.c_size_t, C.c_void_p]),
    'hgn_radius_edges_fill': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_float,
                                        C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    'hgn_forman_curvature': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_int64, C.c_void_p, C.c_void_p]),
    'hgn_forman_post_delta': (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int64, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'hgn_lincomb3': (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_int64,
                               C.c_void_p, C.c_void_p]),
    'hgn_prof_enable': (C.c_int, [C.c_int]),
    'hgn_prof_tag': (C.c_int, [C.c_int]),
    'hgn_prof_reset': (C.c_int, []),
    'hgn_prof_collect': (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
}
EXPORTS = tuple(_SIGS)

_lib = None


class HgnError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HgnError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           f'(or `make -C hyper-graph-nets_amd/csrc`). There is no fallback path.')
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)          # AttributeError if the export is missing -> loud
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str = ''):
    if rc != 0:
        msg = lib().hgn_last_error().decode(errors='replace')
        if rc == -3:
            raise IndexError(f'{what}: {msg}')
        raise HgnError(f'{what} failed (code {rc}): {msg}')


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(t):
    if not t.is_cuda:
        raise HgnError('the hgn_amd product path runs on an MI355X (HIP) device only; got a CPU tensor. '
                       'There is no CPU fallback.')


def _nothing():
    return None


class Volatile(tuple):
    """A cache entry hung on a tensor as an attribute (topologies on index tensors, packed weight images on parameters).  Tensor
    attributes travel in pickles (the reference checkpoints with pickle.dump of the whole model, MeshSimulator.py:492-493): an entry
    of this type pickles as None -- the copy rebuilds its caches -- instead of dragging device buffers, or failing on a weakref."""

    def __reduce__(self):
        return (_nothing, ())

    def __deepcopy__(self, memo):
        return None
