"""torch.autograd wrappers over the C ABI (include/hgn_mp.h).  Every arithmetic step of the message-passing path
runs in libhgn_mp.so; torch supplies device memory, the current stream and the autograd graph only.

Reference lines replaced (paths relative to the reference root):
  EdgeBlockFn   GraphNet._update_edge_features           src/migration/graphnet.py:22-32
  AggregateFn   GraphNet.aggregation / util.unsorted_segment_operation   graphnet.py:50-70, src/util.py:92-134
  MLPFn         node updates graphnet.py:34-48,94-124; LazyMLP(+LayerNorm) meshgraphnet.py:53-60,93-108
"""
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import OP_CODES

LAT = 128
_GATE_LOG = None              # tests only: when a list, every training forward appends (first-layer weight ptr, ReLU sign words, row index)
_ARG_LOG = None               # tests only: when a list, every training edge block with max / min aggregates appends (first-layer weight ptr, argmax, argmin, sorted position -> edge index)

_PRODUCTS = {'fp32': 3, 'fp32-f16x2': 3, 'fp32-bf16x3': 6, 'bf16': 1, 'fp16': 2}
_ENV = __import__('os').environ
# Process-wide DEFAULTS, resolved once here (the library itself reads no environment variable): what a Context that does not say
# otherwise follows.  HGN_FP32_MFMA: the plain fp32-MFMA kernels everywhere; HGN_NO_FUSED_BWD: hgn_mlp_bwd + hgn_mlp_wgrad instead of
# hgn_edge_bwd_fused; HGN_NO_EDGE_FWD: the general forward kernel for training edge blocks too.
_DEFAULTS = {'precision': _ENV.get('HGN_PRECISION', 'fp32'), 'fp32_mfma': bool(_ENV.get('HGN_FP32_MFMA')), 'general_fwd': bool(_ENV.get('HGN_NO_EDGE_FWD')),
             'fused_edge_bwd': not bool(_ENV.get('HGN_NO_FUSED_BWD')) and not bool(_ENV.get('HGN_FP32_MFMA'))}
_defaults_epoch = 0           # bumped when a default changes: packed weight images of every context are rebuilt on their next use
_FUSED_SEG_MAX_ROWS = 65      # a segment of <= 65 consecutive rows touches at most two 64-row tiles
_DEFER_NODE_WGRAD = not bool(_ENV.get('HGN_NO_DEFERRED_WGRAD'))
_SEG_PAIR = not bool(_ENV.get('HGN_NO_SEG_PAIR'))      # sender and receiver sums of dz1 in one pass (hgn_segment_sum_pair) instead of two launches


class Context:
    """Everything a model's launches depend on BESIDES their arguments -- precision mode and kernel-selection flags (they travel in
    every argument struct: include/hgn_mp.h `products` / `flags`), the deferred weight-gradient queue, the pack epoch, workspaces, the
    side stream.  One per MeshGraphNet (`model._hgn_ctx`): two models, or two threads, in one process share none of it.  A field left at
    None follows the process default (set_matmul_precision / set_fused_edge_backward / the environment switches above).

    The current context is thread-local (`using`); an autograd Function records the context of its forward and hands it to its
    backward explicitly, because the engine runs backward passes on threads of its own."""

    def __init__(self, precision=None, fused_edge_bwd=None, fp32_mfma=None, general_fwd=None):
        if precision is not None and precision not in _PRODUCTS:
            raise ValueError("matmul precision must be one of " + ', '.join(repr(k) for k in _PRODUCTS))
        self.precision, self.fused_edge_bwd, self.fp32_mfma, self.general_fwd = precision, fused_edge_bwd, fp32_mfma, general_fwd
        self.wgrad_stream = None          # weight-gradient launches go to this side stream (parallel.DataParallelTrainer)
        self.wq = {}                      # (M, device) -> [tasks, tensors kept alive, set of queued dW / db target addresses]
        self.wq_graph_task = -1           # id of the autograd engine run whose final callback will flush the queue (-1: none armed)
        self.flushing = False             # inside flush_wgrad: the launches it makes put their pending sums straight into redq
        self.redq = []                    # deferred chunk-slab sums of weight-gradient launches: (WRed descriptor, workspace kept alive)
        self.lnq = []                     # deferred LayerNorm-affine sums: (workspace, rows, d_gamma, d_beta) of backward calls made with F_DEFER_LN
        self.pack_epoch = 0
        self.pack_in_capture = True       # graphs.GraphedForward keeps the pack launches out of its captured graph
        self.pack_recorder = None         # list that collects the (weights, form) pairs a forward pass packs
        self.pack_plan = None             # PackPlan of a trainer's step: every image the step uses, refreshed by ONE launch
        # gradients of node latents shared between the consumers of ONE tensor inside one backward pass (MLPFn -> EdgeBlockFn):
        # {data_ptr of the forward tensor: (engine run, gradient tensor)}; see share_grad / shared_grad
        self.grad_share = {}
        self.plan_epoch = -1              # pack_epoch at which the plan last ran: packs_of then trusts its cache, also during a capture
        self._plan_rec = None
        self.ws_cache = {}
        self.post_result = None

    # -- what travels in the argument structs ---------------------------------------------------------------------------------
    def mode(self) -> str:
        return self.precision if self.precision is not None else _DEFAULTS['precision']

    def products(self) -> int:
        return _PRODUCTS[self.mode()]

    def flags(self) -> int:
        f32 = self.fp32_mfma if self.fp32_mfma is not None else _DEFAULTS['fp32_mfma']
        gen = self.general_fwd if self.general_fwd is not None else _DEFAULTS['general_fwd']
        return (_lib.F_FP32_MFMA if f32 else 0) | (_lib.F_GENERAL_FWD if gen else 0) | (_lib.F_TILE64_FWD if _ENV.get('HGN_TILE64_FWD') else 0)

    def fp32_only(self) -> bool:
        return bool(self.flags() & _lib.F_FP32_MFMA)

    def fused(self) -> bool:
        return self.fused_edge_bwd if self.fused_edge_bwd is not None else _DEFAULTS['fused_edge_bwd']

    def stamp(self, args) -> None:
        """Fill the per-call options of an MlpFwd / MlpBwd / WTask."""
        args.products, args.flags = self.products(), self.flags()

    # -- state --------------------------------------------------------------------------------------------------------------------
    def set_matmul_precision(self, mode) -> None:
        """This context's own precision ('fp32' / 'bf16' / 'fp16'; None: follow the process default again)."""
        if mode is not None and mode not in _PRODUCTS:
            raise ValueError("matmul precision must be one of " + ', '.join(repr(k) for k in _PRODUCTS))
        self.precision = mode
        self.invalidate_packs()

    def invalidate_packs(self) -> None:
        """Call after the parameters were changed behind torch's back (the flat-buffer Adam kernels): packed images are rebuilt
        on their next use.  In-place torch updates of a parameter are detected through its version counter."""
        self.pack_epoch += 1

    def workspace(self, device, nbytes: int, tag: str = 'wgrad') -> torch.Tensor:
        key = (device.type, device.index, tag)
        ws = self.ws_cache.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
            self.ws_cache[key] = ws
        return ws

    def __getstate__(self):               # a pickled model starts with a fresh queue / cache (settings travel)
        return {'precision': self.precision, 'fused_edge_bwd': self.fused_edge_bwd, 'fp32_mfma': self.fp32_mfma, 'general_fwd': self.general_fwd}

    def __setstate__(self, st):
        self.__init__(**st)


_DEFAULT_CTX = Context()
_tls = __import__('threading').local()


def default_context() -> Context:
    return _DEFAULT_CTX


def current() -> Context:
    stack = getattr(_tls, 'stack', None)
    return stack[-1] if stack else _DEFAULT_CTX


class using:
    """with ops.using(ctx): ...   -- launches issued by this thread inside the block belong to `ctx`."""

    def __init__(self, ctx: Context):
        self.ctx = ctx

    def __enter__(self):
        if not hasattr(_tls, 'stack'):
            _tls.stack = []
        _tls.stack.append(self.ctx)
        return self.ctx

    def __exit__(self, *exc):
        _tls.stack.pop()
        return False


def set_fused_edge_backward(on) -> None:
    """Process default: True: eligible edge-block backwards go through hgn_edge_bwd_fused (the default); False: hgn_mlp_bwd +
    hgn_mlp_wgrad; None: back to the start-up default.  (A Context's own `fused_edge_bwd` overrides it.)"""
    _DEFAULTS['fused_edge_bwd'] = (not bool(_ENV.get('HGN_NO_FUSED_BWD')) and not bool(_ENV.get('HGN_FP32_MFMA'))) if on is None else bool(on)


def set_matmul_precision(mode: str) -> None:
    """Process DEFAULT (contexts that set their own precision are not touched).
    'fp32' (default) = 'fp32-f16x2': every 128x128 product as THREE fp16 MFMAs on operands split into two fp16 terms and scaled by powers
    of two (per row / per packed block / per 32-row block of a weight gradient) -- fp32 accurate, the mode all parity claims refer to.
    'fp32-bf16x3': the same accuracy from six bf16 MFMAs on three bf16 terms per operand (no scales; twice the matrix work: the
    default of rounds 1-4, kept as a cross-check).
    'bf16': ONE bf16 MFMA per product (operands rounded to bf16, fp32 accumulation; ~4e-3 relative error per product).
    'fp16': the FORWARD products as ONE fp16 MFMA (11 significant bits: ~5e-4 per product; activations behind a LayerNorm and
    weights are far inside fp16's range), the backward / weight-gradient products as in 'fp32' (scaled two-term fp16: exact derivatives
    of the reduced-precision forward) -- the "fp16 MFMA edge-MLP" of BASELINE.json configs[4].  Opt-in; the mode also becomes the
    library's default for callers of the C ABI that leave `products` at 0 (include/hgn_mp.h: hgn_set_matmul_products); packed
    weight images are rebuilt on their next use."""
    global _defaults_epoch
    if mode not in _PRODUCTS:
        raise ValueError("matmul precision must be one of " + ', '.join(repr(k) for k in _PRODUCTS))
    _lib.check(_lib.lib().hgn_set_matmul_products(_PRODUCTS[mode]), 'hgn_set_matmul_products')
    _DEFAULTS['precision'] = mode
    _defaults_epoch += 1


def get_matmul_precision() -> str:
    return _DEFAULTS['precision']


def invalidate_packs() -> None:
    """EVERY context's packed images are stale (parameters changed behind torch's version counters, and the caller does not know
    whose): rebuilt on their next use.  A trainer that knows its model uses Context.invalidate_packs."""
    global _defaults_epoch
    _defaults_epoch += 1


def set_wgrad_stream(stream):
    """Run every hgn_mlp_wgrad launch of the current context on `stream` (forked from / joined to the current stream by the caller).

    Weight gradients are only consumed by the optimiser, so they need not sit on the critical path of the backward pass;
    on a second stream they co-run with the next layer's backward kernels, whose memory-bound and MFMA-bound phases they
    fill.  Only valid when the consumer joins the stream before reading gradients (the flat-buffer trainer does)."""
    current().wgrad_stream = stream


def _workspace(device, nbytes: int, tag: str = 'wgrad') -> torch.Tensor:
    """Scratch of the current context (feature kernels, topology builds: callers outside the autograd functions)."""
    return current().workspace(device, nbytes, tag)


def _rowmajor(t: torch.Tensor) -> torch.Tensor:
    """2-D tensor whose rows are contiguous (row stride arbitrary) and 16-byte aligned; copies otherwise."""
    if t.dim() != 2:
        raise ValueError('expected a 2-D tensor')
    if t.dtype != torch.float32:
        t = t.float()
    if t.shape[1] > 0 and t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0) if t.dim() > 1 else 1)


class MLPWeights:
    """The eight tensors of one reference `_make_mlp` module, in nn.Linear layout, plus static shape facts."""
    __slots__ = ('w1', 'b1', 'w2', 'b2', 'w3', 'b3', 'ln_w', 'ln_b')

    def __init__(self, w1, b1, w2, b2, w3, b3, ln_w=None, ln_b=None):
        self.w1, self.b1, self.w2, self.b2, self.w3, self.b3, self.ln_w, self.ln_b = w1, b1, w2, b2, w3, b3, ln_w, ln_b

    def tensors(self):
        ts = [self.w1, self.b1, self.w2, self.b2, self.w3, self.b3]
        if self.ln_w is not None:
            ts += [self.ln_w, self.ln_b]
        return ts

    def check(self):
        if self.w2.shape != (LAT, LAT) or self.w1.shape[0] != LAT or self.w3.shape[1] != LAT:
            raise _lib.HgnError(f'the HIP path is built for latent_size=128, num_layers=2 (reference src/model/flag.py:57-58); '
                                f'got weights {tuple(self.w1.shape)}, {tuple(self.w2.shape)}, {tuple(self.w3.shape)}')
        for t in self.tensors():
            if not t.is_contiguous() or t.dtype != torch.float32:
                raise _lib.HgnError('MLP weights must be contiguous fp32')


# ------------------------------------------------------------------------------------------------------------
# packed weight images for the split-bf16 kernels (include/hgn_mp.h: hgn_pack_bf16x3)
# ------------------------------------------------------------------------------------------------------------
# graphs.GraphedForward (inference: the weights do not change between replays) keeps the pack kernels OUT of its captured graph and
# re-packs eagerly when a version counter or a pack epoch has moved (Context.pack_in_capture / pack_recorder); a captured TRAINING step
# packs inside the graph (its own Adam kernel rewrites the weights every replay).


def pack_signature(pairs, ctx: Optional[Context] = None) -> tuple:
    """What packs_of keys its cache on, for a list of (MLPWeights, transposed) pairs."""
    c = ctx if ctx is not None else current()
    return (_defaults_epoch, c.pack_epoch, c.products()) + tuple(v for w, _ in pairs for v in (w.w1._version, w.w2._version, w.w3._version))


# A captured graph has the ADDRESSES of biases, LayerNorm vectors, the decoder's weights and the packed images baked in.  Whatever
# re-homes parameter storage (parallel.FlatParams, Module.to / .float / .cuda through modules.MeshGraphNet._apply) bumps this epoch;
# graphs.GraphedForward compares it -- and the addresses of the first-layer weights it packed -- on every call and captures again.
_storage_epoch = 0


def storage_moved() -> None:
    global _storage_epoch
    _storage_epoch += 1


def storage_signature(pairs) -> tuple:
    return (_storage_epoch,) + tuple(w.w1.data_ptr() for w, _ in pairs)


def _pack_form(transposed: bool, c: Context) -> int:
    # forward-form images of the fp16 mode carry fp16 bit patterns in their leading third (hgn_pack_t.transposed | 2)
    # ... and those of the scaled two-term fp16 mode two fp16 terms + the block's scale exponent (hgn_pack_t.transposed | 4)
    if transposed:                                   # (mode 2 differentiates with mode 3's products: csrc/host.cpp bwd_products)
        return 5 if c.products() in (2, 3) else 1
    return {2: 2, 3: 4}.get(c.products(), 0)


def _pack_key(w: MLPWeights, t: int, c: Context) -> tuple:
    return (_defaults_epoch, c.pack_epoch, t, w.w1._version, w.w2._version, w.w3._version, w.w1.data_ptr(), w.w2.data_ptr(), w.w3.data_ptr())


def _pack_buffer(w: MLPWeights, transposed: bool) -> torch.Tensor:
    nb1 = (w.w1.shape[1] + LAT - 1) // LAT
    st = getattr(w.w1, '_hgn_pk_t' if transposed else '_hgn_pk', None)
    if st is not None and st[1].numel() == (nb1 + 2) * _lib.PACK_BLOCK_BYTES and st[1].device == w.w1.device:
        return st[1]
    return torch.empty((nb1 + 2) * _lib.PACK_BLOCK_BYTES, dtype=torch.uint8, device=w.w1.device)


def _pack_descs(w: MLPWeights, t: int, buf: torch.Tensor, arr, at: int) -> int:
    """Fill arr[at ...] with the descriptors of [W1 block 0 .. nb1-1, W2, W3] -> their number."""
    nb1 = (w.w1.shape[1] + LAT - 1) // LAT
    for b in range(nb1):
        d = arr[at + b]
        d.W = w.w1.data_ptr() + 4 * LAT * b; d.ldw = w.w1.shape[1]; d.n_out = LAT
        d.n_in = min(LAT, w.w1.shape[1] - LAT * b); d.transposed = t
        d.out = buf.data_ptr() + b * _lib.PACK_BLOCK_BYTES
    for i, m in enumerate((w.w2, w.w3)):
        d = arr[at + nb1 + i]
        d.W = m.data_ptr(); d.ldw = LAT; d.n_out = m.shape[0]; d.n_in = LAT; d.transposed = t
        d.out = buf.data_ptr() + (nb1 + i) * _lib.PACK_BLOCK_BYTES
    return nb1 + 2


def _pack_cache(w: MLPWeights, transposed: bool, key, buf) -> None:
    try:
        setattr(w.w1, '_hgn_pk_t' if transposed else '_hgn_pk', _lib.Volatile((key, buf)))
    except Exception:
        pass


def packs_of(w: MLPWeights, transposed: bool = False, ctx: Optional[Context] = None):
    """-> uint8 tensor holding the packed blocks [W1 block 0 .. nb1-1, W2, W3] of this MLP (forward or transposed form; a
    first-layer width that is not a multiple of 128 gives a zero-padded last block), or None for a narrow output (decoder)."""
    if w.w3.shape[0] != LAT or not w.w1.is_cuda:
        return None
    c = ctx if ctx is not None else current()
    nb1 = (w.w1.shape[1] + LAT - 1) // LAT
    t = _pack_form(transposed, c)
    key = _pack_key(w, t, c)
    st = getattr(w.w1, '_hgn_pk_t' if transposed else '_hgn_pk', None)
    if c.pack_recorder is not None:
        c.pack_recorder.append((w, transposed))
    # (a captured training step packs inside its graph -- unless its PackPlan already did, for this very epoch)
    capturing = torch.cuda.is_current_stream_capturing() and c.pack_in_capture and c.plan_epoch != c.pack_epoch
    if st is not None and st[0] == key and not capturing:
        return st[1]
    buf = _pack_buffer(w, transposed)
    arr = (_lib.Pack * (nb1 + 2))()
    n = _pack_descs(w, t, buf, arr, 0)
    for i0 in range(0, n, _lib.HGN_MAX_PACK):
        cnt = min(_lib.HGN_MAX_PACK, n - i0)
        _lib.check(_lib.lib().hgn_pack_bf16x3(C.cast(C.byref(arr, i0 * C.sizeof(_lib.Pack)), C.POINTER(_lib.Pack)), cnt,
                                              _lib.stream_ptr()), 'hgn_pack_bf16x3')
    _pack_cache(w, transposed, key, buf)
    return buf


class PackPlan:
    """Every packed image one training step uses -- recorded during a first step (Context.pack_recorder) -- refreshed by ONE launch
    over a descriptor table in device memory (include/hgn_mp.h: hgn_pack_bf16x3_table) instead of one launch per MLP and form:
    64 launches of ~4.5 us for the 15-block model, a tenth of a one-graph step.  The table is (re)built eagerly whenever an address
    or the precision changed; a capture that finds it stale leaves the packing to packs_of."""

    def __init__(self, pairs):
        seen, self.items = set(), []
        for w, transposed in pairs:
            k = (w.w1.data_ptr(), bool(transposed))
            if w.w3.shape[0] == LAT and w.w1.is_cuda and k not in seen:
                seen.add(k)
                self.items.append((w, bool(transposed)))
        self.sig = self.host = self.table = self.bufs = None
        self.generation = 0               # bumped whenever the table's CONTENT (an address, the precision) changed
        self.retired = []                 # earlier tables / image buffers: kept alive with the plan

    def _signature(self, c: Context) -> tuple:
        return (c.products(), _storage_epoch) + tuple(p for w, _ in self.items for p in (w.w1.data_ptr(), w.w2.data_ptr(), w.w3.data_ptr()))

    def prepare(self, c: Context) -> bool:
        """Build the descriptor table if it is stale (eagerly: a host-to-device copy) -> whether launch() can run."""
        if not self.items:
            return False
        sig = self._signature(c)
        if sig != self.sig:
            if torch.cuda.is_current_stream_capturing():
                return False
            bufs = [_pack_buffer(w, tr) for w, tr in self.items]
            n = sum((w.w1.shape[1] + LAT - 1) // LAT + 2 for w, _ in self.items)
            host = (_lib.Pack * n)()
            at = 0
            for (w, tr), buf in zip(self.items, bufs):
                at += _pack_descs(w, _pack_form(tr, c), buf, host, at)
            raw = bytes(host)
            dev = self.items[0][0].w1.device
            # A captured step (graphs.GraphedTrainStep / GraphedShardStep) has THIS table's device address baked into its pack
            # launch: the table is one allocation per plan for the plan's lifetime, rewritten in place (a stream-ordered copy) when
            # the descriptor bytes changed.  A bumped storage epoch with unchanged addresses (another model moved) leaves the bytes
            # equal: nothing is copied, nothing is freed.  `generation` counts the content changes, so that a captured step can tell
            # that the addresses IT baked in (weights, packed images) are no longer the ones the table describes.
            if self.table is not None and self.table.numel() == len(raw) and self.table.device == dev:
                if self.host is None or bytes(self.host) != raw:
                    self.table.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8), non_blocking=False)
                    self.generation += 1
            else:
                if self.table is not None:
                    self.retired.append((self.table, self.bufs))     # (never dropped while a graph may still replay against them)
                self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
                self.generation += 1
            self.bufs, self.host = bufs, host
            self.sig = sig
        return True

    def launch(self, c: Context) -> bool:
        if not self.prepare(c):
            return False
        _lib.check(_lib.lib().hgn_pack_bf16x3_table(self.host, self.table.data_ptr(), len(self.host), _lib.stream_ptr()),
                   'hgn_pack_bf16x3_table')
        for (w, tr), buf in zip(self.items, self.bufs):
            _pack_cache(w, tr, _pack_key(w, _pack_form(tr, c), c), buf)
        c.plan_epoch = c.pack_epoch
        return True


def begin_step_packs(c: Context) -> None:
    """First thing of a trainer's step: refresh every packed image with one launch (from the second step on; the first one records)."""
    if c.pack_plan is not None:
        c.pack_plan.launch(c)
    elif c.pack_recorder is None and not torch.cuda.is_current_stream_capturing():
        c._plan_rec = c.pack_recorder = []


def end_step_packs(c: Context) -> None:
    rec = c._plan_rec
    if rec is not None:
        if c.pack_recorder is rec:
            c.pack_recorder = None
        c._plan_rec = None
        c.pack_plan = PackPlan(rec) if rec else None


def _fill_common_fwd(a: _lib.MlpFwd, w: MLPWeights, out, res, saves):
    a.ldw1 = w.w1.shape[1]
    a.b1 = w.b1.data_ptr()
    a.W2 = w.w2.data_ptr(); a.b2 = w.b2.data_ptr()
    a.W3 = w.w3.data_ptr(); a.b3 = w.b3.data_ptr()
    a.out_w = w.w3.shape[0]
    if w.ln_w is not None:
        a.ln_g = w.ln_w.data_ptr(); a.ln_b = w.ln_b.data_ptr()
    if res is not None:
        a.res = res.data_ptr(); a.ld_res = _ld(res)
    a.out = out.data_ptr(); a.ld_out = _ld(out)
    if saves is not None:
        z1, z2, xhat, rstd, bits = saves
        a.z1 = z1.data_ptr(); a.z2 = z2.data_ptr(); a.relu_bits = bits.data_ptr()
        if xhat is not None:
            a.xhat = xhat.data_ptr(); a.rstd = rstd.data_ptr()


def _alloc_saves(M, has_ln, dev):
    z1 = torch.empty(M, LAT, device=dev)
    z2 = torch.empty(M, LAT, device=dev)
    xhat = torch.empty(M, LAT, device=dev) if has_ln else None
    rstd = torch.empty(M, device=dev) if has_ln else None
    bits = torch.empty(M, 8, dtype=torch.int32, device=dev)      # ReLU sign patterns: what the backward chain reads instead of z1 / z2
    return z1, z2, xhat, rstd, bits


def _zero_unaccumulated(bufs, accs):
    """Rows == 0 (an empty edge set, e.g. plate `world_edges` with no obstacle in range): nothing to reduce.  Gradient buffers
    that would have been OVERWRITTEN by the launch become zero; accumulating (flat-buffer) targets are left untouched."""
    for b, acc in zip(bufs, accs):
        if not acc:
            b.zero_()


# Deferred node-level weight gradients.  A node MLP hands over 4 tasks of N rows, the pre-projection of an edge block 2 more:
# launched one by one that is ~33 launches per step of a kernel whose fixed costs (slab reduction, tail of a 200 k-row grid)
# are a third of its time.  Weight gradients are consumed by nobody before the optimiser, so tasks that ACCUMULATE into a flat
# gradient buffer (parallel.FlatParams) are queued per row count (Context.wq) and launched 16 at a time; whatever is left goes out
# when the autograd engine finishes the backward pass (queue_callback), i.e. before `backward()` returns -- also under HIP-graph capture.
def _graph_task_id() -> int:
    try:
        return int(torch._C._current_graph_task_id())
    except AttributeError:                    # an older torch: treated as "not inside an engine run"
        return -1


def flush_wgrad(ctx: Optional[Context] = None) -> None:
    """Launch every queued weight-gradient task of the context now."""
    c = ctx if ctx is not None else current()
    c.wq_graph_task = -1
    c.flushing = True
    try:
        for (M, dev), q in list(c.wq.items()):
            if q[0]:
                _run_wgrad_here(c, q[0], M, dev, False)
        L = _lib.lib()
        # pending chunk-slab sums (the queued launches above have just added theirs): batches of distinct targets, in queue order
        batch, seen = [], set()

        def go():
            if batch:
                arr = (_lib.WRed * len(batch))(*batch)
                _lib.check(L.hgn_slab_reduce_batch(arr, len(batch), _lib.stream_ptr()), 'hgn_slab_reduce_batch')
            batch.clear(); seen.clear()
        for r, _ws in c.redq:
            tg = [p_ for p_ in (r.dW, r.db) if p_]
            if len(batch) == _lib.HGN_MAX_WRED or any(p_ in seen for p_ in tg):
                go()
            batch.append(r); seen.update(tg)
        go()
        chunk, seen_ln = [], set()

        def go_ln():
            if chunk:
                arr = (_lib.LnTask * len(chunk))()
                for t, (ws, M, dg, db) in zip(arr, chunk):
                    t.ln_ws = ws.data_ptr(); t.M = M; t.d_gamma = dg.data_ptr(); t.d_beta = db.data_ptr(); t.accumulate = 1
                _lib.check(L.hgn_ln_reduce_batch(arr, len(chunk), _lib.stream_ptr()), 'hgn_ln_reduce_batch')
            chunk.clear(); seen_ln.clear()
        for ent in c.lnq:                   # (one LayerNorm applied twice -- `repeated` blocks -- : two launches, in queue order)
            if len(chunk) == _lib.HGN_MAX_LN_TASK or ent[2].data_ptr() in seen_ln:
                go_ln()
            chunk.append(ent); seen_ln.add(ent[2].data_ptr())
        go_ln()
    finally:
        c.flushing = False
        c.wq.clear()
        c.lnq.clear()
        c.redq.clear()


def discard_stale_wgrad(ctx: Optional[Context] = None) -> int:
    """Drop queued tasks without launching them and disarm the callback.  Tasks can only be left over when a backward pass
    raised after queueing them (the engine then drops its callbacks): their operand pointers belong to that failed step and
    must not be launched into the next step's gradient buffer.  -> number of tasks dropped.  Called at the start of every
    trainer step (parallel.DataParallelTrainer, graphs.*) and whenever a task is queued from a different engine run."""
    c = ctx if ctx is not None else current()
    n = sum(len(q[0]) for q in c.wq.values()) + len(c.lnq) + len(c.redq)
    c.wq.clear()
    c.lnq.clear()
    c.redq.clear()
    c.wq_graph_task = -1
    return n


def _wtask_targets(t):
    return [p for p in (t.dW, t.db) if p]


def _arm_flush(c: Context, gid: int) -> None:
    if gid != c.wq_graph_task:
        # first deferred piece of THIS engine run.  Anything still queued was left by a run that raised: never launch it.
        discard_stale_wgrad(c)
        torch.autograd.Variable._execution_engine.queue_callback(lambda: flush_wgrad(c))
        c.wq_graph_task = gid


_DEFER_LN = not bool(_ENV.get('HGN_NO_DEFER_LN'))


def _ln_defer(c: Context, b, M: int, dev, dg: torch.Tensor, db: torch.Tensor, accumulate) -> bool:
    """LayerNorm-affine gradients that ACCUMULATE into a flat gradient buffer are read by nobody before the optimiser: the backward call
    leaves its partial slabs in a workspace of its own (F_DEFER_LN) and ONE launch sums the slabs of all calls when the engine run
    ends (flush_wgrad: 31 reductions of 7-9 us per step of the 15-layer model otherwise).  -> whether the call was set up that way."""
    gid = _graph_task_id()
    if not (_DEFER_LN and accumulate and gid >= 0 and M > 0 and c.wgrad_stream is None):
        return False
    _arm_flush(c, gid)
    nb = C.c_size_t(0)
    _lib.check(_lib.lib().hgn_mlp_bwd_ln_workspace_bytes(M, C.byref(nb)), 'hgn_mlp_bwd_ln_workspace_bytes')
    ws = torch.empty((nb.value + 3) // 4, dtype=torch.float32, device=dev)
    b.ln_ws = ws.data_ptr()
    b.flags |= _lib.F_DEFER_LN
    c.lnq.append((ws, M, dg, db))
    return True


def _defer_wgrad(c: Context, tasks, M, dev, keep):
    gid = _graph_task_id()
    if gid < 0:                              # backward driven by hand, outside an engine run: nothing will call back
        _run_wgrad_here(c, tasks, M, dev, False)
        return
    _arm_flush(c, gid)
    q = c.wq.setdefault((M, dev), [[], [], set()])
    tg = [p for t in tasks for p in _wtask_targets(t)]
    # wgrad_reduce_kernel adds one task's result onto its target without atomics, one grid slice per task: two tasks of ONE
    # launch must never share a dW / db target (the same MLP applied twice at one row count: `repeated` blocks, the cross
    # model and the mesh-edge model of `multiscale`).  Launch what is queued first; stream order then serialises the two.
    if len(q[0]) + len(tasks) > _lib.HGN_MAX_WTASK or any(p in q[2] for p in tg) or len(set(tg)) != len(tg):
        if q[0]:
            _run_wgrad_here(c, q[0], M, dev, False)
        q[0], q[1], q[2] = [], [], set()
    if len(set(tg)) != len(tg):              # the hand-over itself repeats a target: one launch per task
        for t in tasks:
            _run_wgrad_here(c, [t], M, dev, False)
        return
    q[0] += tasks
    q[1] += list(keep)
    q[2].update(tg)


def _run_wgrad(c: Context, tasks: List[_lib.WTask], M: int, dev, edge_level: bool = False, keep=(), defer: bool = False):
    if M == 0:          # operands of an empty set have null data pointers; the callers zero what the launch would have written
        return
    for t in tasks:
        c.stamp(t)
    side = c.wgrad_stream
    if defer and _DEFER_NODE_WGRAD and side is None and all(t.accumulate for t in tasks):
        _defer_wgrad(c, tasks, M, dev, keep)
        return
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())
        for t in keep:                       # operands were allocated on the main stream: keep them alive for the side stream
            t.record_stream(side)
        with torch.cuda.stream(side):
            _run_wgrad_here(c, tasks, M, dev, edge_level)
        return
    _run_wgrad_here(c, tasks, M, dev, edge_level)


_DEFER_WRED = not bool(_ENV.get('HGN_NO_DEFER_WRED'))


def _wred_deferrable(c: Context, accumulate) -> bool:
    """Weight gradients that ACCUMULATE into a flat gradient buffer are read by nobody before the optimiser: the launch leaves its
    chunk slabs in a workspace of its own and ONE launch per <= 48 pending sums adds them when the engine run ends (flush_wgrad;
    39 reductions of 6-8 us per step of the 15-layer model otherwise)."""
    if not (_DEFER_WRED and accumulate and c.wgrad_stream is None):
        return False
    if c.flushing:                          # a queued launch going out from flush_wgrad: its sums are taken a few lines further down
        return True
    gid = _graph_task_id()
    if gid < 0:
        return False
    _arm_flush(c, gid)
    return True


def _run_wgrad_here(c: Context, tasks: List[_lib.WTask], M: int, dev, edge_level: bool):
    L = _lib.lib()
    L.hgn_prof_tag(0 if edge_level else 1)
    for i in range(0, len(tasks), _lib.HGN_MAX_WTASK):
        chunk = tasks[i:i + _lib.HGN_MAX_WTASK]
        arr = (_lib.WTask * len(chunk))(*chunk)
        nb = C.c_size_t(0)
        _lib.check(L.hgn_wgrad_workspace_bytes(M, len(chunk), C.byref(nb)), 'hgn_wgrad_workspace_bytes')
        if _wred_deferrable(c, all(t.accumulate for t in chunk)):
            ws = torch.empty((nb.value + 3) // 4, dtype=torch.float32, device=dev)
            red = (_lib.WRed * len(chunk))()
            _lib.check(L.hgn_mlp_wgrad_partial(arr, len(chunk), M, ws.data_ptr(), 4 * ws.numel(), red, _lib.stream_ptr()), 'hgn_mlp_wgrad_partial')
            c.redq.extend((_lib.WRed.from_buffer_copy(r), ws) for r in red)
            continue
        ws = c.workspace(dev, nb.value)
        _lib.check(L.hgn_mlp_wgrad(arr, len(chunk), M, ws.data_ptr(), ws.numel(), _lib.stream_ptr()), 'hgn_mlp_wgrad')


def _ln_workspace(c: Context, M: int, dev) -> torch.Tensor:
    nb = C.c_size_t(0)
    _lib.check(_lib.lib().hgn_mlp_bwd_ln_workspace_bytes(M, C.byref(nb)), 'hgn_mlp_bwd_ln_workspace_bytes')
    return c.workspace(dev, nb.value, 'ln')


def _grad_targets(wt):
    """Flat-gradient mode (parallel.FlatParams): a parameter tagged with ``_hgn_grad`` receives its gradient by
    ACCUMULATION straight into that buffer (zeroed once per step) and autograd gets None for it -- no per-parameter
    add kernels.  Untagged parameters get a fresh tensor returned to autograd.  The tag only counts while it IS the parameter's
    .grad storage: a pickled / reloaded parameter carries a copy of the attribute that belongs to no trainer."""
    out = []
    for t in wt:
        tg = getattr(t, '_hgn_grad', None)
        if tg is not None and (t.grad is None or t.grad.data_ptr() != tg.data_ptr()):
            tg = None
        out.append(tg)
    return out


def _grad_bufs(wt, targets):
    """-> (buffers to write, accumulate flags, values to return to autograd)"""
    bufs, accs, rets = [], [], []
    for t, tg in zip(wt, targets):
        if tg is not None:
            bufs.append(tg); accs.append(1); rets.append(None)
        else:
            b = torch.empty_like(t)
            bufs.append(b); accs.append(0); rets.append(b)
    return bufs, accs, rets


def _wtask(typ, A, lda, K, idxA, G, ldg, n_out, dW_ptr, ldw, db_ptr, acc=0):
    t = _lib.WTask()
    t.accumulate = acc
    t.type = typ; t.A = A; t.lda = lda; t.K = K; t.idxA = idxA; t.G = G; t.ldg = ldg
    t.n_out = n_out; t.dW = dW_ptr; t.ldw = ldw; t.db = db_ptr
    return t


# ------------------------------------------------------------------------------------------------------------
# generic fused MLP over concatenated sources
# ------------------------------------------------------------------------------------------------------------
_SHARE_GRADS = not bool(_ENV.get('HGN_NO_SHARED_DH'))
share_stats = {'accumulated': 0, 'own': 0}      # edge-block gradients for a node latent: added into the node update's tensor / returned as a tensor of their own


def _share_table(c: Context):
    """The table of the CURRENT engine run (entries of an earlier run belong to nobody)."""
    gid = _graph_task_id()
    if gid < 0 or not _SHARE_GRADS:
        return None
    if c.grad_share.get('run') != gid:
        c.grad_share.clear()
        c.grad_share['run'] = gid
    return c.grad_share


def share_grad(c: Context, src: torch.Tensor, grad: torch.Tensor) -> None:
    """A node latent h is read by the node update (MLPFn source) AND by the edge blocks (EdgeBlockFn: graphnet.py:25-26, 43-47); the
    engine would add the two gradients with a pass of its own over [N, 128] (15 launches of 46 us per step of the headline
    model).  The node update's backward runs first (it comes later in the forward): it leaves the gradient tensor it returns for
    `src` here, keyed by the forward tensor's address; an edge block that finds the entry for ITS h accumulates into that tensor
    (hgn_linear_bwd6a) and returns no gradient of its own.  The engine holds the FIRST gradient it is handed for a tensor by
    reference until every consumer has reported, so what it passes on is the finished sum -- as long as nobody hands it a second
    tensor for the same h: it would then add the two into a buffer of its own and the shared tensor would be a dead copy.  Hence:
    an entry is made only by the first of our functions to report for a tensor, and whoever returns a gradient tensor of its own
    for a tensor closes its entry for the rest of the run (`unshare_grad`)."""
    t = _share_table(c)
    if t is None:
        return
    k = src.data_ptr()
    t[k] = None if k in t else (grad, tuple(src.shape))      # a second reporter: the engine adds -- nothing to share any more


def unshare_grad(c: Context, src: torch.Tensor) -> None:
    t = _share_table(c)
    if t is not None:
        t[src.data_ptr()] = None


def shared_grad(c: Context, src: torch.Tensor):
    t = _share_table(c)
    ent = t.get(src.data_ptr()) if t is not None else None
    if ent is None or ent[1] != tuple(src.shape) or ent[0].shape != src.shape or not ent[0].is_contiguous():
        return None
    return ent[0]


class MLPFn(torch.autograd.Function):
    """out = [src[residual] +] [LN](MLP(cat(src_0[idx_0], src_1[idx_1], ...)))   without materialising the cat."""

    @staticmethod
    def forward(ctx, meta, *tensors):
        n_src, idxs, residual, has_ln, train = meta[:5]
        post = meta[5] if len(meta) > 5 else None          # (packed next-block weights, zero-fill wanted), see fused_mlp
        given = meta[6] if len(meta) > 6 else None         # first W1 column of every source (fused_mlp: cols), or None: back to back
        share = bool(meta[7]) if len(meta) > 7 else False   # the caller vouches for the consumers of source 0 (share_grad)
        c = current()
        srcs = [_rowmajor(t) for t in tensors[:n_src]]
        wt = tensors[n_src:]
        w = MLPWeights(*wt)
        w.check()
        dev = wt[0].device
        for s in srcs:
            _lib.require_gpu(s)
        M = idxs[0].shape[0] if idxs[0] is not None else srcs[0].shape[0]
        out_w = w.w3.shape[0]
        out = torch.empty(M, out_w, device=dev)
        a = _lib.MlpFwd()
        c.stamp(a)
        a.M = M
        a.n_src = n_src
        col = 0
        cols = []
        # every source must start on a 128-column boundary of W1 (all but the last a multiple of 128 wide)
        if given is not None:
            pk = packs_of(w, ctx=c) if all(g % LAT == 0 for g in given) else None
        else:
            pk = packs_of(w, ctx=c) if all(s.shape[1] % LAT == 0 for s in srcs[:-1]) else None
        nb1 = (w.w1.shape[1] + LAT - 1) // LAT
        for i, s in enumerate(srcs):
            if given is not None:
                if given[i] < col or given[i] + s.shape[1] > w.w1.shape[1]:
                    raise _lib.HgnError(f'MLP source {i}: columns [{given[i]}, {given[i] + s.shape[1]}) overlap the source before or exceed '
                                        f'in_features {w.w1.shape[1]}')
                col = given[i]
            e = a.src[i]
            e.x = s.data_ptr(); e.ld = _ld(s); e.K = s.shape[1]
            e.idx = idxs[i].data_ptr() if idxs[i] is not None else None
            e.W = w.w1.data_ptr() + 4 * col
            if pk is not None:
                e.Wpk = pk.data_ptr() + (col // LAT) * _lib.PACK_BLOCK_BYTES
            cols.append(col)
            col += s.shape[1]
        if pk is not None:
            a.W2pk = pk.data_ptr() + nb1 * _lib.PACK_BLOCK_BYTES
            a.W3pk = pk.data_ptr() + (nb1 + 1) * _lib.PACK_BLOCK_BYTES
        if given is None and col != w.w1.shape[1]:
            raise _lib.HgnError(f'MLP input width {col} does not match weight in_features {w.w1.shape[1]}')
        saves = _alloc_saves(M, has_ln, dev) if train else None
        res = srcs[residual] if residual >= 0 else None
        _fill_common_fwd(a, w, out, res, saves)
        if post is not None:
            pk_next, want_zero = post
            P = torch.empty(M, 2 * LAT, device=dev)
            zero = torch.empty(M, LAT, device=dev) if want_zero else None
            a.n_post = 2; a.post_out = P.data_ptr(); a.ld_post = 2 * LAT
            a.post_pk[0] = pk_next.data_ptr(); a.post_pk[1] = pk_next.data_ptr() + _lib.PACK_BLOCK_BYTES
            if zero is not None:
                a.post_zero = zero.data_ptr(); a.ld_post_zero = LAT
            if M > 0 and _lib.lib().hgn_mlp_fwd_post_eligible(C.byref(a)):
                c.post_result = (P, zero)
            else:                                            # not a split-product launch: the pre-projection stays a launch of its own
                a.n_post = 0; a.post_out = None; a.post_zero = None
                c.post_result = None
        if M > 0:
            _lib.check(_lib.lib().hgn_mlp_fwd(C.byref(a), _lib.stream_ptr()), 'hgn_mlp_fwd')
        if train and _GATE_LOG is not None:
            _GATE_LOG.append((wt[0].data_ptr(), saves[4], idxs[0]))
        if train:
            ctx.meta = (n_src, idxs, residual, has_ln, cols, M)
            ctx.gaps = sum(s.shape[1] for s in srcs) != w.w1.shape[1]      # W1 columns no source feeds: their gradient is zero
            ctx.share = share
            ctx.saves = saves
            ctx.targets = _grad_targets(wt)
            ctx.pk_t = packs_of(w, transposed=True, ctx=c) if pk is not None else None
            ctx.hgn = c
            ctx.save_for_backward(*srcs, *wt)
        return out

    @staticmethod
    def backward(ctx, d_out):
        n_src, idxs, residual, has_ln, cols, M = ctx.meta
        c = ctx.hgn                                      # (the engine runs this on a thread of its own: the context comes with the node)
        pk_t = ctx.pk_t
        saved = ctx.saved_tensors
        srcs, wt = saved[:n_src], saved[n_src:]
        w = MLPWeights(*wt)
        z1, z2, xhat, rstd, bits = ctx.saves
        dev = d_out.device
        d_out = _rowmajor(d_out)
        L = _lib.lib()
        out_w = w.w3.shape[0]
        dz3 = torch.empty(M, LAT, device=dev)
        dz2 = torch.empty(M, LAT, device=dev)
        dz1 = torch.empty(M, LAT, device=dev)
        b = _lib.MlpBwd()
        c.stamp(b)
        b.M = M
        b.d_out = d_out.data_ptr(); b.ld_dout = _ld(d_out); b.out_w = out_w
        if has_ln:
            b.ln_g = w.ln_w.data_ptr(); b.xhat = xhat.data_ptr(); b.rstd = rstd.data_ptr()
        b.z2 = z2.data_ptr(); b.z1 = z1.data_ptr(); b.relu_bits = bits.data_ptr()
        b.W3 = w.w3.data_ptr(); b.W2 = w.w2.data_ptr(); b.ldw1 = w.w1.shape[1]
        b.dz3 = dz3.data_ptr(); b.dz2 = dz2.data_ptr(); b.dz1 = dz1.data_ptr()
        dxs = [None] * n_src
        nd = 0
        for i in range(n_src):
            if ctx.needs_input_grad[1 + i]:
                K = srcs[i].shape[1]
                dx = torch.empty(M, K, device=dev)
                d = b.dx[nd]
                d.W = w.w1.data_ptr() + 4 * cols[i]; d.K = K; d.dx = dx.data_ptr(); d.ld = K
                d.residual = 1 if i == residual else 0
                if pk_t is not None:
                    d.Wpk_t = pk_t.data_ptr() + (cols[i] // LAT) * _lib.PACK_BLOCK_BYTES
                dxs[i] = dx
                nd += 1
        b.n_dx = nd
        if pk_t is not None:
            nb1 = (w.w1.shape[1] + LAT - 1) // LAT
            b.W2pk_t = pk_t.data_ptr() + nb1 * _lib.PACK_BLOCK_BYTES
            b.W3pk_t = pk_t.data_ptr() + (nb1 + 1) * _lib.PACK_BLOCK_BYTES
        bufs, accs, grads_w = _grad_bufs(wt, ctx.targets)
        if has_ln:           # LayerNorm-affine gradients come out of the same pass
            b.d_gamma = bufs[6].data_ptr(); b.d_beta = bufs[7].data_ptr(); b.ln_accumulate = accs[6]
            if not _ln_defer(c, b, M, dev, bufs[6], bufs[7], accs[6] and accs[7]):
                b.ln_ws = _ln_workspace(c, M, dev).data_ptr()
        if M > 0:
            _lib.check(L.hgn_mlp_bwd(C.byref(b), _lib.stream_ptr()), 'hgn_mlp_bwd')
        elif has_ln and not accs[6]:
            bufs[6].zero_(); bufs[7].zero_()
        # ---- parameter gradients -------------------------------------------------------------------------------
        dw1, db1, dw2, db2, dw3, db3 = bufs[:6]
        tasks = [_wtask(0, z2.data_ptr(), LAT, LAT, None, dz3.data_ptr(), LAT, out_w, dw3.data_ptr(), LAT, db3.data_ptr(), accs[4]),
                 _wtask(0, z1.data_ptr(), LAT, LAT, None, dz2.data_ptr(), LAT, LAT, dw2.data_ptr(), LAT, db2.data_ptr(), accs[2])]
        first = True
        ldw1 = w.w1.shape[1]
        for i, s in enumerate(srcs):
            K = s.shape[1]
            for k0 in range(0, K, LAT):
                kw = min(LAT, K - k0)
                tasks.append(_wtask(0, s.data_ptr() + 4 * k0, _ld(s), kw, idxs[i].data_ptr() if idxs[i] is not None else None,
                                    dz1.data_ptr(), LAT, LAT, dw1.data_ptr() + 4 * (cols[i] + k0), ldw1,
                                    db1.data_ptr() if first else None, accs[0]))
                first = False
        if M == 0:
            _zero_unaccumulated(bufs[:6], accs[:6])
        elif ctx.gaps and not accs[0]:
            dw1.zero_()
        # (gathered / narrow sources = encoders: few launches, operands of E rows: not worth keeping alive)
        _run_wgrad(c, tasks, M, dev, keep=[z1, z2, dz1, dz2, dz3, *srcs], defer=all(i is None for i in idxs) and pk_t is not None)
        # ---- un-gather source gradients -----------------------------------------------------------------------
        for i in range(n_src):
            if dxs[i] is not None and idxs[i] is not None:
                full = torch.zeros_like(srcs[i])
                full.index_add_(0, idxs[i].long(), dxs[i])
                dxs[i] = full
            elif dxs[i] is not None and i == 0 and ctx.share and srcs[i].shape[1] == LAT and M > 0:
                share_grad(c, srcs[i], dxs[i])
            elif dxs[i] is not None:
                unshare_grad(c, srcs[i])
        return (None, *dxs, *grads_w)


def fused_mlp(srcs: Sequence[torch.Tensor], w: MLPWeights, idxs: Optional[Sequence[Optional[torch.Tensor]]] = None,
              residual: int = -1, post=None, cols: Optional[Sequence[int]] = None, share: bool = False):
    """`cols`: first W1 column of every source when the sources do NOT cover the input back to back -- column ranges left out stand for
    inputs that are zero for every row of this launch (an aggregate over edges none of which arrive at these rows): no operand, no
    product, a zero weight gradient.
    `post`: (packs_of(weights of the NEXT edge block), zero-fill wanted) -- the node-level pre-projection of that block (its P = [h W1s^T | h W1r^T]) and the zero fill of its
    aggregate buffer come out of the same launch: -> (out, (P, zeros) or None).
    `share`: the caller vouches that source 0 (a node latent) is otherwise consumed by EdgeBlockFn nodes only: its gradient tensor is
    offered to them as their accumulation target (share_grad)."""
    idxs = tuple(idxs) if idxs is not None else (None,) * len(srcs)
    wt = w.tensors()
    train = torch.is_grad_enabled() and any(t.requires_grad for t in list(srcs) + wt)
    cols = tuple(int(x) for x in cols) if cols is not None else None
    if post is None:
        return MLPFn.apply((len(srcs), idxs, residual, w.ln_w is not None, train, None, cols, share), *srcs, *wt)
    c = current()
    c.post_result = None
    out = MLPFn.apply((len(srcs), idxs, residual, w.ln_w is not None, train, post, cols, share), *srcs, *wt)
    got, c.post_result = c.post_result, None
    return out, got


# ------------------------------------------------------------------------------------------------------------
# edge block: split first layer  (h W_s^T)[snd] + (h W_r^T)[rcv] + e W_e^T
# ------------------------------------------------------------------------------------------------------------
class EdgeBlockFn(torch.autograd.Function):
    """e' = e + LN(MLP([h[snd] ; h[rcv] ; e])) with e, e' in receiver-sorted order, optionally followed IN THE SAME
    autograd node by the aggregation of e' over receivers (so that the backward adds d(e') and the scattered d(agg)
    inside the segment-reduce kernel instead of in a separate pass).

    Node rows.  ``parts is None``: ``h_s`` holds all ``topo.num_nodes`` rows (``h_r`` is None).  ``parts = (off_s, off_r)``: the set's
    senders all lie in rows [off_s, off_s + len(h_s)) and its receivers in [off_r, off_r + len(h_r)) of the node numbering -- the mesh rows
    and the hyper rows of a hierarchical graph live in tensors of their own and are never concatenated (hypergraphnet.py:21-54 reads
    ``graph.node_features`` as a list; only the gather indices are global).  ``h_r is None``: the same tensor serves both.  Then only
    those rows are projected, summed over and given a gradient, and the aggregate has one row per RECEIVER-part row."""

    @staticmethod
    def forward(ctx, topo, train, agg_ops, pre, parts, h_s, h_r, e, *wt):
        w = MLPWeights(*wt)
        w.check()
        if w.w1.shape[1] != 3 * LAT or w.ln_w is None:
            raise _lib.HgnError('edge model must be Linear(384,128)...+LayerNorm')
        h_s = _rowmajor(h_s)
        same = h_r is None
        h_r = h_s if same else _rowmajor(h_r)
        e = _rowmajor(e)
        _lib.require_gpu(e)
        dev = e.device
        L = _lib.lib()
        st = _lib.stream_ptr()
        c = current()
        E, N = topo.num_edges, topo.num_nodes
        off_s, off_r = parts if parts is not None else (0, 0)
        Ns, Nr = h_s.shape[0], h_r.shape[0]
        if e.shape[0] != E or (parts is None and Ns != N) or off_s + Ns > N or off_r + Nr > N or (same and off_s != off_r):
            raise _lib.HgnError(f'edge block: got {e.shape[0]} edge rows / node rows [{off_s}, {off_s + Ns}) and [{off_r}, {off_r + Nr}), '
                                f'topology has {E} / {N}')
        # P = [h W1s^T | h W1r^T] of the rows the set touches: one [Ns, 256] array, or one 128-wide array per part
        if same:
            P = torch.empty(Ns, 2 * LAT, device=dev)
            Ps, Pr, ldp = P.data_ptr(), P.data_ptr() + 4 * LAT, 2 * LAT
        else:
            P = (torch.empty(Ns, LAT, device=dev), torch.empty(Nr, LAT, device=dev))
            Ps, Pr, ldp = P[0].data_ptr(), P[1].data_ptr(), LAT
        pk = packs_of(w, ctx=c) if (all(_ld(h) % 4 == 0 and h.data_ptr() % 16 == 0 for h in (h_s, h_r)) and not c.fp32_only()) else None
        # the `sum` aggregate formed inside the edge kernel needs a zero-filled [N, 128] buffer: filled by the pre-projection launch,
        # which passes over the same node rows anyway (hgn_linear_fwd6z), instead of a launch of its own
        agg_zeroed = None
        if pre is not None and pk is not None and parts is None and tuple(pre[0].shape) == (N, 2 * LAT):
            P, agg_zeroed = pre                              # formed by the node kernel of the block before (fused_mlp: post)
            Ps, Pr = P.data_ptr(), P.data_ptr() + 4 * LAT
        else:
            if pk is not None and agg_ops == ('sum',) and 0 < E and topo.r.max_rows <= _FUSED_SEG_MAX_ROWS:
                agg_zeroed = torch.empty(Nr, LAT, device=dev)
            # (rows, first W1 column block, blocks, output): both halves in one launch when one tensor serves senders and receivers
            for h, n, b0, nb, optr, zero in (((h_s, Ns, 0, 2, Ps, agg_zeroed),) if same else
                                            ((h_s, Ns, 0, 1, Ps, None), (h_r, Nr, 1, 1, Pr, agg_zeroed))):
                if pk is not None:
                    pb = (C.c_void_p * 2)(*[pk.data_ptr() + (b0 + j) * _lib.PACK_BLOCK_BYTES for j in range(nb)], *([None] * (2 - nb)))
                    _lib.check(L.hgn_linear_fwd6z(h.data_ptr(), _ld(h), n, pb, nb, optr, ldp, zero.data_ptr() if zero is not None else None,
                                                  LAT, c.products(), st), 'hgn_linear_fwd6z')
                else:
                    wb = (C.c_void_p * 2)(*[w.w1.data_ptr() + 4 * LAT * (b0 + j) for j in range(nb)], *([None] * (2 - nb)))
                    _lib.check(L.hgn_linear_fwd(h.data_ptr(), _ld(h), n, wb, nb, 3 * LAT, optr, ldp, st), 'hgn_linear_fwd')
        out = torch.empty(E, LAT, device=dev)
        a = _lib.MlpFwd()
        c.stamp(a)
        a.M = E
        a.n_src = 1
        s = a.src[0]
        s.x = e.data_ptr(); s.ld = _ld(e); s.K = LAT; s.idx = None; s.W = w.w1.data_ptr() + 4 * 2 * LAT
        if pk is not None:
            s.Wpk = pk.data_ptr() + 2 * _lib.PACK_BLOCK_BYTES
            a.W2pk = pk.data_ptr() + 3 * _lib.PACK_BLOCK_BYTES
            a.W3pk = pk.data_ptr() + 4 * _lib.PACK_BLOCK_BYTES
        a.n_add = 2
        # (the indices are global row numbers: the bases are those of row 0, which only exists when the part starts there)
        a.add[0].P = Ps - 4 * ldp * off_s; a.add[0].ld = ldp; a.add[0].idx = topo.snd.data_ptr()
        a.add[1].P = Pr - 4 * ldp * off_r; a.add[1].ld = ldp; a.add[1].idx = topo.rcv.data_ptr()
        saves = _alloc_saves(E, True, dev) if train else None
        _fill_common_fwd(a, w, out, e, saves)
        agg, amax, amin = None, None, None
        # `sum` aggregation inside the edge kernel itself (no second pass over e') when the segments are short enough for the
        # in-kernel sums to be order independent (include/hgn_mp.h: seg_out)
        fuse_agg = (agg_ops == ('sum',) and pk is not None and 0 < E and topo.r.max_rows <= _FUSED_SEG_MAX_ROWS
                    and L.hgn_mlp_fwd6_eligible(C.byref(a)))
        if fuse_agg:
            agg = agg_zeroed if agg_zeroed is not None else torch.zeros(Nr, LAT, device=dev)
            a.seg_out = agg.data_ptr() - 4 * LAT * off_r; a.ld_seg_out = LAT; a.seg_ids = topo.rcv.data_ptr()
        if E > 0:
            _lib.check(L.hgn_mlp_fwd(C.byref(a), st), 'hgn_mlp_fwd')
        if agg_ops is not None and not fuse_agg:
            arr, codes = _ops_array(agg_ops)
            k = len(codes)
            agg = torch.empty(Nr, k * LAT, device=dev)
            if train and 2 in codes:
                amax = torch.empty(Nr, LAT, dtype=torch.int32, device=dev)
            if train and 3 in codes:
                amin = torch.empty(Nr, LAT, dtype=torch.int32, device=dev)
            L.hgn_prof_tag(2)
            _lib.check(L.hgn_segment_reduce_fwd(out.data_ptr(), LAT, LAT, None, topo.r.rowptr.data_ptr() + 4 * off_r, Nr, arr, k,
                                                agg.data_ptr(), k * LAT, amax.data_ptr() if amax is not None else None,
                                                amin.data_ptr() if amin is not None else None, st), 'hgn_segment_reduce_fwd')
            L.hgn_prof_tag(0)
        if train and _GATE_LOG is not None:
            _GATE_LOG.append((wt[0].data_ptr(), saves[4], topo.r.perm))
        if train and _ARG_LOG is not None and (amax is not None or amin is not None):
            _ARG_LOG.append((wt[0].data_ptr(), amax, amin, topo.r.perm, off_r, N))
        if train:
            ctx.set_materialize_grads(False)
            ctx.topo = topo
            ctx.saves = saves
            ctx.agg = (agg_ops, amax, amin)
            ctx.targets = _grad_targets(wt)
            ctx.pk_t = packs_of(w, transposed=True, ctx=c) if pk is not None else None
            ctx.hgn = c
            ctx.rows = (same, off_s, off_r)
            ctx.save_for_backward(h_s, None if same else h_r, e, *wt)
        return (out, agg) if agg_ops is not None else out

    @staticmethod
    def backward(ctx, d_out, d_agg=None):
        topo = ctx.topo
        h_s, h_r, e, *wt = ctx.saved_tensors
        same, off_s, off_r = ctx.rows
        if same:
            h_r = h_s
        Ns, Nr = h_s.shape[0], h_r.shape[0]
        w = MLPWeights(*wt)
        z1, z2, xhat, rstd, bits = ctx.saves
        agg_ops, amax, amin = ctx.agg
        L = _lib.lib()
        E, N = topo.num_edges, topo.num_nodes
        st = _lib.stream_ptr()
        c = ctx.hgn
        if d_out is None and d_agg is None:
            return (None,) * (8 + len(wt))
        dev = (d_out if d_out is not None else d_agg).device
        dz3 = dz2 = None              # [E,128] each, only on the two-launch path (the fused kernel keeps them on chip)
        dz1 = torch.empty((E + 63) // 64 * 64, LAT, device=dev)[:E]      # whole 64-row tiles: hgn_edge_bwd_fused stores the padding rows too
        de = torch.empty(E, LAT, device=dev)
        b = _lib.MlpBwd()
        c.stamp(b)
        b.M = E
        b.out_w = LAT
        if d_out is not None:
            d_out = _rowmajor(d_out)
            b.d_out = d_out.data_ptr(); b.ld_dout = _ld(d_out)
        if d_agg is not None:
            # d(e') + the aggregation backward are summed inside the kernel's load of d_out (no dE tensor, no extra pass)
            d_agg = _rowmajor(d_agg)
            arr, codes = _ops_array(agg_ops)
            # (per-receiver arrays have one row per row of the receiver part; the kernels index them by global row number)
            b.agg_dout = d_agg.data_ptr() - 4 * _ld(d_agg) * off_r; b.ld_agg = _ld(d_agg); b.n_agg_ops = len(codes)
            for i, cde in enumerate(codes):
                b.agg_ops[i] = cde
            b.agg_seg = topo.rcv.data_ptr(); b.agg_rowptr = topo.r.rowptr.data_ptr()
            b.agg_argmax = amax.data_ptr() - 4 * LAT * off_r if amax is not None else None
            b.agg_argmin = amin.data_ptr() - 4 * LAT * off_r if amin is not None else None
        b.ln_g = w.ln_w.data_ptr(); b.xhat = xhat.data_ptr(); b.rstd = rstd.data_ptr()
        b.z2 = z2.data_ptr(); b.z1 = z1.data_ptr(); b.relu_bits = bits.data_ptr()
        b.W3 = w.w3.data_ptr(); b.W2 = w.w2.data_ptr(); b.ldw1 = 3 * LAT
        b.dz1 = dz1.data_ptr()
        b.n_dx = 1
        d = b.dx[0]
        d.W = w.w1.data_ptr() + 4 * 2 * LAT; d.K = LAT; d.dx = de.data_ptr(); d.ld = LAT; d.residual = 1
        pk_t = ctx.pk_t
        if pk_t is not None:
            d.Wpk_t = pk_t.data_ptr() + 2 * _lib.PACK_BLOCK_BYTES
            b.W2pk_t = pk_t.data_ptr() + 3 * _lib.PACK_BLOCK_BYTES
            b.W3pk_t = pk_t.data_ptr() + 4 * _lib.PACK_BLOCK_BYTES
        bufs, accs, grads_w = _grad_bufs(wt, ctx.targets)
        dw1, db1, dw2, db2, dw3, db3, dg, dbt = bufs
        b.d_gamma = dg.data_ptr(); b.d_beta = dbt.data_ptr(); b.ln_accumulate = accs[6]
        if not _ln_defer(c, b, E, dev, dg, dbt, accs[6] and accs[7]):
            b.ln_ws = _ln_workspace(c, E, dev).data_ptr()
        # dP = [sum over edges sent by n of dz1 | sum over edges received by n of dz1]; the receiver half comes out of the
        # backward kernel itself when the segments are short (include/hgn_mp.h: seg_dz1)
        if same:
            dP = torch.empty(Ns, 2 * LAT, device=dev)
            dPs, dPr, ldd = dP.data_ptr(), dP.data_ptr() + 4 * LAT, 2 * LAT
        else:
            dP = (torch.empty(Ns, LAT, device=dev), torch.empty(Nr, LAT, device=dev))
            dPs, dPr, ldd = dP[0].data_ptr(), dP[1].data_ptr(), LAT
        # One pass for data gradients AND weight gradients (include/hgn_mp.h: hgn_edge_bwd_fused): dz3 / dz2 never reach memory.
        may_fuse = c.fused() and pk_t is not None and E > 0 and c.wgrad_stream is None and accs[2] == accs[4]
        fused = may_fuse and bool(L.hgn_edge_bwd_fused_eligible(C.byref(b)))      # (dW3 / dW2 share one accumulate flag in hgn_wfuse_t)
        if may_fuse and not fused and d_agg is not None and (d_out is None or _ld(d_out) == LAT):
            # Several aggregates (pna) or arg-routed ones: the fused kernel gathers ONE `sum` row per edge.  The gradient that reaches e'
            # -- d(e') + the aggregation backward scattered back to the edge rows -- is formed once by the streaming kernel
            # (hgn_segment_reduce_bwd with `base`: the same values added in the same order as hgn_mlp_bwd's in-register d_out_eff)
            # and handed over as d_out: 1 KB per edge row more traffic, against the two-launch backward's 2.4 KB.
            b2 = _lib.MlpBwd.from_buffer_copy(b)
            b2.agg_dout = None; b2.n_agg_ops = 0; b2.agg_seg = None; b2.agg_rowptr = None; b2.agg_argmax = None; b2.agg_argmin = None
            b2.d_out = dz1.data_ptr(); b2.ld_dout = LAT                # (any aligned address: eligibility looks at shapes, not at the data)
            if L.hgn_edge_bwd_fused_eligible(C.byref(b2)):
                g_eff = torch.empty(E, LAT, device=dev)
                arr, codes = _ops_array(agg_ops)
                if E <= 16 * Nr:
                    # rows are in receiver order: one half-wave per receiver loads its gradient / arg rows once (hgn_segment_reduce_bwd_sorted)
                    _lib.check(L.hgn_segment_reduce_bwd_sorted(d_agg.data_ptr(), _ld(d_agg), topo.r.rowptr.data_ptr() + 4 * off_r, Nr, arr, len(codes),
                                                               amax.data_ptr() if amax is not None else None,
                                                               amin.data_ptr() if amin is not None else None,
                                                               d_out.data_ptr() if d_out is not None else None, g_eff.data_ptr(), LAT, st),
                               'hgn_segment_reduce_bwd_sorted')
                else:   # few long segments (the rows that arrive at a hyper node): a half-wave per ROW keeps the chip busy, one per segment would not
                    _lib.check(L.hgn_segment_reduce_bwd(d_agg.data_ptr() - 4 * _ld(d_agg) * off_r, _ld(d_agg), LAT, None, topo.rcv.data_ptr(),
                                                        topo.r.rowptr.data_ptr(), E, arr, len(codes),
                                                        amax.data_ptr() - 4 * LAT * off_r if amax is not None else None,
                                                        amin.data_ptr() - 4 * LAT * off_r if amin is not None else None,
                                                        d_out.data_ptr() if d_out is not None else None, g_eff.data_ptr(), LAT, st),
                               'hgn_segment_reduce_bwd')
                b2.d_out = g_eff.data_ptr()
                b, fused = b2, True
        fuse_seg = (not fused and pk_t is not None and E > 0 and topo.r.max_rows <= _FUSED_SEG_MAX_ROWS
                    and L.hgn_mlp_bwd6_eligible(C.byref(b)))
        if fuse_seg:
            (dP[:, LAT:] if same else dP[1]).zero_()
            b.seg_dz1 = dPr - 4 * ldd * off_r; b.ld_seg_dz1 = ldd; b.seg_ids = topo.rcv.data_ptr()
        if fused:
            wf = _lib.WFuse()
            wf.z2 = z2.data_ptr(); wf.z1 = z1.data_ptr()
            wf.dW3 = dw3.data_ptr(); wf.db3 = db3.data_ptr(); wf.dW2 = dw2.data_ptr(); wf.db2 = db2.data_ptr()
            wf.accumulate = accs[2]
            nb = C.c_size_t(0)
            _lib.check(L.hgn_edge_bwd_fused_workspace_bytes(E, C.byref(nb)), 'hgn_edge_bwd_fused_workspace_bytes')
            if _wred_deferrable(c, accs[2] and accs[4]):
                ws = torch.empty((nb.value + 3) // 4, dtype=torch.float32, device=dev)
                red = (_lib.WRed * 2)()
                _lib.check(L.hgn_edge_bwd_fused_partial(C.byref(b), C.byref(wf), ws.data_ptr(), 4 * ws.numel(), red, st), 'hgn_edge_bwd_fused_partial')
                c.redq.extend((_lib.WRed.from_buffer_copy(r), ws) for r in red)
            else:
                ws = c.workspace(dev, nb.value, 'fused')
                _lib.check(L.hgn_edge_bwd_fused(C.byref(b), C.byref(wf), ws.data_ptr(), ws.numel(), st), 'hgn_edge_bwd_fused')
            # dW1's edge block: dz1 is in memory anyway (the sender / receiver sums read it), one streaming task
            _run_wgrad(c, [_wtask(0, e.data_ptr(), _ld(e), LAT, None, dz1.data_ptr(), LAT, LAT, dw1.data_ptr() + 4 * 2 * LAT, 3 * LAT,
                               db1.data_ptr(), accs[0])], E, dev, edge_level=True, keep=[e, dz1])
        else:
            dz3 = torch.empty(E, LAT, device=dev)
            dz2 = torch.empty(E, LAT, device=dev)
            b.dz3 = dz3.data_ptr(); b.dz2 = dz2.data_ptr()
            if E > 0:
                _lib.check(L.hgn_mlp_bwd(C.byref(b), st), 'hgn_mlp_bwd')
            elif not accs[6]:
                dg.zero_(); dbt.zero_()
            tasks = [_wtask(0, z2.data_ptr(), LAT, LAT, None, dz3.data_ptr(), LAT, LAT, dw3.data_ptr(), LAT, db3.data_ptr(), accs[4]),
                     _wtask(0, z1.data_ptr(), LAT, LAT, None, dz2.data_ptr(), LAT, LAT, dw2.data_ptr(), LAT, db2.data_ptr(), accs[2]),
                     _wtask(0, e.data_ptr(), _ld(e), LAT, None, dz1.data_ptr(), LAT, LAT, dw1.data_ptr() + 4 * 2 * LAT, 3 * LAT,
                            db1.data_ptr(), accs[0])]
            if E == 0:      # dW1's node-row column blocks are written by the node-level launch below (from dP = 0)
                _zero_unaccumulated(bufs[:6], accs[:6])
            _run_wgrad(c, tasks, E, dev, edge_level=True, keep=[z1, z2, e, dz1, dz2, dz3])
        ops = (C.c_int32 * 1)(0)
        if _SEG_PAIR and not fuse_seg and same and dz1.data_ptr() % 16 == 0:
            # both sums in one pass over dz1: the rows a node sends are rows its neighbours receive (include/hgn_mp.h: hgn_segment_sum_pair)
            _lib.check(L.hgn_segment_sum_pair(dz1.data_ptr(), LAT, topo.r.rowptr.data_ptr() + 4 * off_r, topo.s.perm.data_ptr(),
                                              topo.s.rowptr.data_ptr() + 4 * off_s, Ns, dPr, ldd, dPs, ldd, st), 'segment_sum_pair')
        else:
            _lib.check(L.hgn_segment_reduce_fwd(dz1.data_ptr(), LAT, LAT, topo.s.perm.data_ptr(), topo.s.rowptr.data_ptr() + 4 * off_s, Ns,
                                                ops, 1, dPs, ldd, None, None, st), 'segment_reduce(senders)')
            if not fuse_seg:
                _lib.check(L.hgn_segment_reduce_fwd(dz1.data_ptr(), LAT, LAT, None, topo.r.rowptr.data_ptr() + 4 * off_r, Nr, ops, 1,
                                                    dPr, ldd, None, None, st), 'segment_reduce(receivers)')
        task_s = _wtask(0, h_s.data_ptr(), _ld(h_s), LAT, None, dPs, ldd, LAT, dw1.data_ptr(), 3 * LAT, None, accs[0])
        task_r = _wtask(0, h_r.data_ptr(), _ld(h_r), LAT, None, dPr, ldd, LAT, dw1.data_ptr() + 4 * LAT, 3 * LAT, None, accs[0])
        if same:
            _run_wgrad(c, [task_s, task_r], Ns, dev, keep=[h_s, dP], defer=True)
        else:
            _run_wgrad(c, [task_s], Ns, dev, keep=[h_s, dP[0]], defer=True)
            _run_wgrad(c, [task_r], Nr, dev, keep=[h_r, dP[1]], defer=True)
        # inputs: topo, train, agg_ops, pre, parts, h_s, h_r, e, *weights
        dhs = [None, None]
        for slot, (need, n, b0, nb, gptr) in enumerate(((ctx.needs_input_grad[5], Ns, 0, 2 if same else 1, dPs),
                                                         (not same and ctx.needs_input_grad[6], Nr, 1, 1, dPr))):
            if not need:
                continue
            hsrc = h_s if slot == 0 else h_r
            dh = shared_grad(c, hsrc) if (pk_t is not None and n > 0) else None
            if dh is not None:                       # the node update's gradient for this very tensor: add into it, return nothing
                pb = (C.c_void_p * 2)(*[pk_t.data_ptr() + (b0 + j) * _lib.PACK_BLOCK_BYTES for j in range(nb)], *([None] * (2 - nb)))
                _lib.check(L.hgn_linear_bwd6a(gptr, ldd, n, pb, nb, dh.data_ptr(), LAT, 1, c.products(), st), 'hgn_linear_bwd6a')
                share_stats['accumulated'] += 1
                continue
            unshare_grad(c, hsrc)                    # a gradient tensor of our own for this h: the engine will add
            share_stats['own'] += 1
            dh = torch.empty(n, LAT, device=dev)
            if pk_t is not None:
                pb = (C.c_void_p * 2)(*[pk_t.data_ptr() + (b0 + j) * _lib.PACK_BLOCK_BYTES for j in range(nb)], *([None] * (2 - nb)))
                _lib.check(L.hgn_linear_bwd6(gptr, ldd, n, pb, nb, dh.data_ptr(), LAT, c.products(), st), 'hgn_linear_bwd6')
            else:
                wb = (C.c_void_p * 2)(*[w.w1.data_ptr() + 4 * LAT * (b0 + j) for j in range(nb)], *([None] * (2 - nb)))
                _lib.check(L.hgn_linear_bwd(gptr, ldd, n, wb, nb, 3 * LAT, dh.data_ptr(), LAT, st), 'hgn_linear_bwd')
            dhs[slot] = dh
        return (None, None, None, None, None, dhs[0], dhs[1], de, *grads_w)


def edge_block(h_all: torch.Tensor, e_sorted: torch.Tensor, topo, w: MLPWeights, agg_ops=None, pre=None, parts=None, h_r=None):
    """-> e'   or, with ``agg_ops`` (e.g. ('sum',) or the four PNA ops),  (e', agg[N, len(ops)*128]).
    ``parts = (off_s, off_r)`` [+ ``h_r``]: node rows by part (EdgeBlockFn); the aggregate then has the receiver part's rows only."""
    wt = w.tensors()
    hs = [h_all, e_sorted] + ([h_r] if h_r is not None else [])
    train = torch.is_grad_enabled() and any(t.requires_grad for t in hs + wt)
    return EdgeBlockFn.apply(topo, train, tuple(agg_ops) if agg_ops is not None else None, pre, parts, h_all, h_r,
                             e_sorted, *wt)


# ------------------------------------------------------------------------------------------------------------
# aggregation
# ------------------------------------------------------------------------------------------------------------
def _ops_array(ops: Sequence[str]):
    codes = []
    for o in ops:
        if o not in OP_CODES:
            raise Exception('Invalid operation type!')           # src/util.py:132
        codes.append(OP_CODES[o])
    return (C.c_int32 * len(codes))(*codes), codes


class AggregateFn(torch.autograd.Function):
    """[N, n_sets * n_ops * D]: per edge set (sorted layout, perm=None, or user layout with its CSR perm), per op."""

    @staticmethod
    def forward(ctx, csrs, ops, train, *datas):
        L = _lib.lib()
        arr, codes = _ops_array(ops)
        datas = [_rowmajor(d) for d in datas]
        D = datas[0].shape[1]
        dev = datas[0].device
        _lib.require_gpu(datas[0])
        N = csrs[0][1].shape[0] - 1
        k = len(codes)
        width = len(datas) * k * D
        out = torch.empty(N, width, device=dev)
        need_arg = train and any(c >= 2 for c in codes)
        args = []
        for i, (d, (perm, rowptr, seg)) in enumerate(zip(datas, csrs)):
            amax = torch.empty(N, D, dtype=torch.int32, device=dev) if need_arg and 2 in codes else None
            amin = torch.empty(N, D, dtype=torch.int32, device=dev) if need_arg and 3 in codes else None
            _lib.check(L.hgn_segment_reduce_fwd(d.data_ptr(), _ld(d), D, perm.data_ptr() if perm is not None else None,
                                                rowptr.data_ptr(), N, arr, k, out.data_ptr() + 4 * i * k * D, width,
                                                amax.data_ptr() if amax is not None else None,
                                                amin.data_ptr() if amin is not None else None, _lib.stream_ptr()),
                       'hgn_segment_reduce_fwd')
            args.append((amax, amin))
        if train:
            ctx.cfg = (csrs, ops, [tuple(d.shape) for d in datas], args, width)
        return out

    @staticmethod
    def backward(ctx, d_out):
        csrs, ops, shapes, args, width = ctx.cfg
        L = _lib.lib()
        arr, codes = _ops_array(ops)
        k = len(codes)
        d_out = _rowmajor(d_out)
        dev = d_out.device
        grads = []
        for i, ((perm, rowptr, seg), shp, (amax, amin)) in enumerate(zip(csrs, shapes, args)):
            if not ctx.needs_input_grad[3 + i]:
                grads.append(None)
                continue
            E, D = shp
            g = torch.empty(E, D, device=dev)
            _lib.check(L.hgn_segment_reduce_bwd(d_out.data_ptr() + 4 * i * k * D, _ld(d_out), D,
                                                perm.data_ptr() if perm is not None else None, seg.data_ptr(),
                                                rowptr.data_ptr(), E, arr, k,
                                                amax.data_ptr() if amax is not None else None,
                                                amin.data_ptr() if amin is not None else None, None, g.data_ptr(), D,
                                                _lib.stream_ptr()), 'hgn_segment_reduce_bwd')
            grads.append(g)
        return (None, None, None, *grads)


class SegmentStdFn(torch.autograd.Function):
    """'std' of util.unsorted_segment_operation (src/util.py:129-130 -> torch_scatter.scatter_std, unbiased): include/hgn_mp.h
    hgn_segment_std_fwd / _bwd.  csr = (perm or None, rowptr, seg)."""

    @staticmethod
    def forward(ctx, csr, data):
        perm, rowptr, seg = csr
        data = _rowmajor(data)
        _lib.require_gpu(data)
        N, D = rowptr.shape[0] - 1, data.shape[1]
        out = torch.empty(N, D, device=data.device)
        mean = torch.empty(N, D, device=data.device)
        _lib.check(_lib.lib().hgn_segment_std_fwd(data.data_ptr(), _ld(data), D, perm.data_ptr() if perm is not None else None,
                                                  rowptr.data_ptr(), N, out.data_ptr(), D, mean.data_ptr(), _lib.stream_ptr()),
                   'hgn_segment_std_fwd')
        ctx.csr = csr
        ctx.save_for_backward(data, out, mean)
        return out

    @staticmethod
    def backward(ctx, d_out):
        perm, rowptr, seg = ctx.csr
        data, out, mean = ctx.saved_tensors
        d_out = _rowmajor(d_out).contiguous()
        E, D = data.shape
        g = torch.empty(E, D, device=data.device)
        _lib.check(_lib.lib().hgn_segment_std_bwd(d_out.data_ptr(), out.data_ptr(), mean.data_ptr(), D, data.data_ptr(), _ld(data), D,
                                                  perm.data_ptr() if perm is not None else None, seg.data_ptr(), rowptr.data_ptr(),
                                                  E, g.data_ptr(), D, _lib.stream_ptr()), 'hgn_segment_std_bwd')
        return None, g


def segment_std(data: torch.Tensor, csr) -> torch.Tensor:
    return SegmentStdFn.apply(tuple(csr), data)


def aggregate(datas: Sequence[torch.Tensor], csrs: Sequence[Tuple], ops: Sequence[str]) -> torch.Tensor:
    """csrs[i] = (perm or None, rowptr, seg) for data i."""
    train = torch.is_grad_enabled() and any(d.requires_grad for d in datas)
    return AggregateFn.apply(tuple(csrs), tuple(ops), train, *datas)


# ------------------------------------------------------------------------------------------------------------
# optimiser / profiler helpers
# ------------------------------------------------------------------------------------------------------------
def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    for t in (p, g, m, v):
        _lib.require_gpu(t)
        if not t.is_contiguous() or t.dtype != torch.float32:
            raise _lib.HgnError('adam_step needs contiguous fp32 flat buffers')
    _lib.check(_lib.lib().hgn_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2,
                                        eps, step, grad_scale, _lib.stream_ptr()), 'hgn_adam_step')


def adam_step_dev(p, g, m, v, lr, beta1, beta2, eps, step_dev, grad_scale=1.0):
    """Adam with the step counter on the device (int32 tensor, incremented by the call): capturable into a HIP graph."""
    for t in (p, g, m, v, step_dev):
        _lib.require_gpu(t)
    if step_dev.dtype != torch.int32 or step_dev.numel() != 1:
        raise _lib.HgnError('step_dev must be a one-element int32 tensor')
    _lib.check(_lib.lib().hgn_adam_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2,
                                            eps, step_dev.data_ptr(), grad_scale, _lib.stream_ptr()), 'hgn_adam_step_dev')


def prof_enable(on: bool):
    _lib.check(_lib.lib().hgn_prof_enable(1 if on else 0))


def prof_reset():
    _lib.check(_lib.lib().hgn_prof_reset())


def prof_collect():
    n = _lib.NUM_KERNEL_IDS
    ms = (C.c_double * n)(); cnt = (C.c_int64 * n)(); units = (C.c_double * n)()
    _lib.check(_lib.lib().hgn_prof_collect(ms, cnt, units), 'hgn_prof_collect')
    return {_lib.KERNEL_NAMES[i]: {'ms': ms[i], 'count': cnt[i], 'units': units[i]} for i in range(n) if cnt[i]}
