"""Host wrappers of include/hgn_features.h: the frame -> graph-feature step in front of the message-passing path
(SURVEY.md section 8 rows f2 / f3).  Device tensors in, device tensors out; nothing here falls back to torch math.
"""
import ctypes as C

import torch

from . import _lib
from .ops import _workspace


def _f32_rows(t: torch.Tensor) -> torch.Tensor:
    """fp32 2-D view whose rows are unit-stride (a column slice of a wider row-major tensor is fine)."""
    if t.dim() == 1:
        t = t.unsqueeze(1)
    if t.dtype != torch.float32:
        t = t.float()
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(int(t.shape[1]), 1)


def _ids(t: torch.Tensor, dev) -> torch.Tensor:
    return t.to(device=dev, dtype=torch.int64).contiguous()


def cells_to_edges(cells: torch.Tensor, deform: bool = False):
    """src/util.py:50-89: -> (senders_two_way, receivers_two_way, n_undirected); int64 device tensors."""
    _lib.require_gpu(cells)
    verts = 4 if deform else 3
    if cells.dim() != 2 or cells.shape[1] < verts:
        raise ValueError(f'cells must be [n_cells, {verts}]')
    cells = cells[:, :verts].to(torch.int64).contiguous()
    F = cells.shape[0]
    dev = cells.device
    L = _lib.lib()
    nb = C.c_size_t(0)
    _lib.check(L.hgn_cells_to_edges_workspace_bytes(F, verts, C.byref(nb)), 'hgn_cells_to_edges_workspace_bytes')
    ws = _workspace(dev, nb.value, 'cells')
    s = torch.empty(2 * verts * F, dtype=torch.int64, device=dev)
    r = torch.empty(2 * verts * F, dtype=torch.int64, device=dev)
    n = C.c_int64(0)
    _lib.check(L.hgn_cells_to_edges(cells.data_ptr(), F, verts, s.data_ptr(), r.data_ptr(), C.byref(n), ws.data_ptr(),
                                    ws.numel(), _lib.stream_ptr()), 'hgn_cells_to_edges')
    return s[:2 * n.value], r[:2 * n.value], n.value


def rel_edge_features(a: torch.Tensor, b, senders: torch.Tensor, receivers: torch.Tensor, want_feat: bool = True,
                      want_len: bool = False):
    """[a[s]-a[r], |.|, b[s]-b[r], |.|] per edge (b may be None) and/or the length |a[s]-a[r]|."""
    _lib.require_gpu(a)
    dev = a.device
    a = _f32_rows(a)
    da = a.shape[1]
    db = 0
    if b is not None:
        b = _f32_rows(b.to(dev))
        db = b.shape[1]
        if b.shape[0] != a.shape[0]:
            raise ValueError('a and b must have the same number of rows')
    s, r = _ids(senders, dev), _ids(receivers, dev)
    E = s.shape[0]
    if r.shape[0] != E:
        raise ValueError('senders / receivers length mismatch')
    W = da + 1 + (db + 1 if db else 0)
    feat = torch.empty(E, W, dtype=torch.float32, device=dev) if want_feat else None
    ln = torch.empty(E, dtype=torch.float32, device=dev) if want_len else None
    _lib.check(_lib.lib().hgn_rel_edge_features(
        a.data_ptr(), _ld(a), da, b.data_ptr() if db else None, _ld(b) if db else 0, db, a.shape[0], s.data_ptr(),
        r.data_ptr(), E, feat.data_ptr() if want_feat else None, W, ln.data_ptr() if want_len else None,
        _lib.stream_ptr()), 'hgn_rel_edge_features')
    return feat, ln


_map_cache = {}


def _class_map(mapping, dev):
    if mapping is None:
        return None
    key = (tuple(mapping), dev.type, dev.index)
    t = _map_cache.get(key)
    if t is None:
        t = torch.tensor(list(mapping), dtype=torch.int32, device=dev)
        _map_cache[key] = t
    return t


def node_features(cur, prev, node_type: torch.Tensor, mapping, n_classes: int, vel_first: bool = True,
                  vel_mask_type: int = -1, d: int = None) -> torch.Tensor:
    """[velocity | one-hot(class)] (or [one-hot | velocity]); ``node_type`` is the reference's [N, 1] (or [N]) tensor,
    ``mapping`` an optional tuple raw type -> class."""
    _lib.require_gpu(node_type)
    dev = node_type.device
    nt = node_type.to(torch.int64)
    if nt.dim() == 2:
        nt = nt[:, 0]
    N = nt.shape[0]
    ldt = nt.stride(0) if N > 1 else 1
    if cur is not None:
        cur = _f32_rows(cur.to(dev))
        d = cur.shape[1]
        ld = _ld(cur)
        if prev is not None:
            prev = _f32_rows(prev.to(dev))
            if prev.shape != cur.shape:
                raise ValueError('cur / prev shape mismatch')
            if _ld(prev) != ld:
                cur, prev = cur.contiguous(), prev.contiguous()
                ld = _ld(cur)
    else:
        ld = d = int(d or 0)
    m = _class_map(mapping, dev)
    out = torch.empty(N, d + n_classes, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().hgn_node_features(
        cur.data_ptr() if cur is not None else None, prev.data_ptr() if prev is not None else None, ld, d,
        nt.data_ptr(), ldt, m.data_ptr() if m is not None else None, len(mapping) if mapping is not None else 0,
        n_classes, 1 if vel_first else 0, vel_mask_type, N, out.data_ptr(), d + n_classes, _lib.stream_ptr()),
        'hgn_node_features')
    return out


def col_stats(x: torch.Tensor) -> torch.Tensor:
    """-> device [2F]: column sums then column sums of squares (fp64 accumulation, deterministic)."""
    _lib.require_gpu(x)
    x = x.reshape(x.shape[0], -1) if x.dim() != 2 else x
    x = x.float().contiguous()
    rows, F = x.shape
    L = _lib.lib()
    nb = C.c_size_t(0)
    _lib.check(L.hgn_col_stats_workspace_bytes(rows, F, C.byref(nb)), 'hgn_col_stats_workspace_bytes')
    ws = _workspace(x.device, nb.value, 'stats')
    batch = torch.empty(2 * F, dtype=torch.float32, device=x.device)
    _lib.check(L.hgn_col_stats(x.data_ptr(), rows, F, batch.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
               'hgn_col_stats')
    return batch


def normalizer_update(acc_sum, acc_sumsq, acc_count, num_acc, batch, count, max_acc: float) -> None:
    _lib.require_gpu(acc_sum)
    F = acc_sum.numel()
    _lib.check(_lib.lib().hgn_normalizer_update(acc_sum.data_ptr(), acc_sumsq.data_ptr(), acc_count.data_ptr(),
                                                num_acc.data_ptr(), batch.data_ptr(), count.data_ptr(), F,
                                                float(max_acc), _lib.stream_ptr()), 'hgn_normalizer_update')


def normalize(x: torch.Tensor, acc_sum, acc_sumsq, acc_count, eps: float, inverse: bool = False) -> torch.Tensor:
    _lib.require_gpu(x)
    shape = x.shape
    F = acc_sum.numel()
    if x.dim() == 0 or shape[-1] != F and not (F == 1):
        raise ValueError(f'last dimension {tuple(shape)} does not match the normaliser width {F}')
    xf = x.float().contiguous()
    rows = xf.numel() // F
    out = torch.empty_like(xf)
    _lib.check(_lib.lib().hgn_normalize(xf.data_ptr(), rows, F, acc_sum.data_ptr(), acc_sumsq.data_ptr(),
                                        acc_count.data_ptr(), float(eps), 1 if inverse else 0, out.data_ptr(),
                                        _lib.stream_ptr()), 'hgn_normalize')
    return out


def lincomb3(a: torch.Tensor, ca: float, b: torch.Tensor, cb: float, c=None, cc: float = 0.0) -> torch.Tensor:
    """(ca*a + cb*b) + cc*c, every product / sum rounded on its own (the reference's left-to-right fp32 order)."""
    _lib.require_gpu(a)
    dev = a.device
    a = a.float().contiguous()
    b = b.to(dev).float().contiguous()
    if b.shape != a.shape:
        raise ValueError('lincomb3: shape mismatch')
    if c is not None:
        c = c.to(dev).float().contiguous()
        if c.shape != a.shape:
            raise ValueError('lincomb3: shape mismatch')
    out = torch.empty_like(a)
    _lib.check(_lib.lib().hgn_lincomb3(a.data_ptr(), float(ca), b.data_ptr(), float(cb),
                                       c.data_ptr() if c is not None else None, float(cc), a.numel(), out.data_ptr(),
                                       _lib.stream_ptr()), 'hgn_lincomb3')
    return out


def radius_edges(pos: torch.Tensor, node_type: torch.Tensor, radius: float, sender_type: int, receiver_type: int,
                 nbr_rowptr=None, nbr=None):
    """plate.py:84-110: directed pairs closer than ``radius`` with the given endpoint types that are not mesh
    neighbours (CSR ``nbr_rowptr`` / ``nbr``); ascending (sender, receiver) order.  -> (senders, receivers) int64."""
    _lib.require_gpu(pos)
    dev = pos.device
    pos = _f32_rows(pos)
    nt = node_type.to(device=dev, dtype=torch.int64)
    if nt.dim() == 2:
        nt = nt[:, 0]
    N = pos.shape[0]
    ldt = nt.stride(0) if N > 1 else 1
    L = _lib.lib()
    nb = C.c_size_t(0)
    _lib.check(L.hgn_radius_edges_workspace_bytes(N, C.byref(nb)), 'hgn_radius_edges_workspace_bytes')
    ws = _workspace(dev, nb.value, 'radius')
    offsets = torch.empty(N + 1, dtype=torch.int32, device=dev)
    total = C.c_int64(0)
    args = (pos.data_ptr(), _ld(pos), pos.shape[1], nt.data_ptr(), ldt, N, float(radius), int(sender_type),
            int(receiver_type), nbr_rowptr.data_ptr() if nbr_rowptr is not None else None,
            nbr.data_ptr() if nbr is not None else None)
    _lib.check(L.hgn_radius_edges_count(*args, offsets.data_ptr(), C.byref(total), ws.data_ptr(), ws.numel(),
                                        _lib.stream_ptr()), 'hgn_radius_edges_count')
    s = torch.empty(total.value, dtype=torch.int64, device=dev)
    r = torch.empty(total.value, dtype=torch.int64, device=dev)
    if total.value:
        _lib.check(L.hgn_radius_edges_fill(*args, offsets.data_ptr(), s.data_ptr(), r.data_ptr(), _lib.stream_ptr()),
                   'hgn_radius_edges_fill')
    return s, r
