"""Per-topology CSR cache: receiver-sorted edge order (and the sender-sorted twin for the backward scatter).

The reference rebuilds an [E, 128] int64 id broadcast on every aggregation call (src/util.py:107-110) and moves
index tensors to the device per layer (graphnet.py:25-26).  Here the index work is done ONCE per edge-index
CONTENT: a stable radix sort by receiver gives ``perm`` / ``rowptr``; all edge latents then live in that sorted
order for the whole processor, so aggregation streams them and `h[receiver]` gathers are quasi-sequential.

Cache levels of ``edge_topology`` (cheapest first):
  1. the tensor OBJECT seen before (attribute on the receivers tensor) -- a batch reused for many steps;
  2. an explicit topology key the producer attached (``tag_topology``: the batcher / ``build_graph_batch`` know that a batch
     is B copies of one mesh, so (mesh key, B) names the topology without looking at the ids);
  3. the index CONTENT (64-bit fingerprints from ``hgn_index_fingerprint``, one small kernel + a 16-byte read-back): the
     reference's fit loop builds FRESH batched index tensors for every batch (MeshSimulator.py:136,159-234) although all
     batches of a trajectory share one mesh -- equal content finds the topology built for the first batch, so the two radix
     sorts run once per mesh and batch size, not once per step.
"""
import collections
import ctypes as C
import weakref

import torch

from . import _lib


class CSR:
    """perm[j] = original position of sorted element j (stable), seg[j] = its segment id, rowptr[n..n+1] its row."""
    __slots__ = ('perm', 'seg', 'rowptr', 'num_segments', 'num_items', 'max_rows')

    def __init__(self, ids: torch.Tensor, num_segments: int):
        _lib.require_gpu(ids)
        L = _lib.lib()
        ids = ids.contiguous().to(torch.int64)
        E = ids.numel()
        dev = ids.device
        nb = C.c_size_t(0)
        _lib.check(L.hgn_csr_workspace_bytes(E, num_segments, C.byref(nb)), 'hgn_csr_workspace_bytes')
        ws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
        self.perm = torch.empty(max(E, 1), dtype=torch.int32, device=dev)[:E]
        self.seg = torch.empty(max(E, 1), dtype=torch.int32, device=dev)[:E]
        self.rowptr = torch.empty(num_segments + 1, dtype=torch.int32, device=dev)
        # longest segment, from the build's own read-back (no second sync): the fused in-kernel segment sums of the edge MLP
        # kernels are bit-reproducible only while no segment spans more than two 64-row tiles (include/hgn_mp.h: seg_out)
        mx = C.c_int32(0)
        _lib.check(L.hgn_csr_build(ids.data_ptr(), E, num_segments, self.perm.data_ptr(), self.seg.data_ptr(),
                                   self.rowptr.data_ptr(), ws.data_ptr(), nb.value, C.byref(mx), _lib.stream_ptr()),
                   'hgn_csr_build')
        self.num_segments = num_segments
        self.num_items = E
        self.max_rows = int(mx.value)


class EdgeTopology:
    """Everything index-shaped the kernels need for one edge set over ``num_nodes`` (mesh + hyper) rows.

    r   : CSR by receiver over the user's edge order   (perm = sorted position -> user position)
    snd : senders in receiver-sorted order (int32);  rcv = r.seg
    s   : CSR by sender over the *receiver-sorted* order (used only by the backward sender scatter)
    """
    __slots__ = ('r', 'snd', 'rcv', 's', 'num_nodes', 'num_edges', 'inv_perm', 'span', '__weakref__')

    def __init__(self, senders: torch.Tensor, receivers: torch.Tensor, num_nodes: int, device):
        L = _lib.lib()
        senders = senders.to(device=device, dtype=torch.int64).contiguous()
        receivers = receivers.to(device=device, dtype=torch.int64).contiguous()
        if senders.shape != receivers.shape or senders.dim() != 1:
            raise ValueError('senders / receivers must be 1-D tensors of equal length')
        E = senders.numel()
        self.num_nodes, self.num_edges = num_nodes, E
        self.r = CSR(receivers, num_nodes)
        self.rcv = self.r.seg
        self.snd = torch.empty(max(E, 1), dtype=torch.int32, device=device)[:E]
        _lib.check(L.hgn_narrow_gather_i64(senders.data_ptr(), self.r.perm.data_ptr(), E, self.snd.data_ptr(),
                                           _lib.stream_ptr()), 'hgn_narrow_gather_i64')
        self.s = CSR(self.snd, num_nodes)     # also range-checks the senders
        self.inv_perm = None
        # lowest / highest sender and receiver row (one 32-byte read-back per topology BUILD): which node parts the set touches
        self.span = (tuple(int(x) for x in torch.stack([self.snd.min(), self.snd.max(), self.rcv[0], self.rcv[-1]]).tolist())
                     if E > 0 else None)

    def parts(self, n_mesh: int):
        """(sender part, receiver part) of a graph whose node rows are [mesh rows | hyper rows] split at ``n_mesh`` -- 0 / 1 -- or None
        when the set is empty or either side has rows in both parts (the caller then hands over all rows as one tensor)."""
        if self.span is None:
            return None
        s0, s1, r0, r1 = self.span
        ps = 0 if s1 < n_mesh else (1 if s0 >= n_mesh else None)
        pr = 0 if r1 < n_mesh else (1 if r0 >= n_mesh else None)
        return None if ps is None or pr is None else (ps, pr)

    def inverse_perm(self):
        if self.inv_perm is None:
            inv = torch.empty_like(self.r.perm, dtype=torch.int64)
            inv[self.r.perm.long()] = torch.arange(self.num_edges, device=inv.device)
            self.inv_perm = inv
        return self.inv_perm


_CACHE_ATTR = '_hgn_topology'
_KEY_ATTR = '_hgn_topology_key'
_MAX_ENTRIES = 64
_by_key = collections.OrderedDict()         # explicit producer key or content fingerprint -> EdgeTopology  (LRU)
stats = {'object_hits': 0, 'key_hits': 0, 'content_hits': 0, 'builds': 0}


def clear_cache() -> None:
    """Forget every keyed / content-addressed topology (object-attached ones die with their tensors)."""
    _by_key.clear()


def tag_topology(senders: torch.Tensor, receivers: torch.Tensor, key) -> None:
    """Producer side of cache level 2: name the topology these index tensors describe.  Equal keys MUST mean equal index content
    (e.g. (id of the mesh's edge tensors, batch size, n_mesh, n_hyper, mapping mode) for a disjoint union of one mesh)."""
    for t in (senders, receivers):
        try:
            setattr(t, _KEY_ATTR, key)
        except Exception:
            pass


def _lookup(key):
    hit = _by_key.get(key)
    if hit is not None:
        _by_key.move_to_end(key)
    return hit


def _store(key, topo):
    _by_key[key] = topo
    while len(_by_key) > _MAX_ENTRIES:
        _by_key.popitem(last=False)


def _fingerprint(senders: torch.Tensor, receivers: torch.Tensor, device):
    """(E, word(senders), word(receivers)) of the index CONTENT; one launch + one 16-byte read-back."""
    s = senders.to(device=device, dtype=torch.int64).contiguous()
    r = receivers.to(device=device, dtype=torch.int64).contiguous()
    if s.shape != r.shape or s.dim() != 1:
        raise ValueError('senders / receivers must be 1-D tensors of equal length')
    scratch = torch.empty(2, dtype=torch.int64, device=device)
    host = (C.c_uint64 * 2)()
    _lib.check(_lib.lib().hgn_index_fingerprint(s.data_ptr(), r.data_ptr(), s.numel(), scratch.data_ptr(), host, _lib.stream_ptr()),
               'hgn_index_fingerprint')
    return s, r, (int(s.numel()), int(host[0]), int(host[1]))


def edge_topology(senders: torch.Tensor, receivers: torch.Tensor, num_nodes: int, device) -> EdgeTopology:
    okey = (id(senders), senders._version, receivers._version, num_nodes, str(device), senders.shape[0])
    cached = getattr(receivers, _CACHE_ATTR, None)
    if cached is not None and cached[0] == okey and cached[2]() is senders:
        stats['object_hits'] += 1
        return cached[1]
    topo = None
    pkey = getattr(receivers, _KEY_ATTR, None)
    if pkey is not None and getattr(senders, _KEY_ATTR, None) == pkey:
        pkey = ('producer', pkey, num_nodes, str(device), senders.shape[0])
        topo = _lookup(pkey)
        if topo is not None:
            stats['key_hits'] += 1
    else:
        pkey = None
    if topo is None:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.HgnError('topology lookup during HIP-graph capture: warm the topology cache with one eager call first')
        s, r, fp = _fingerprint(senders, receivers, device)
        ckey = ('content', fp, num_nodes, str(device))
        topo = _lookup(ckey)
        if topo is not None:
            stats['content_hits'] += 1
        else:
            topo = EdgeTopology(s, r, num_nodes, device)
            stats['builds'] += 1
            _store(ckey, topo)
        if pkey is not None:
            _store(pkey, topo)
    try:
        setattr(receivers, _CACHE_ATTR, _lib.Volatile((okey, topo, weakref.ref(senders))))
    except Exception:
        pass
    return topo


def segment_csr(segment_ids: torch.Tensor, num_segments: int, device) -> CSR:
    key = (segment_ids._version, num_segments, str(device), segment_ids.shape[0])
    cached = getattr(segment_ids, '_hgn_csr', None)
    if cached is not None and cached[0] == key:
        return cached[1]
    csr = CSR(segment_ids.to(device), num_segments)
    try:
        setattr(segment_ids, '_hgn_csr', _lib.Volatile((key, csr)))
    except Exception:
        pass
    return csr
