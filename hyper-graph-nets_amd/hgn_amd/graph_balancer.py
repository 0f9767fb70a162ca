"""Graph balancers: extra 'balance' edges (and removed mesh edges) chosen once per trajectory, then re-featured every frame
(reference: src/graph_balancer/{abstract_graph_balancer,graph_balancer,random_balancing,ricci,get_graph_balancer}.py).

Split as in rmp.py: WHICH edges to add / remove is a once-per-trajectory step with host control flow -- random pairs, or
SDRF (stochastic discrete Ricci flow: repeatedly take the most negatively curved edge, sample a neighbouring non-edge by
softmax of the curvature improvement, optionally drop the most positively curved edge).  The reference runs SDRF's two
curvature kernels as numba-CUDA thread-per-matrix-entry loops over a dense N x N adjacency; here they are wavefront-per-edge
HIP kernels (include/hgn_features.h: hgn_forman_curvature / hgn_forman_post_delta) that only visit existing edges and
reduce the four-cycle terms inside the wave, with A*A from the library GEMM.  The per-frame part (features of the balance
edges, masking the removed mesh edges, normalisation) is device work on the feature kernels.
"""
import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib, features
from .util import EdgeSet, MultiGraphWithPos, device


def _i32(values, dev) -> torch.Tensor:
    return torch.tensor([int(v) for v in values], dtype=torch.int32, device=dev)


def forman_curvature(A: torch.Tensor) -> torch.Tensor:
    """ricci.py:128-150: dense curvature matrix of the undirected 0/1 adjacency A (device fp32); 0 for non-edges."""
    _lib.require_gpu(A)
    A = A.float().contiguous()
    N = A.shape[0]
    A2 = torch.matmul(A, A)
    d_in, d_out = A.sum(dim=0), A.sum(dim=1)
    nz = torch.nonzero(A)
    ei, ej = nz[:, 0].to(torch.int32).contiguous(), nz[:, 1].to(torch.int32).contiguous()
    Cm = torch.zeros(N, N, dtype=torch.float32, device=A.device)
    _lib.check(_lib.lib().hgn_forman_curvature(A.data_ptr(), A2.data_ptr(), d_in.data_ptr(), d_out.data_ptr(), N,
                                               ei.data_ptr(), ej.data_ptr(), ei.shape[0], Cm.data_ptr(), _lib.stream_ptr()),
               'hgn_forman_curvature')
    return Cm


def forman_post_delta(A: torch.Tensor, x: int, y: int, i_neighbors, j_neighbors) -> torch.Tensor:
    """ricci.py:272-301."""
    _lib.require_gpu(A)
    A = A.float().contiguous()
    N = A.shape[0]
    A2 = torch.matmul(A, A)
    d_in_x, d_out_y = float(A[:, x].sum()), float(A[y].sum())
    ib, jb = _i32(i_neighbors, A.device), _i32(j_neighbors, A.device)
    D = torch.zeros(len(i_neighbors), len(j_neighbors), dtype=torch.float32, device=A.device)
    _lib.check(_lib.lib().hgn_forman_post_delta(A.data_ptr(), A2.data_ptr(), d_in_x, d_out_y, N, int(x), int(y), ib.data_ptr(),
                                                len(i_neighbors), jb.data_ptr(), len(j_neighbors), D.data_ptr(),
                                                _lib.stream_ptr()), 'hgn_forman_post_delta')
    return D


def softmax(a: np.ndarray, tau: float = 1) -> np.ndarray:
    e = np.exp(a * tau)                                               # ricci.py:303-306
    return e / e.sum()


def sdrf(senders: torch.Tensor, receivers: torch.Tensor, num_nodes: int, loops: int = 10, remove_edges: bool = False,
         removal_bound: float = 0.5, tau: float = 1) -> Tuple[Dict, Dict]:
    """ricci.py:43-126 for undirected graphs: same control flow, same draws from numpy's global generator; the curvature
    matrices come from the HIP kernels.  Candidate lists follow networkx's neighbour order, as in the reference."""
    import networkx as nx
    dev = device
    s, r = senders.to(dev).long(), receivers.to(dev).long()
    N = int(max(int(s.max()), int(r.max()))) + 1 if s.numel() else 0
    A = torch.zeros(N, N, dtype=torch.float32, device=dev)
    A[s, r] = 1
    A[r, s] = 1
    A.fill_diagonal_(0)
    G = nx.DiGraph()
    G.add_nodes_from(range(num_nodes))
    G.add_edges_from(zip(s.tolist(), r.tolist()))
    G = G.to_undirected()
    added = {'senders': [], 'receivers': []}
    removed = {'senders': [], 'receivers': []}
    for _ in range(loops):
        can_add = True
        Cm = forman_curvature(A)
        ix = int(Cm.argmin())
        x, y = ix // N, ix % N
        xn = list(G.neighbors(x)) + [x]
        yn = list(G.neighbors(y)) + [y]
        cand = [(i, j) for i in xn for j in yn if i != j and not G.has_edge(i, j)]
        if cand:
            D = (forman_post_delta(A, x, y, xn, yn) - Cm[x, y]).cpu()
            imp = [D[xn.index(i), yn.index(j)].item() for i, j in cand]
            k, l = cand[np.random.choice(range(len(cand)), p=softmax(np.array(imp), tau=tau))]
            G.add_edge(k, l)
            added['senders'].extend([k, l]); added['receivers'].extend([l, k])
            A[k, l] = 1
            A[l, k] = 1
        else:
            can_add = False
            if not remove_edges:
                break
        if remove_edges:
            ix = int(Cm.argmax())
            x, y = ix // N, ix % N
            if float(Cm[x, y]) > removal_bound:
                G.remove_edge(x, y)
                removed['senders'].extend([x, y]); removed['receivers'].extend([y, x])
                A[x, y] = 0
                A[y, x] = 0
            elif not can_add:
                break
    return added, removed


class AbstractGraphBalancer:
    """abstract_graph_balancer.py:9-99."""

    def __init__(self):
        self._added_edges = None
        self._mask = None
        self._added_dev = None

    def run(self, graph: MultiGraphWithPos):
        raise NotImplementedError

    @staticmethod
    def add_graph_balance_edges(graph: MultiGraphWithPos, added_edges: Dict, mesh_edge_normalizer, is_training: bool,
                                ids=None) -> MultiGraphWithPos:
        """abstract_graph_balancer.py:48-63: relative world / mesh position features of the added pairs -> 'balance' set."""
        world_pos, mesh_pos = graph.target_feature.to(device), graph.mesh_features.to(device)
        if ids is None:
            ids = (torch.tensor([int(v) for v in added_edges['senders']], dtype=torch.long, device=device),
                   torch.tensor([int(v) for v in added_edges['receivers']], dtype=torch.long, device=device))
        feats, _ = features.rel_edge_features(world_pos, mesh_pos, ids[0], ids[1])
        graph.edge_sets.append(EdgeSet(name='balance', features=mesh_edge_normalizer(feats, is_training), senders=ids[0],
                                       receivers=ids[1]))
        return graph

    def remove_graph_balance_edges(self, graph: MultiGraphWithPos, mask: torch.Tensor, mesh_edge_normalizer,
                                   is_training: bool) -> MultiGraphWithPos:
        """abstract_graph_balancer.py:65-71 (keeps the reference's double normalisation of the surviving mesh edges and its
        [masked mesh, second set] edge-set list)."""
        e = graph.edge_sets[0]
        keep = self._keep_idx
        ge = EdgeSet(name=e.name, features=mesh_edge_normalizer(e.features.index_select(0, keep), is_training),
                     senders=self._kept[0], receivers=self._kept[1])
        return graph._replace(edge_sets=[ge, graph.edge_sets[1]])

    @staticmethod
    def _determine_mask(graph_edges: EdgeSet, removed_edges: Dict) -> torch.Tensor:
        """abstract_graph_balancer.py:73-81, vectorised: False for every mesh edge whose endpoints are a removed pair."""
        s, r = graph_edges.senders.to(device).long(), graph_edges.receivers.to(device).long()
        if len(removed_edges['senders']) == 0:
            return torch.ones(s.shape[0], dtype=torch.bool, device=device)
        a = torch.tensor([int(v) for v in removed_edges['senders']], dtype=torch.long, device=device)
        b = torch.tensor([int(v) for v in removed_edges['receivers']], dtype=torch.long, device=device)
        n = int(max(int(s.max()), int(r.max()), int(a.max()), int(b.max()))) + 1
        key = lambda u, v: torch.minimum(u, v) * n + torch.maximum(u, v)
        return ~torch.isin(key(s, r), key(a, b))

    def create_graph(self, graph: MultiGraphWithPos, mesh_edge_normalizer, is_training: bool) -> MultiGraphWithPos:
        """abstract_graph_balancer.py:83-95."""
        if self._added_edges is None:
            self._added_edges, removed_edges = self.run(graph)
            self._added_dev = (torch.tensor([int(v) for v in self._added_edges['senders']], dtype=torch.long, device=device),
                               torch.tensor([int(v) for v in self._added_edges['receivers']], dtype=torch.long, device=device))
            if removed_edges is not None:
                self._mask = self._determine_mask(graph.edge_sets[0], removed_edges)
                self._keep_idx = torch.nonzero(self._mask).squeeze(1)
                e = graph.edge_sets[0]
                self._kept = (e.senders.to(device).index_select(0, self._keep_idx).contiguous(),
                              e.receivers.to(device).index_select(0, self._keep_idx).contiguous())
        graph = self.add_graph_balance_edges(graph, self._added_edges, mesh_edge_normalizer, is_training, self._added_dev)
        if self._mask is not None:
            graph = self.remove_graph_balance_edges(graph, self._mask, mesh_edge_normalizer, is_training)
        return graph

    def reset_edges(self):
        self._added_edges = None

    def reset_mask(self):
        self._mask = None


class RandomGraphBalancer(AbstractGraphBalancer):
    """random_balancing.py:8-36."""

    def __init__(self, params):
        super().__init__()
        self._edge_amount = params.get('graph_balancer').get('random').get('edge_amount')
        self._remove_edges = params.get('graph_balancer').get('remove_edges')

    def run(self, graph: MultiGraphWithPos):
        n = graph.node_features[0].shape[0]
        pairs = np.random.choice(n, size=(self._edge_amount, 2), replace=False)
        added = {'senders': [int(e[0]) for e in pairs], 'receivers': [int(e[1]) for e in pairs]}
        if not self._remove_edges:
            return added, None
        pairs = np.random.choice(n, size=(self._edge_amount, 2), replace=False)
        return added, {'senders': [int(e[0]) for e in pairs], 'receivers': [int(e[1]) for e in pairs]}


class Ricci(AbstractGraphBalancer):
    """ricci.py:14-41."""

    def __init__(self, params):
        super().__init__()
        r = params.get('graph_balancer').get('ricci')
        self._loops, self._tau = r.get('loops'), r.get('tau')
        self._remove_edges = params.get('graph_balancer').get('remove_edges')

    def run(self, graph: MultiGraphWithPos):
        e = graph.edge_sets[0]
        added, removed = sdrf(e.senders, e.receivers, graph.node_features[0].shape[0], loops=self._loops,
                              remove_edges=self._remove_edges, tau=self._tau)
        return (added, removed) if self._remove_edges else (added, None)


class GraphBalancer:
    """graph_balancer.py:5-24."""

    def __init__(self, balancer: AbstractGraphBalancer):
        self._balancer = balancer

    def initialize(self):
        pass

    def create_graph(self, graph: MultiGraphWithPos, mesh_edge_normalizer, is_training: bool) -> MultiGraphWithPos:
        return self._balancer.create_graph(graph, mesh_edge_normalizer, is_training)

    def reset_balancer(self):
        self._balancer.reset_edges()
        self._balancer.reset_mask()


def get_balancer(config) -> GraphBalancer:
    """get_graph_balancer.py:11-15."""
    name = str(config['graph_balancer']['algorithm']).lower()
    return GraphBalancer(get_balancer_algorithm(name, config))


def get_balancer_algorithm(name: str, config):
    """get_graph_balancer.py:18-27."""
    if name == 'ricci':
        return Ricci(config)
    if name == 'random':
        return RandomGraphBalancer(config)
    if name == 'none':
        return None
    raise NotImplementedError('Implement your balancing algorithms here!')
