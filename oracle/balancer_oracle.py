"""CPU oracle for the graph balancers (src/graph_balancer): balanced Forman curvature, its post-insertion delta, the SDRF
rewiring loop, random balancing and the 'balance' edge set.

TEST INFRASTRUCTURE ONLY (same rule as mgn_oracle.py).  numpy, dense N x N like the reference (small graphs only).
Arithmetic follows the typing of the reference's numba kernels: integer literals promote float32 operands to float64, results
are rounded when stored into the float32 matrices (ricci.py:152-197, 199-270).

Pinning: tests/golden/balancer.pt, produced by running the reference's own code -- its two numba-CUDA kernel bodies executed
thread by thread on the CPU (tools/oracle_shims/numba), its SDRF loop, RandomGraphBalancer and FlagModel.expand_graph with a
balancer -- in the build container (generator tests/golden/gen_golden_balancer.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import networkx as nx
import numpy as np
import torch

from . import features_oracle as FO
from .mgn_oracle import EdgeSet


def dense_adjacency(senders, receivers) -> np.ndarray:
    """ricci.py:53-56: undirected, coalesced, self loops removed; N = largest index + 1 (torch_geometric.to_dense_adj)."""
    s, r = np.asarray(senders), np.asarray(receivers)
    n = int(max(s.max(), r.max())) + 1
    A = np.zeros((n, n), np.float32)
    A[s, r] = 1
    A[r, s] = 1
    np.fill_diagonal(A, 0)
    return A


def forman_curvature(A: np.ndarray) -> np.ndarray:
    """ricci.py:128-197 (balanced Forman curvature of every edge; 0 elsewhere)."""
    A = A.astype(np.float32)
    N = A.shape[0]
    A2 = (A @ A).astype(np.float32)
    d_in, d_out = A.sum(0), A.sum(1)
    C = np.zeros((N, N), np.float32)
    for i, j in zip(*np.nonzero(A)):
        dmax, dmin = (d_in[i], d_out[j]) if d_in[i] > d_out[j] else (d_out[j], d_in[i])
        if dmax * dmin == 0:
            continue
        t1 = A[:, j] * (A2[i, :] - A[i, :]) * A[i, j]
        t2 = A[i, :] * (A2[:, j] - A[:, j]) * A[i, j]
        sharp = int((t1 > 0).sum() + (t2 > 0).sum())
        lam = float(max(t1.max(initial=0.0), t2.max(initial=0.0), 0.0))
        dmax, dmin = float(dmax), float(dmin)
        c = np.float32(((2 / dmax) + (2 / dmin) - 2) + (2 / dmax + 1 / dmin) * float(A2[i, j]) * float(A[i, j]))
        if lam > 0:
            c = np.float32(float(c) + sharp / (dmax * lam))
        C[i, j] = c
    return C


def post_delta(A: np.ndarray, x: int, y: int, i_neighbors: Sequence[int], j_neighbors: Sequence[int]) -> np.ndarray:
    """ricci.py:199-301: curvature of edge (x, y) after inserting (i, j), for i in i_neighbors, j in j_neighbors."""
    A = A.astype(np.float32)
    N = A.shape[0]
    A2 = (A @ A).astype(np.float32)
    d_in_x0, d_out_y0 = float(A[:, x].sum()), float(A[y].sum())
    D = np.zeros((len(i_neighbors), len(j_neighbors)), np.float32)
    z = np.arange(N)
    for I, i in enumerate(i_neighbors):
        for J, j in enumerate(j_neighbors):
            if i == j or A[i, j] != 0:
                D[I, J] = -1000
                continue
            d_in_x, d_out_y = d_in_x0, d_out_y0
            if j == x:
                d_in_x += 1
            elif i == y:
                d_out_y += 1
            if d_in_x * d_out_y == 0:
                D[I, J] = 0
                continue
            dmax, dmin = (d_in_x, d_out_y) if d_in_x > d_out_y else (d_out_y, d_in_x)
            a2xy = float(A2[x, y])
            if x == i and A[j, y] != 0:
                a2xy += float(A[j, y])
            elif y == j and A[x, i] != 0:
                a2xy += float(A[x, i])
            Azy = A[:, y].astype(np.float64) + ((z == i) & (y == j))
            Axz = A[x, :].astype(np.float64) + ((x == i) & (z == j))
            A2zy = A2[:, y].astype(np.float64) + np.where((z == i) & (A[j, y] != 0), A[j, y], 0) + \
                np.where((y == j) & (A[:, i] != 0), A[:, i], 0)
            A2xz = A2[x, :].astype(np.float64) + np.where((x == i) & (A[j, :] != 0), A[j, :], 0) + \
                np.where((z == j) & (A[x, i] != 0), A[x, i], 0)
            t1 = Azy * (A2xz - Axz) * float(A[x, y])
            t2 = Axz * (A2zy - Azy) * float(A[x, y])
            sharp = int((t1 > 0).sum() + (t2 > 0).sum())
            lam = float(max(t1.max(initial=0.0), t2.max(initial=0.0), 0.0))
            d = np.float32(((2 / dmax) + (2 / dmin) - 2) + (2 / dmax + 1 / dmin) * a2xy * float(A[x, y]))
            if lam > 0:
                d = np.float32(float(d) + sharp / (dmax * lam))
            D[I, J] = d
    return D


def softmax(a: np.ndarray, tau: float = 1) -> np.ndarray:
    e = np.exp(a * tau)                                               # ricci.py:303-306
    return e / e.sum()


def sdrf(senders, receivers, num_nodes: int, loops: int = 10, remove_edges: bool = False, removal_bound: float = 0.5,
         tau: float = 1) -> Tuple[Dict[str, List[int]], Dict[str, List[int]]]:
    """ricci.py:43-126 (undirected).  Draws from numpy's global generator exactly where the reference does."""
    A = dense_adjacency(senders, receivers)
    N = A.shape[0]
    G = nx.DiGraph()
    G.add_nodes_from(range(num_nodes))
    G.add_edges_from(zip(np.asarray(senders).tolist(), np.asarray(receivers).tolist()))
    G = G.to_undirected()
    added = {'senders': [], 'receivers': []}
    removed = {'senders': [], 'receivers': []}
    for _ in range(loops):
        can_add = True
        C = forman_curvature(A)
        ix = int(C.argmin())
        x, y = ix // N, ix % N
        xn = list(G.neighbors(x)) + [x]
        yn = list(G.neighbors(y)) + [y]
        cand = [(i, j) for i in xn for j in yn if i != j and not G.has_edge(i, j)]
        if cand:
            D = post_delta(A, x, y, xn, yn)
            imp = [float(np.float32(D[xn.index(i), yn.index(j)] - C[x, y])) for i, j in cand]
            k, l = cand[np.random.choice(range(len(cand)), p=softmax(np.array(imp), tau=tau))]
            G.add_edge(k, l)
            added['senders'].extend([k, l]); added['receivers'].extend([l, k])
            A[k, l] = A[l, k] = 1
        else:
            can_add = False
            if not remove_edges:
                break
        if remove_edges:
            ix = int(C.argmax())
            x, y = ix // N, ix % N
            if C[x, y] > removal_bound:
                G.remove_edge(x, y)
                removed['senders'].extend([x, y]); removed['receivers'].extend([y, x])
                A[x, y] = A[y, x] = 0
            elif not can_add:
                break
    return added, removed


def random_balance(num_vertices: int, edge_amount: int, remove_edges: bool):
    """random_balancing.py:19-36."""
    pairs = np.random.choice(num_vertices, size=(edge_amount, 2), replace=False)
    added = {'senders': [int(e[0]) for e in pairs], 'receivers': [int(e[1]) for e in pairs]}
    if not remove_edges:
        return added, None
    pairs = np.random.choice(num_vertices, size=(edge_amount, 2), replace=False)
    return added, {'senders': [int(e[0]) for e in pairs], 'receivers': [int(e[1]) for e in pairs]}


def determine_mask(senders: torch.Tensor, receivers: torch.Tensor, removed: Dict) -> torch.Tensor:
    """abstract_graph_balancer.py:73-81: drop every mesh edge whose endpoints form a removed (undirected) pair."""
    pairs = {frozenset((int(a), int(b))) for a, b in zip(removed['senders'], removed['receivers'])}
    return torch.tensor([frozenset((int(s), int(r))) not in pairs for s, r in zip(senders.tolist(), receivers.tolist())],
                        dtype=torch.bool)


def apply_balancer(graph: dict, added: Dict, mask: Optional[torch.Tensor], mesh_edge_normalizer, is_training: bool) -> list:
    """abstract_graph_balancer.py:48-71,83-95 on a graph from features_oracle.*.build_graph -> new edge-set list.
    The 'balance' features go through the mesh-edge normaliser first, then (if a mask exists) the masked, ALREADY
    normalised mesh-edge features go through it again (the reference normalises them twice)."""
    s = torch.tensor([int(v) for v in added['senders']], dtype=torch.long)
    r = torch.tensor([int(v) for v in added['receivers']], dtype=torch.long)
    feats = FO.rel_features(graph['target_feature'], graph['mesh_features'], s, r)
    sets = list(graph['edge_sets']) + [EdgeSet('balance', mesh_edge_normalizer(feats, is_training), s, r)]
    if mask is not None:
        m = sets[0]
        sets = [EdgeSet(m.name, mesh_edge_normalizer(m.features[mask, :], is_training), m.senders[mask], m.receivers[mask]),
                sets[1]]
    return sets
