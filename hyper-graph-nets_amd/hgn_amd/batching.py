"""Disjoint-union batching of graphs that share one topology signature.

Replaces the per-element Python list comprehension of ``MeshSimulator._get_batched`` (src/algorithms/MeshSimulator.py:159-234)
with a handful of vectorised tensor ops that run on whatever device the inputs live on (no host round trip, no sync).

Index mapping.  Node rows of the union are ordered [mesh rows of all graphs ; hyper rows of all graphs].
``reference_compat=False`` (default) maps hyper ids correctly: graph i's hyper id ``n_mesh + h`` becomes
``B*n_mesh + i*n_hyper + h``.  ``reference_compat=True`` reproduces the reference arithmetic bit for bit, including its
mis-mapping for batch sizes >= 2 (an id is treated as a hyper id only if it is >= ``B*n_mesh``, MeshSimulator.py:196,
206-208) -- pinned by golden G6 -- for parity runs only.
"""
from typing import Optional, Sequence

import torch

from .util import EdgeSet, MultiGraph


def batch_graphs(graphs: Sequence[MultiGraph], reference_compat: bool = False, batch_size: Optional[int] = None) -> MultiGraph:
    """``batch_size``: the configured batch size when it differs from ``len(graphs)`` (the last, shorter batch of a trajectory:
    the reference offsets hyper ids by the CONFIGURED size, MeshSimulator.py:196); only used with ``reference_compat``."""
    B = len(graphs)
    if B == 0:
        raise ValueError('need at least one graph')
    Bc = B if batch_size is None else int(batch_size)
    names = [e.name for e in graphs[0].edge_sets]
    n_parts = len(graphs[0].node_features)
    n_mesh = graphs[0].node_features[0].shape[0]
    n_hyp = graphs[0].node_features[1].shape[0] if n_parts > 1 else 0
    for g in graphs:
        if [e.name for e in g.edge_sets] != names or g.node_features[0].shape[0] != n_mesh or len(g.node_features) != n_parts \
                or (n_parts > 1 and g.node_features[1].shape[0] != n_hyp):
            raise ValueError('graphs of one batch must share edge-set names and node counts (the reference batches '
                             'consecutive time steps of one trajectory; its offsets are i * num_nodes, MeshSimulator.py:191-196)')
    sets = []
    for k, name in enumerate(names):
        feats = torch.cat([g.edge_sets[k].features for g in graphs], dim=0)
        # per-graph edge counts may differ (plate `world_edges`, `balance`, sampled remote sets vary per frame; the reference
        # concatenates per-graph lists, MeshSimulator.py:186-231): graph index per edge instead of a [B, E] stack
        lens = torch.tensor([g.edge_sets[k].senders.shape[0] for g in graphs])
        out = []
        for which in ('senders', 'receivers'):
            idx = torch.cat([getattr(g.edge_sets[k], which).long().reshape(-1) for g in graphs], dim=0)
            i = torch.repeat_interleave(torch.arange(B, dtype=idx.dtype), lens).to(idx.device)
            mesh = idx + i * n_mesh
            if reference_compat:
                hyp = idx + (Bc - 1) * n_mesh + i * n_hyp
                res = torch.where(idx < Bc * n_mesh, mesh, hyp)
            else:
                hyp = (idx - n_mesh) + B * n_mesh + i * n_hyp
                res = torch.where(idx < n_mesh, mesh, hyp)
            out.append(res)
        sets.append(EdgeSet(name, feats, out[0], out[1]))
    nodes = [torch.cat([g.node_features[j] for g in graphs], dim=0) for j in range(n_parts)]
    return MultiGraph(nodes, sets)
