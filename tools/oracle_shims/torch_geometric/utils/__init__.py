"""Stand-in for the four torch_geometric.utils functions src/graph_balancer/ricci.py calls (torch_geometric is absent from the
image), test tooling only.  Published semantics restated: `to_undirected` adds the reverse edges and coalesces (sorted by
(row, col), duplicates removed); `remove_self_loops` drops i == i pairs and returns (edge_index, edge_attr); `to_dense_adj`
returns a [1, N, N] float tensor, N = max index + 1, entries = edge multiplicity; `to_networkx` builds a DiGraph over
range(num_nodes) with one edge per column of edge_index."""
import networkx as nx
import torch


def to_undirected(edge_index, *a, **k):
    row, col = edge_index
    both = torch.stack([torch.cat([row, col]), torch.cat([col, row])])
    n = int(both.max()) + 1 if both.numel() else 0
    key = torch.unique(both[0] * n + both[1])
    return torch.stack([key // n, key % n])


def remove_self_loops(edge_index, edge_attr=None):
    m = edge_index[0] != edge_index[1]
    return edge_index[:, m], (edge_attr[m] if edge_attr is not None else None)


def to_dense_adj(edge_index, *a, **k):
    n = int(edge_index.max()) + 1 if edge_index.numel() else 0
    A = torch.zeros(n, n)
    A.index_put_((edge_index[0], edge_index[1]), torch.ones(edge_index.shape[1]), accumulate=True)
    return A.unsqueeze(0)


def to_networkx(data, *a, **k):
    G = nx.DiGraph()
    G.add_nodes_from(range(data.num_nodes))
    G.add_edges_from(zip(data.edge_index[0].tolist(), data.edge_index[1].tolist()))
    return G


def from_networkx(*a, **k):
    raise RuntimeError('torch_geometric stand-in: from_networkx is not used by the reference path')
