"""Import-only stub (test tooling)."""
def _na(*a, **k):
    raise RuntimeError("torch_geometric stub")
to_networkx = from_networkx = to_dense_adj = remove_self_loops = to_undirected = _na
