"""Stand-in for torch_geometric.data (absent from the image), test tooling only: the attribute bag the reference uses."""
class Data:
    def __init__(self, x=None, edge_index=None, edge_attr=None, **kw):
        self.x, self.edge_index, self.edge_attr = x, edge_index, edge_attr
        self.__dict__.update(kw)

    @property
    def num_nodes(self):
        return self.x.shape[0] if self.x is not None else int(self.edge_index.max()) + 1
