"""Primitives of the reference's ``src/util.py`` that sit on the hot path, with the same names and signatures."""
import collections
import enum

import numpy as np
import torch

device = torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')          # src/util.py:10
EdgeSet = collections.namedtuple('EdgeSet', ['name', 'features', 'senders', 'receivers'])      # src/util.py:11
MultiGraph = collections.namedtuple('MultiGraph', ['node_features', 'edge_sets'])              # src/util.py:12
MultiGraphWithPos = collections.namedtuple('MultiGraph', ['node_features', 'edge_sets', 'target_feature',
                                                          'mesh_features', 'model_type', 'node_dynamic',
                                                          'unnormalized_edges', 'obstacle_nodes'])   # src/util.py:14-16


def detach(tensor: torch.Tensor) -> np.array:
    return tensor.detach().cpu().numpy()


class NodeType(enum.IntEnum):                                                                  # src/util.py:27-35
    NORMAL = 0
    OBSTACLE = 1
    AIRFOIL = 2
    HANDLE = 3
    INFLOW = 4
    OUTFLOW = 5
    WALL_BOUNDARY = 6
    SIZE = 9


def read_yaml(config_name: str):
    """src/util.py:38-47: the YAML document whose ``name`` is 'DEFAULT'."""
    import yaml
    with open(f'configs/{config_name}.yaml', 'r') as stream:
        try:
            for doc in yaml.safe_load_all(stream):
                if doc['name'] == 'DEFAULT':
                    return doc
        except yaml.YAMLError as e:
            print(e)
            return None


def triangles_to_edges(faces: torch.Tensor, deform: bool = False):
    """src/util.py:50-89: unique undirected cell edges, then both directions (cells of 3 or, `deform`, 4 nodes).
    Device radix sort + unique (include/hgn_features.h: hgn_cells_to_edges); same row order as torch.unique(dim=0)."""
    from . import features
    s2, r2, n = features.cells_to_edges(faces.to(device), deform)
    return {'two_way_connectivity': (s2, r2), 'senders': s2[:n], 'receivers': r2[:n]}


def unsorted_segment_operation(data, segment_ids, num_segments, operation):
    """src/util.py:92-134 on the HIP segment-reduce kernel (no id broadcast, one pass, differentiable).

    ``data`` [E, ...] (any trailing shape, incl. 1-D), ``segment_ids`` [E] unsorted / repeated ids in
    [0, num_segments); empty segments give 0; result has ``data``'s dtype."""
    from . import ops, topology
    assert all([i in data.shape for i in segment_ids.shape]), "segment_ids.shape should be a prefix of data.shape"
    if operation not in ('sum', 'mean', 'max', 'min', 'std'):
        raise Exception('Invalid operation type!')
    if segment_ids.dim() != 1:
        segment_ids = segment_ids.reshape(segment_ids.shape[0], -1)[:, 0]
    data = data.to(device)
    E = data.shape[0]
    inner = 1
    for d in data.shape[1:]:
        inner *= int(d)
    flat = data.reshape(E, inner).float()
    csr = topology.segment_csr(segment_ids, int(num_segments), data.device)
    if operation == 'std':                                  # util.py:129-130: torch_scatter.scatter_std (unbiased), its own kernels
        out = ops.segment_std(flat, (csr.perm, csr.rowptr, csr.seg))
    else:
        out = ops.aggregate([flat], [(csr.perm, csr.rowptr, csr.seg)], (operation,))
    return out.reshape((int(num_segments),) + tuple(data.shape[1:])).type(data.dtype)
