"""reference: src/util.py -- EdgeSet, MultiGraph, MultiGraphWithPos, NodeType, device, detach, read_yaml,
triangles_to_edges, unsorted_segment_operation (src/util.py:10-134) on the HIP kernels."""
from hgn_amd.util import (EdgeSet, MultiGraph, MultiGraphWithPos, NodeType, detach, device, read_yaml,  # noqa: F401
                          triangles_to_edges, unsorted_segment_operation)
