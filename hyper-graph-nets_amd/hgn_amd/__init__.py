"""MI355X-native message-passing core for HyperGraphNets.

Public surface mirrors the reference's hot-path modules:
    hgn_amd.modules     <-> src/migration/{meshgraphnet,graphnet,hypergraphnet,...,encoder,processor,decoder}.py
    hgn_amd.normalizer  <-> src/migration/normalizer.py
    hgn_amd.system_model <-> src/model/{abstract_system_model,flag,cylinder,plate,get_model}.py
    hgn_amd.rmp         <-> src/rmp/*.py (remote message passing: clustering on the host, graph assembly on the device)
    hgn_amd.features    <-> the feature arithmetic of src/model/*.py, src/util.triangles_to_edges (include/hgn_features.h)
    hgn_amd.util        <-> src/util.py  (EdgeSet, MultiGraph, device, unsorted_segment_operation, ...)
Everything numeric runs in libhgn_mp.so (hand-written HIP for gfx950); importing the package does not load it,
the first kernel call does, and fails loudly if it is not built.
"""
from .util import EdgeSet, MultiGraph, MultiGraphWithPos, NodeType, device, unsorted_segment_operation  # noqa: F401
from .modules import (MeshGraphNet, GraphNet, HyperGraphNet, HeteroGraphNet, MultiScaleGraphNet, MultiGraphNet,  # noqa: F401
                      RepeatedGraphNet, Encoder, Processor, Decoder, LazyMLP)
from .normalizer import Normalizer  # noqa: F401
from .ops import set_matmul_precision, get_matmul_precision  # noqa: F401,E402
