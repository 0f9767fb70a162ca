"""reference: src/migration/meshgraphnet.py:21-108."""
from hgn_amd.modules import LazyMLP, MeshGraphNet  # noqa: F401
