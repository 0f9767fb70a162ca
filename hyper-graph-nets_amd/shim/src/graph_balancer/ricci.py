"""reference: src/graph_balancer/ricci.py:43-301."""
from hgn_amd.graph_balancer import Ricci  # noqa: F401
