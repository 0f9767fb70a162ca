"""reference: src/graph_balancer/graph_balancer.py."""
from hgn_amd.graph_balancer import GraphBalancer  # noqa: F401
