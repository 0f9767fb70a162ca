"""reference: src/graph_balancer/abstract_graph_balancer.py."""
from hgn_amd.graph_balancer import AbstractGraphBalancer  # noqa: F401
