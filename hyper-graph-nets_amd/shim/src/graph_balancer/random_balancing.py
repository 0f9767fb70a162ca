"""reference: src/graph_balancer/random_balancing.py."""
from hgn_amd.graph_balancer import RandomGraphBalancer  # noqa: F401
