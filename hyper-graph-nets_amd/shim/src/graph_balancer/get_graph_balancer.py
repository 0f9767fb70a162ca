"""reference: src/graph_balancer/get_graph_balancer.py:11-27."""
from hgn_amd.graph_balancer import (AbstractGraphBalancer, GraphBalancer, RandomGraphBalancer, Ricci, get_balancer,  # noqa: F401
                                    get_balancer_algorithm)
