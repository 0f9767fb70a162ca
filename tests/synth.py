"""Test-side access to the synthetic graph generator (lives in the package so bench.py can use it too)."""
import importlib.util
import os
import sys

_PKG = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'hyper-graph-nets_amd')
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
from hgn_amd.synthetic import *          # noqa: F401,F403,E402
from hgn_amd.synthetic import grid_graph, batch, EdgeSet, MultiGraph   # noqa: F401,E402
