"""reference: src/rmp/random_clustering.py."""
from hgn_amd.rmp import RandomClustering  # noqa: F401
