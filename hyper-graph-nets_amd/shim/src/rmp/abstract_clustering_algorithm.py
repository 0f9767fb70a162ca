"""reference: src/rmp/abstract_clustering_algorithm.py."""
from hgn_amd.rmp import AbstractClusteringAlgorithm  # noqa: F401
