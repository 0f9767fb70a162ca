"""reference: src/rmp/k_means_clustering.py."""
from hgn_amd.rmp import KMeansClustering  # noqa: F401
