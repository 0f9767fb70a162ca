"""reference: src/rmp/spectral_clustering.py."""
from hgn_amd.rmp import SpectralClustering  # noqa: F401
