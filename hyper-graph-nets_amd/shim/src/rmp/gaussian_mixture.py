"""reference: src/rmp/gaussian_mixture.py."""
from hgn_amd.rmp import GaussianMixtureClustering  # noqa: F401
