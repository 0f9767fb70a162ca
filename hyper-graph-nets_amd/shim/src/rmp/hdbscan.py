"""reference: src/rmp/hdbscan.py."""
from hgn_amd.rmp import HDBSCANClustering as HDBSCAN  # noqa: F401
