"""reference: src/rmp/get_rmp.py:19-96."""
from hgn_amd.rmp import RemoteMessagePassing, get_clustering_algorithm, get_connector, get_rmp  # noqa: F401
