"""reference: src/rmp/hierarchical_connector.py:27-143."""
from hgn_amd.rmp import HierarchicalConnector  # noqa: F401
