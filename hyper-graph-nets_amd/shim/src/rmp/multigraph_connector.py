"""reference: src/rmp/multigraph_connector.py."""
from hgn_amd.rmp import MultigraphConnector  # noqa: F401
