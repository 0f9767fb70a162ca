"""reference: src/rmp/abstract_connector.py."""
from hgn_amd.rmp import AbstractConnector  # noqa: F401
