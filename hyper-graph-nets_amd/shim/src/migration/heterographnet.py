"""reference: src/migration/heterographnet.py:10-33."""
from hgn_amd.modules import HeteroGraphNet  # noqa: F401
