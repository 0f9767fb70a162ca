"""reference: src/migration/hypergraphnet.py:11-54."""
from hgn_amd.modules import HyperGraphNet  # noqa: F401
