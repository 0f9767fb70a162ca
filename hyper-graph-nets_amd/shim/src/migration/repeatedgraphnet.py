"""reference: src/migration/repeatedgraphnet.py:11-22."""
from hgn_amd.modules import RepeatedGraphNet  # noqa: F401
