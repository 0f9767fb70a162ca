"""reference: src/migration/graphnet.py:11-124."""
from hgn_amd.modules import GraphNet  # noqa: F401
