"""reference: src/migration/multigraphnet.py:10-18."""
from hgn_amd.modules import MultiGraphNet  # noqa: F401
