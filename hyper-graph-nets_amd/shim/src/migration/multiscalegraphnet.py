"""reference: src/migration/multiscalegraphnet.py:10-63."""
from hgn_amd.modules import MultiScaleGraphNet  # noqa: F401
