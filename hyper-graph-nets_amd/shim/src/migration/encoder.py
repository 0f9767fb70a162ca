"""reference: src/migration/encoder.py:9-47."""
from hgn_amd.modules import Encoder  # noqa: F401
