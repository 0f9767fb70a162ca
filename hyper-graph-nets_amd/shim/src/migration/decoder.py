"""reference: src/migration/decoder.py:8-16."""
from hgn_amd.modules import Decoder  # noqa: F401
