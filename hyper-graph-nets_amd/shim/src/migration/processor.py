"""reference: src/migration/processor.py:10-28."""
from hgn_amd.modules import Processor  # noqa: F401
