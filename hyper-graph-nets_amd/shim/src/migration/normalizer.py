"""reference: src/migration/normalizer.py:9-75."""
from hgn_amd.normalizer import Normalizer  # noqa: F401
