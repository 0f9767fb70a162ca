"""Import-only stub (test tooling)."""
from .dataset import TFRecordDataset  # noqa: F401
