"""reference: src/model/plate.py:21-340."""
from hgn_amd.system_model import PlateModel  # noqa: F401
