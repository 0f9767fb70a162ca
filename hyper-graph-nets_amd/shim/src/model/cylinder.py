"""reference: src/model/cylinder.py:17-232."""
from hgn_amd.system_model import CylinderModel  # noqa: F401
