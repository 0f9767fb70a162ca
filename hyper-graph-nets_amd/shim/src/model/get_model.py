"""reference: src/model/get_model.py:13-22."""
from hgn_amd.system_model import AbstractSystemModel, CylinderModel, FlagModel, PlateModel, get_model  # noqa: F401
