"""reference: src/model/abstract_system_model.py:10-190."""
from hgn_amd.system_model import AbstractSystemModel  # noqa: F401
