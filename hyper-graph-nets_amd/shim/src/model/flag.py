"""reference: src/model/flag.py:17-260."""
from hgn_amd.system_model import FlagModel  # noqa: F401
