"""reference: src/rmp/remote_message_passing.py."""
from hgn_amd.rmp import RemoteMessagePassing  # noqa: F401
