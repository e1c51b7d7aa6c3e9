"""gma/extractor.py:115-188 is the same BasicEncoder as RAFT's."""
from ..raft.extractor import BasicEncoder, ResidualBlock  # noqa: F401
