"""gma/corr.py:8-58 is the same CorrBlock as RAFT's."""
from ..raft.corr import CorrBlock  # noqa: F401
