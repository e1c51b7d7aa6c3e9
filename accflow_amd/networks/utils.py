"""networks.utils surface used by test_cvo.py:8 and AccFlow_ (reference networks/utils.py:96-124)."""
from .raft.utils.utils import backwarp, coords_grid  # noqa: F401
