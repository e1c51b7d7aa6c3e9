from ...raft.utils.utils import backwarp, coords_grid  # noqa: F401
