from accflow_amd.networks.utils import backwarp, coords_grid  # noqa: F401
