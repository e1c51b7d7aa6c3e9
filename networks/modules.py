from accflow_amd.networks.modules import ZeroConv2d  # noqa: F401
