from accflow_amd.networks.AccPlus import AccFlow, AccPlus  # noqa: F401
