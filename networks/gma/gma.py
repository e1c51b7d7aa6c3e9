from accflow_amd.networks.gma.gma import RAFTGMA  # noqa: F401
