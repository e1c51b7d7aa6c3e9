from accflow_amd.networks.raft.raft import RAFT  # noqa: F401
