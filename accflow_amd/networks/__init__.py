"""Model factory with the reference's contract (networks/__init__.py:4-23): the estimator is picked
by substring ("raft" / "gma") and its hyper-parameters are fixed here."""
import argparse


def build_flow_estimator(name):
    lowered = name.lower()
    if "raft" in lowered:
        from .raft.raft import RAFT
        return RAFT(argparse.Namespace(small=False, mixed_precision=True))
    if "gma" in lowered:
        from .gma.gma import RAFTGMA
        return RAFTGMA(argparse.Namespace(num_heads=1, mixed_precision=True, position_only=False,
                                          position_and_content=False))
    raise NotImplementedError("not supported yet..")
