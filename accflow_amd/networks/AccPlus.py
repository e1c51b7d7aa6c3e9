"""BASELINE.json names the accumulation module "networks.AccPlus"; the importable class of the
reference is networks.AccFlow_.AccPlus (its own networks/AccPlus.py cannot be imported: it needs the
missing networks.raft.softsplat, AccPlus.py:8).  This shim re-exports the live implementation."""
from .AccFlow_ import AccFlow, AccPlus, Blending, FlowDecoder, FlowEncoder, downflow8, getOcc  # noqa: F401
