"""Drop-in import surface of the reference (test_cvo.py:5-8): `from networks import
build_flow_estimator`, `from networks.AccFlow_ import AccFlow`, `from networks.utils import backwarp`.
Everything is implemented in accflow_amd.networks on MI355X HIP kernels."""
from accflow_amd.networks import build_flow_estimator  # noqa: F401
