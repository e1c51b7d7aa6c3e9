"""`from data import dataset` as in the reference's test_cvo.py:5."""
from accflow_amd.data.dataset import SyntheticCVO, fetch_valid_dataloader  # noqa: F401
