"""`from data import dataset` as in the reference's test_cvo.py:5."""
from accflow_amd.data.dataset import CVO, CVO_sampler_lmdb, SyntheticCVO, fetch_valid_dataloader, totensor  # noqa: F401
