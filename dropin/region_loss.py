"""Drop-in for the reference's region_loss.py (train.py does `from region_loss import RegionLoss`)."""
from singleshotpose_amd.region_loss import RegionLoss  # noqa: F401
