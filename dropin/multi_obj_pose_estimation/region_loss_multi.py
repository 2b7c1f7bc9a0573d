"""Drop-in for multi_obj_pose_estimation/region_loss_multi.py (`from region_loss_multi import RegionLoss`)."""
from singleshotpose_amd.region_loss import RegionLossMulti as RegionLoss  # noqa: F401
