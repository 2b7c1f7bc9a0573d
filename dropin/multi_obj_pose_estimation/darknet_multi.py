"""Drop-in for multi_obj_pose_estimation/darknet_multi.py (`from darknet_multi import Darknet`)."""
from singleshotpose_amd.darknet import DarknetMulti as Darknet  # noqa: F401
from singleshotpose_amd.darknet import EmptyModule, GlobalAvgPool2d, MaxPoolStride1, Reorg  # noqa: F401
