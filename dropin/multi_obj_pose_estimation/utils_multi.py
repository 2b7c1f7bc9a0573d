"""Drop-in for multi_obj_pose_estimation/utils_multi.py (`from utils_multi import *`)."""
from singleshotpose_amd.utils_multi import *  # noqa: F401,F403
from singleshotpose_amd.utils_multi import bbox_iou, get_multi_region_boxes, nms  # noqa: F401
