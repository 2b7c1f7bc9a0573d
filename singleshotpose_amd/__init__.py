"""singleshotpose hot path on MI355X (gfx950): Darknet-19 yolo-pose forward/backward, RegionLoss, decode, PnP.

Host side mirrors the reference's Python surface (cfg.py, darknet.py, region_loss.py, utils.py); all device work is
hand-written HIP behind the C ABI in include/ssp_hip.h (libssp_hip.so).  See DESIGN.md.
"""
from . import _lib  # noqa: F401
from .cfg import parse_cfg, print_cfg  # noqa: F401
from .darknet import Darknet  # noqa: F401
from .region_loss import RegionLoss, RegionLossMulti  # noqa: F401
from . import optim  # noqa: F401

__all__ = ['Darknet', 'RegionLoss', 'RegionLossMulti', 'parse_cfg', 'print_cfg']
