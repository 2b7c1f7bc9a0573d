"""singleshotpose hot path on MI355X (gfx950): Darknet-19 yolo-pose forward/backward, RegionLoss, decode, PnP.

Host side mirrors the reference's Python surface (cfg.py, darknet.py, region_loss.py, utils.py); all device work is
hand-written HIP behind the C ABI in include/ssp_hip.h (libssp_hip.so).  See DESIGN.md.
"""
import os as _os

# HIP hands a process's streams to GPU_MAX_HW_QUEUES hardware queues (default 4); with RCCL's own streams in the process the
# engine's two compute streams then share one queue and lose their overlap (bench.py has the numbers: 35.7 ms per step
# against 28.9 ms with 8 queues).  The runtime reads the variable when it initialises - this default only helps a process
# that imports the package before its first HIP call; multi-GPU launch scripts should export it themselves
# (INTEGRATION.md section 4).  An explicit setting wins.
if 'GPU_MAX_HW_QUEUES' not in _os.environ:
    _os.environ['GPU_MAX_HW_QUEUES'] = '8'
    try:        # too late when the HIP runtime is already up: say so instead of silently running with 4 queues
        import torch as _torch
        if _torch.cuda.is_available() and _torch.cuda.is_initialized():
            import warnings as _warnings
            _warnings.warn("singleshotpose_amd: the HIP runtime was initialised before this package was imported, so "
                           "GPU_MAX_HW_QUEUES=8 does not take effect in this process; under RCCL the step's two compute "
                           "streams may then share a hardware queue (~25 % slower steps).  Export GPU_MAX_HW_QUEUES=8 in the "
                           "launch script (tools/launch_dp.sh does).")
    except Exception:
        pass

from . import _lib  # noqa: F401
from .cfg import parse_cfg, print_cfg  # noqa: F401
from .darknet import Darknet  # noqa: F401
from .region_loss import RegionLoss, RegionLossMulti  # noqa: F401
from . import optim  # noqa: F401

__all__ = ['Darknet', 'RegionLoss', 'RegionLossMulti', 'parse_cfg', 'print_cfg']
