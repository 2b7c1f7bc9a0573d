"""Drop-in for the reference's darknet.py (train.py / valid.py do `from darknet import Darknet`)."""
from singleshotpose_amd.darknet import (Darknet, EmptyModule, GlobalAvgPool2d, MaxPoolStride1, Reorg)  # noqa: F401
import _compat  # noqa: F401,E402  (torch-0.4 Tensor.__array__ behaviour for the unchanged drivers)
