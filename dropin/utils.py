"""Drop-in for the reference's utils.py (`from utils import *` in train.py / valid.py / dataset.py)."""
from singleshotpose_amd.utils import *  # noqa: F401,F403
import _compat  # noqa: F401,E402  (torch-0.4 Tensor.__array__ behaviour for the unchanged drivers)
