"""Drop-in for the reference's utils.py (`from utils import *` in train.py / valid.py / dataset.py)."""
from singleshotpose_amd.utils import *  # noqa: F401,F403
