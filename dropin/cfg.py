"""Drop-in for the reference's top-level cfg.py: put this directory on PYTHONPATH ahead of the reference checkout."""
from singleshotpose_amd.cfg import *  # noqa: F401,F403
from singleshotpose_amd.cfg import (load_conv, load_conv_bn, load_fc, parse_cfg, print_cfg, save_conv,  # noqa: F401
                                    save_conv_bn, save_fc)
