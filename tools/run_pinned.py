#!/usr/bin/env python
"""Runs an UNMODIFIED driver script (the reference's train.py) with its randomness pinned, so that two runs - the
reference on the CPU, the HIP drop-in on the MI355X - draw the same shuffles, augmentations and initial head weights:

    PYTHONPATH=<repo>:<repo>/dropin python tools/run_pinned.py [--seed N] /path/to/singleshotpose/train.py <its args>

train.py seeds torch with int(time.time()) (train.py:326-330) and never seeds `random`, which dataset.py / image.py
use for shuffling and augmentation.  The script file is executed as __main__ from its own directory entry on sys.path,
exactly as `python train.py` would; nothing in it is edited.  What is pinned: random.seed / numpy.random.seed once, and
torch.manual_seed / torch.cuda.manual_seed are wrapped to ignore the wall-clock value they are handed.
"""
import random
import runpy
import sys


def pin(seed):
    import numpy as np
    import torch
    random.seed(seed)
    np.random.seed(seed)
    real = torch.manual_seed
    real(seed)
    torch.manual_seed = lambda _ignored=None: real(seed)
    if hasattr(torch.cuda, 'manual_seed'):
        torch.cuda.manual_seed = lambda _ignored=None: None


def main(argv):
    seed = 0
    if argv and argv[0] == '--seed':
        seed, argv = int(argv[1]), argv[2:]
    if not argv:
        raise SystemExit(__doc__)
    script = argv[0]
    pin(seed)
    sys.argv = argv
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))      # what `python script.py` puts in front
    runpy.run_path(script, run_name='__main__')


if __name__ == '__main__':
    main(sys.argv[1:])
