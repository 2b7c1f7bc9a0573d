#!/usr/bin/env python
"""Iterates `dataset.listDataset(train=True)` inside a DataLoader exactly as train.py:56-65 / :75-83 does and saves what
each batch holds once it is on the GPU - whichever `dataset` module PYTHONPATH resolves:

    PYTHONPATH=<reference callers dir>:<repo>:<repo>/dropin  ->  the reference's dataset.py + image.py (Pillow, host)
    PYTHONPATH=<repo>:<repo>/dropin                          ->  dropin/dataset.py (one GPU pass per batch)

    python tools/dump_dataset_epoch.py <fixture root> <out.npz> [--seed N] [--seen N] [--epochs N] [--batch N] [--workers N] [--cpu]

Run from the fixture root's parent is not needed: paths in the train list are relative to <fixture root>, the script
chdirs there like the training script's caller does.  Output: u8_<i> (B, H, W, 3) uint8 pixels of batch i (a float
ToTensor batch is turned back into the bytes it was divided from: exact), lab_<i> its labels, and `module` = the file the
`dataset` module came from.  tests/test_gpu_dropin.py runs it once per PYTHONPATH and compares the archives byte for byte.
"""
import argparse
import os
import random
import sys

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('root')
    ap.add_argument('out')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--seen', type=int, default=0)
    ap.add_argument('--epochs', type=int, default=1)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--workers', type=int, default=0)
    ap.add_argument('--cpu', action='store_true', help="no .cuda() (the reference's host pipeline in the build container)")
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.chdir(args.root)
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)

    import dataset
    from torchvision import transforms
    bgdir = os.path.join('VOCdevkit', 'VOC2012', 'JPEGImages')
    bg_file_names = [os.path.join(bgdir, f) for f in sorted(os.listdir(bgdir))]
    res = {'module': np.array(os.path.abspath(dataset.__file__))}
    n, seen = 0, args.seen
    for epoch in range(args.epochs):
        loader = torch.utils.data.DataLoader(
            dataset.listDataset(os.path.join('LINEMOD', 'ape', 'train.txt'), shape=(416, 416), shuffle=True,
                                transform=transforms.Compose([transforms.ToTensor(), ]), train=True, seen=seen,
                                batch_size=args.batch, num_workers=args.workers, bg_file_names=bg_file_names),
            batch_size=args.batch, shuffle=False, num_workers=args.workers, pin_memory=True)
        for data, target in loader:
            if not args.cpu:
                data = data.cuda()
            if data.dtype != torch.uint8:       # ToTensor: u8 / 255 in float32; * 255 and rounding gives the byte back
                data = (data * 255).round().to(torch.uint8).permute(0, 2, 3, 1)
            res['u8_%d' % n] = data.contiguous().cpu().numpy()
            res['lab_%d' % n] = target.numpy()
            seen += data.size(0)
            n += 1
    np.savez_compressed(out, **res)
    print('%d batches from %s' % (n, res['module']))


if __name__ == '__main__':
    sys.exit(main())
