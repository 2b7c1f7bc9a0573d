#!/usr/bin/env python
"""One training step against the oracle (oracle/step_check.py) from the command line - test tooling.
   python tools/step_check_cli.py [--batch 64] [--size 416] [--cfg cfg/yolo-pose.cfg] [--opt name=value,...] [--top 8]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--cfg', default=os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
    ap.add_argument('--opt', default='')
    ap.add_argument('--top', type=int, default=8)
    ap.add_argument('--epoch', type=int, default=20)
    ap.add_argument('--exact', action='store_true', help='also the float64 yardstick (oracle/step_check.py)')
    args = ap.parse_args()
    from bench import synthetic_batch
    from oracle.step_check import check_train_step, summarize
    from singleshotpose_amd import _lib
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.region_loss import RegionLoss
    for kv in filter(None, args.opt.split(',')):
        k, v = kv.split('=')
        _lib.call('ssp_set_option', k.encode(), int(v))
    torch.manual_seed(0)
    model = Darknet(args.cfg).cuda()
    x, tgt = synthetic_batch(args.batch, args.size, args.size, 1000, 'cpu')
    res = check_train_step(model, RegionLoss(), x, tgt, args.epoch, exact=args.exact)
    worst = sorted(res['grad_by_param'].items(), key=lambda kv: -kv[1])[:args.top]
    print('STEPCHECK opt=%r env FIRST_FUSED=%s BN_FUSE=%s: %s' % (args.opt, os.environ.get('SSP_FIRST_FUSED', '1'),
                                                                    os.environ.get('SSP_BN_FUSE', '1'), summarize(res)))
    print('  worst grads:', [(k, float('%.3g' % v)) for k, v in worst])
    if args.exact:
        w64 = sorted(res['grad64_by_param'].items(), key=lambda kv: -kv[1][0])[:args.top]
        print('  vs float64 (product, fp32 oracle):', [(k, float('%.3g' % a), float('%.3g' % b)) for k, (a, b) in w64])
    print('  plans:', [(i, f, d) for i, f, d in res['plans'] if f or d])
    print('  fp64 fallbacks:', {k: {a: float('%.3g' % b) for a, b in v.items()} for k, v in res.get('grad_fp64_oracle', {}).items()})


if __name__ == '__main__':
    main()
