#!/usr/bin/env python
"""Runs the same training step twice in one process - default kernels and under an igemm_variant - and reports, layer by
layer in backward order, where the saved tensors (forward BN vectors, dL/d activation, dL/d raw, parameter gradients)
first differ.  Debugging tool.   SSP_TUNE_CACHE=<pinned plan set> python tools/diff_variants.py [variant]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def step(variant):
    from bench import synthetic_batch
    from singleshotpose_amd import _lib
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.region_loss import RegionLoss
    _lib.call('ssp_set_option', b'igemm_variant', variant)
    torch.manual_seed(0)
    model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg')).cuda().train()
    x, tgt = synthetic_batch(64, 416, 416, 1000, 'cuda')
    crit = RegionLoss()
    crit.verbose = False
    out = model(x)
    plan = list(model._plans.values())[0]
    fwd = {i: (cs.raw.clone(), cs.vec.clone()) for i, cs in plan.convs.items()}
    crit(out, tgt, 20).backward()
    torch.cuda.synchronize()
    _lib.call('ssp_set_option', b'igemm_variant', 0)
    rec = {}
    for i, cs in plan.convs.items():
        oind = i + 1 if cs.pool else i
        g = plan.grads.get(oind)
        rec[i] = dict(raw=fwd[i][0], vec=fwd[i][1], g=None if g is None else g.t.clone(), dx=cs.raw.clone(),
                      dw=cs.conv.weight.grad.clone(), vec_after=cs.vec.clone())
    return rec, [(i, cs.plan_fwd, cs.plan_dgrad) for i, cs in plan.convs.items()]


def rel(a, b):
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def main():
    v = int(sys.argv[1]) if len(sys.argv) > 1 else 63
    a, pa = step(0)
    b, pb = step(v)
    assert pa == pb, (pa, pb)
    print('plans', [p for p in pa if p[1] or p[2]])
    for i in sorted(a, reverse=True):
        ra, rb = a[i], b[i]
        line = 'layer %2d: raw %.1e vec(mean,istd,sc,sh) %s' % (i, rel(ra['raw'], rb['raw']),
                                                               ['%.1e' % rel(ra['vec'][k], rb['vec'][k]) for k in range(4)])
        if ra['g'] is not None:
            line += ' g %.1e' % rel(ra['g'], rb['g'])
        line += ' dx %.1e dw %.1e c1,c2 %s' % (rel(ra['dx'], rb['dx']), rel(ra['dw'], rb['dw']),
                                              ['%.1e' % rel(ra['vec_after'][k], rb['vec_after'][k]) for k in (4, 5)])
        print(line)


if __name__ == '__main__':
    main()
