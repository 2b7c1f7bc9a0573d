#!/usr/bin/env python
"""Multi-step training parity probe: the product (HIP, torch.optim.SGD as train.py:388 builds it) against the CPU
oracle trained with the same optimizer on the same batches; prints per-step losses and the worst parameter difference.
Usage: python tools/train_parity.py [cfg] [steps] [B] [size]   (test tooling: imports oracle/)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    from helpers import load_state_into, make_targets
    from oracle.darknet_ref import forward_ref, seeded_state
    from oracle.region_loss_ref import region_loss_ref
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.region_loss import RegionLoss
    cfg = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'tests', 'golden', 'tiny-pose.cfg')
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    size = int(sys.argv[4]) if len(sys.argv) > 4 else 96
    epoch = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    model = Darknet(cfg)
    state = seeded_state(model.blocks, 3)
    load_state_into(model, model.blocks, state)
    model = model.cuda().train()
    crit = RegionLoss()
    crit.verbose = False
    kw = dict(lr=1e-4 / B, momentum=0.9, dampening=0, weight_decay=0.0005 * B)
    opt = torch.optim.SGD(model.parameters(), **kw)
    names, cpu_params = [], []
    for ind, e in enumerate(state):
        if e is None:
            continue
        for k in ('weight', 'bias', 'bn_weight', 'bn_bias'):       # module order: conv.weight, conv.bias | bn.weight, bn.bias
            if k in e:
                e[k].requires_grad_(True)
                names.append('%d.%s' % (ind, k))
                cpu_params.append(e[k])
    opt_c = torch.optim.SGD(cpu_params, **kw)
    gpu_named = dict(model.named_parameters())
    rs = np.random.RandomState(0)
    for step in range(steps):
        x = torch.from_numpy(rs.uniform(0, 1, (B, 3, size, size)).astype(np.float32))
        tgt = torch.from_numpy(make_targets(rs, B, [1] * B))
        opt.zero_grad()
        loss = crit(model(x.cuda()), tgt, epoch)
        loss.backward()
        opt.step()
        opt_c.zero_grad()
        y = forward_ref(model.blocks, state, x, training=True)
        r = region_loss_ref(y.detach(), tgt, epoch)
        y.backward(r['grad'])
        gerr = 0.0
        for (n, p), q in zip(model.named_parameters(), cpu_params):
            gerr = max(gerr, float((p.grad.cpu() - q.grad).abs().max() / max(float(q.grad.abs().max()), 1e-30)))
        opt_c.step()
        perr, worst = 0.0, None
        for (n, p), q, qn in zip(model.named_parameters(), cpu_params, names):
            d = float((p.detach().cpu() - q.detach()).abs().max())
            if d > perr:
                perr, worst = d, (n, qn)
        print('step %d: loss gpu %.6f cpu %.6f rel %.2e | worst grad err %.2e | worst param abs diff %.3e %s' % (
            step, float(loss), r['loss'], abs(float(loss) - r['loss']) / abs(r['loss']), gerr, perr, worst))


if __name__ == '__main__':
    main()
