#!/usr/bin/env python
"""One-off parity sweep of the FULL yolo-pose.cfg over multi-scale training resolutions (dataset.py:66-90: 224..832 in
steps of 32) against the CPU oracle: train-mode forward + backward, output and every conv / BN gradient norm.
Exercises the per-shape plans the autotuner picks (hybrid launches, deep split-K, XCD-ordered wgrad, folded taps)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import clone_state, load_state_into, rel_err
from oracle.darknet_ref import forward_ref, seeded_state
from singleshotpose_amd.darknet import Darknet

sizes = [int(s) for s in sys.argv[1].split(',')] if len(sys.argv) > 1 else [224, 352, 480, 608]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
state = seeded_state(model.blocks, 3)
load_state_into(model, model.blocks, state)
model = model.cuda().train()
worst = 0.0
for s in sizes:
    rs = np.random.RandomState(s)
    x = torch.from_numpy(rs.uniform(0, 1, (B, 3, s, s)).astype(np.float32))
    t0 = time.time()
    model.zero_grad()
    out = model(x.cuda())
    st = clone_state(state, requires_grad=True)
    y = forward_ref(model.blocks, st, x, training=True)
    probe = torch.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32))
    (out * probe.cuda()).sum().backward()
    (y * probe).sum().backward()
    e_out = rel_err(out.detach().cpu().numpy(), y.detach().numpy())
    errs = []
    for ind, e in enumerate(st):
        if e is not None:
            g = model.models[ind][0].weight.grad.cpu().numpy()
            r = e['weight'].grad.numpy()
            errs.append(abs(np.linalg.norm(g) / np.linalg.norm(r) - 1))
    worst = max(worst, e_out, max(errs))
    print('size %3d B=%d: out rel err %.2e, conv-grad norm rel err max %.2e mean %.2e  (%.1f s)' %
          (s, B, e_out, max(errs), float(np.mean(errs)), time.time() - t0))
    # BN running statistics were updated by both: reset the module's from the oracle's copy for the next size
    load_state_into(model, model.blocks, state)
print('worst', worst)
sys.exit(0 if worst < 2e-2 else 1)
