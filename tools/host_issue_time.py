#!/usr/bin/env python
"""Host time to QUEUE one training step (no synchronisation inside the loop): at batch 64 the GPU needs ~28 ms per step,
so the time step() takes to return is what Python + ctypes + the HIP launch calls cost for the step's ~320 launches -
the floor a small-batch step cannot get under while it is launched from Python."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_batch
from singleshotpose_amd.darknet import Darknet
from singleshotpose_amd.optim import SGD
from singleshotpose_amd.region_loss import RegionLoss

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda', 0)
torch.manual_seed(0)
m = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg')).to(dev).train()
crit = RegionLoss(); crit.verbose = False
opt = SGD(m.parameters(), lr=1e-3 / B, momentum=0.9, dampening=0, weight_decay=0.0005 * B)
x, t = synthetic_batch(B, 416, 416, 1, dev)
def step():
    opt.zero_grad(set_to_none=True)
    t0 = time.perf_counter(); out = m(x); t1 = time.perf_counter()
    loss = crit(out, t, 20); t2 = time.perf_counter()
    loss.backward(); t3 = time.perf_counter()
    opt.step(); t4 = time.perf_counter()
    return (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
for _ in range(5):
    step()
torch.cuda.synchronize()
rec = []
for _ in range(4):          # few steps: the queue must not fill up
    rec.append(step())
torch.cuda.synchronize()
r = np.array(rec) * 1e3
print('batch %d: host ms to queue forward %.2f, loss %.2f, backward %.2f, optimizer %.2f, total %.2f (median of 4)' %
      ((B,) + tuple(np.median(r, axis=0)) + (float(np.median(r.sum(axis=1))),)))
