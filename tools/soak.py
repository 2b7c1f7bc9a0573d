#!/usr/bin/env python
"""Soak: N training steps; device memory and step time must not drift (allocator churn, event / stream leaks)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from singleshotpose_amd.darknet import Darknet
from singleshotpose_amd.optim import SGD
from singleshotpose_amd.region_loss import RegionLoss
from bench import synthetic_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(0)
m = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg')).cuda().train()
crit = RegionLoss(); crit.verbose = False
opt = SGD(m.parameters(), lr=1e-3 / 64, momentum=0.9, weight_decay=0.0005 * 64)
x, tgt = synthetic_batch(64, 416, 416, 1, 'cuda')
def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(m(x), tgt, 20)
    loss.backward()
    opt.step()
    return loss
for _ in range(5): step()
torch.cuda.synchronize()
a0, r0 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
marks = []
t0 = time.perf_counter()
for i in range(N):
    loss = step()
    if (i + 1) % 100 == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        marks.append((i + 1, (t1 - t0) / 100 * 1e3, float(loss)))
        t0 = t1
a1, r1 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
for n, ms, l in marks:
    print('steps %4d: %.3f ms/step, loss %.4f' % (n, ms, l))
print('allocated %.1f -> %.1f MB, reserved %.1f -> %.1f MB' % (a0 / 1e6, a1 / 1e6, r0 / 1e6, r1 / 1e6))
assert abs(a1 - a0) < 64e6 and r1 <= r0 * 1.05 + 64e6, "device memory drifted"
assert all(l == l for _, _, l in marks), "loss became NaN"
