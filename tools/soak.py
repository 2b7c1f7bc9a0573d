#!/usr/bin/env python
"""Soak: N training steps of the headline configuration; device memory and step time must not drift (allocator churn,
event / stream leaks, clock ramps).

  python tools/soak.py [N=400] [json out]

Every step is timed on the device (one HIP event pair per step, recorded in stream order: no host synchronisation inside
the loop) and the core clock is sampled from `rocm-smi --showclocks` every 50 steps; the record holds min / median / p99 /
max step time, the per-100-step means, the clock samples and the allocator figures before / after."""
import json
import os
import re
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_batch  # noqa: E402
from singleshotpose_amd.darknet import Darknet  # noqa: E402
from singleshotpose_amd.optim import SGD  # noqa: E402
from singleshotpose_amd.region_loss import RegionLoss  # noqa: E402


def sclk_mhz():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                             timeout=10).stdout
        m = re.search(r'sclk clock level:?\s*\d*:?\s*\(?(\d+)\s*Mhz', out, re.I)
        return int(m.group(1)) if m else None
    except Exception:
        return None


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    torch.manual_seed(0)
    m = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg')).cuda().train()
    crit = RegionLoss()
    crit.verbose = False
    opt = SGD(m.parameters(), lr=1e-3 / 64, momentum=0.9, weight_decay=0.0005 * 64)
    x, tgt = synthetic_batch(64, 416, 416, 1, 'cuda')

    def step():
        opt.zero_grad(set_to_none=True)
        loss = crit(m(x), tgt, 20)
        loss.backward()
        opt.step()
        return loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    a0, r0 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    clocks = []
    evs[0].record()
    t_wall = time.perf_counter()
    for i in range(N):
        loss = step()
        evs[i + 1].record()
        if (i + 1) % 50 == 0:
            clocks.append((i + 1, sclk_mhz()))      # a subprocess: the host runs ahead of the GPU, the queue stays full
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_wall
    a1, r1 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
    ms = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(N)])
    rec = dict(what="cfg/yolo-pose.cfg training step, batch 64, 416x416, %d consecutive steps, one HIP event pair per step" % N,
               steps=N, wall_s=round(wall, 2), images_per_s=round(64 * N / wall, 1),
               step_ms=dict(min=round(float(ms.min()), 3), median=round(float(np.median(ms)), 3),
                            p99=round(float(np.percentile(ms, 99)), 3), max=round(float(ms.max()), 3),
                            first_100_mean=round(float(ms[:100].mean()), 3), last_100_mean=round(float(ms[-100:].mean()), 3)),
               per_100_steps_ms=[round(float(ms[i:i + 100].mean()), 3) for i in range(0, N, 100)],
               sclk_mhz=clocks, final_loss=float(loss.detach()),
               memory_mb=dict(allocated_before=round(a0 / 1e6, 1), allocated_after=round(a1 / 1e6, 1),
                              reserved_before=round(r0 / 1e6, 1), reserved_after=round(r1 / 1e6, 1)))
    print(json.dumps(rec))
    if out_path:
        with open(out_path, 'w') as f:
            json.dump(rec, f, indent=1)
    assert abs(a1 - a0) < 64e6 and r1 <= r0 * 1.05 + 64e6, "device memory drifted"
    assert rec['final_loss'] == rec['final_loss'], "loss became NaN"
    assert rec['step_ms']['last_100_mean'] < 1.05 * rec['step_ms']['first_100_mean'], "step time drifted"


if __name__ == '__main__':
    main()
