#!/usr/bin/env python
"""Multi-scale soak: the training loop of train.py after epoch 10 in miniature - ONE model, batch 64, a new resolution
every few steps drawn from dataset.py:66-90's set (224 ... 832), shapes revisited - watching device memory (plan cache
eviction, per-layer Winograd workspaces), step time per shape and the loss for NaNs.

  python tools/soak_multiscale.py [visits=24] [steps per visit=5] [json out]
"""
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_batch  # noqa: E402
from singleshotpose_amd.darknet import Darknet  # noqa: E402
from singleshotpose_amd.optim import SGD  # noqa: E402
from singleshotpose_amd.region_loss import RegionLoss  # noqa: E402


def main():
    visits = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    B = 64
    torch.manual_seed(0)
    rnd = random.Random(0)
    m = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg')).cuda().train()
    crit = RegionLoss()
    crit.verbose = False
    opt = SGD(m.parameters(), lr=1e-4 / B, momentum=0.9, weight_decay=0.0005 * B)
    batches = {}
    rec, peak = [], 0
    t_all = time.perf_counter()
    for v in range(visits):
        size = (rnd.randint(0, 19) + 7) * 32              # dataset.py:88: the widest range of the schedule
        if size not in batches:
            batches[size] = synthetic_batch(B, size, size, size, 'cuda')
        x, tgt = batches[size]
        new = (B, size, size, 0) not in m._plans
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(per):
            opt.zero_grad(set_to_none=True)
            loss = crit(m(x), tgt, 20)
            loss.backward()
            opt.step()
            if i == 0:
                torch.cuda.synchronize()
                first = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steady = (dt - first) / max(per - 1, 1)
        peak = max(peak, torch.cuda.memory_reserved())
        l = float(loss.detach())
        assert l == l, "loss became NaN at visit %d (size %d)" % (v, size)
        rec.append(dict(visit=v, size=size, new_shape=new, first_step_ms=round(first * 1e3, 1), steady_ms=round(steady * 1e3, 2),
                        images_per_s=round(B / steady, 1), plans_cached=len(m._plans),
                        reserved_gb=round(torch.cuda.memory_reserved() / 1e9, 2), loss=round(l, 3)))
        print(rec[-1], flush=True)
    out = dict(what="cfg/yolo-pose.cfg multi-scale training soak, batch 64, %d visits x %d steps" % (visits, per),
               wall_s=round(time.perf_counter() - t_all, 1), peak_reserved_gb=round(peak / 1e9, 2),
               distinct_shapes=len(batches), visits=rec)
    if out_path:
        json.dump(out, open(out_path, 'w'), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != 'visits'}))


if __name__ == '__main__':
    main()
