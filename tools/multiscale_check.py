#!/usr/bin/env python
"""Parity sweep of the FULL yolo-pose.cfg over the multi-scale training resolutions (dataset.py:66-90: 224..832 in steps
of 32) with ONE model, as train.py visits them: every shape's training step through oracle/step_check.py (decision-frozen
oracle backward) with the bars of tests/test_gpu_fullsize.py - head / loss / running statistics / every conv launch /
every parameter gradient <= 7e-5 (the first layer's filter gradient against its float64 re-evaluation), head <= 7e-5.  Exercises the
per-shape plans the autotuner picks (hybrid launches, deep split-K, XCD-ordered wgrad, folded taps).

  python tools/multiscale_check.py [sizes, comma separated | all] [batch] [json out]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import load_state_into, make_targets  # noqa: E402
from oracle.darknet_ref import seeded_state  # noqa: E402
from oracle.step_check import check_train_step, summarize  # noqa: E402
from singleshotpose_amd import engine  # noqa: E402
from singleshotpose_amd.darknet import Darknet  # noqa: E402
from singleshotpose_amd.region_loss import RegionLoss  # noqa: E402

arg = sys.argv[1] if len(sys.argv) > 1 else '224,352,480,608,832'
sizes = list(range(224, 833, 32)) if arg == 'all' else [int(s) for s in arg.split(',')]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
out_path = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != '-' else None
exact = len(sys.argv) > 4 and sys.argv[4] == 'exact'      # also the float64 yardstick (product and fp32 oracle against it)
model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
load_state_into(model, model.blocks, seeded_state(model.blocks, 3))
model = model.cuda()
crit = RegionLoss()
rec, ok_all = [], True
for s in sizes:
    rs = np.random.RandomState(s)
    x = torch.from_numpy(rs.uniform(0, 1, (B, 3, s, s)).astype(np.float32))
    tgt = torch.from_numpy(make_targets(rs, B, [1] * B))
    t0 = time.time()
    r = check_train_step(model, crit, x, tgt, 20, exact=exact)
    worst = max(r['grad_by_param'].items(), key=lambda kv: kv[1])
    if exact:
        w64 = sorted(r['grad64_by_param'].items(), key=lambda kv: -kv[1][0])[:4]
        print('   vs float64 (product, fp32 oracle):', [(k, float('%.3g' % a), float('%.3g' % b)) for k, (a, b) in w64], flush=True)
    ok = (r['head'] < 7e-5 and all(r[k] < 1e-4 for k in ('loss', 'running', 'conv', 'grad_out')) and
          all(e < 7e-5 for n, e in r['grad_by_param'].items()))      # the bars of tests/test_gpu_fullsize.py (round 5)
    ok_all = ok_all and ok
    others = max(e for n, e in r['grad_by_param'].items() if n != '0.weight')
    hb = (model._plans[(B, s, s, 0)].head_budget or {}) if (B, s, s, 0) in model._plans else {}
    print('size %3d B=%d: %s | worst grad %s %.2e | first filter %.2e, other params %.2e | budget moved %s | %s (%.1f s)' % (
        s, B, summarize(r), worst[0], worst[1], r['grad_by_param'].get('0.weight', 0.0), others,
        [(i, a, b) for i, a, b in hb.get('moved', [])], 'ok' if ok else 'FAIL', time.time() - t0), flush=True)
    rec.append(dict(size=s, batch=B, ok=ok, worst_grad_param=worst[0],
                    tuned_plans=sum(1 for _, f, d in r['plans'] if f or d),
                    **{k: float('%.3g' % r[k]) for k in ('head', 'loss', 'running', 'conv', 'grad_out', 'grad')}))
print('tune rejections', engine.TUNE_REJECTED)
if out_path:
    with open(out_path, 'w') as f:
        json.dump(dict(what="cfg/yolo-pose.cfg train step per multi-scale resolution vs oracle/step_check.py "
                            "(decision-frozen backward), one model visiting the shapes in order",
                       bars=dict(head=7e-5, loss=1e-4, running=1e-4, conv=1e-4, grad_out=1e-4, grad=7e-5,
                                 grad_first_filter=7e-5),
                       tune_rejected=[list(map(str, t)) for t in engine.TUNE_REJECTED], shapes=rec), f, indent=1)
sys.exit(0 if ok_all else 1)
