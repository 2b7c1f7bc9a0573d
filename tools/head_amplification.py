#!/usr/bin/env python
"""How much of a rounding error in ONE conv block's output reaches the network's head - a CPU experiment (float64, no GPU).

  python tools/head_amplification.py [batch]

yolo-pose.cfg, torch-default initial weights (bench.py's model), training-mode BatchNorm, the bench's synthetic batch.  For
every conv block: white noise of 1e-6 of the raw output's range is added to that block's raw conv output and the forward is
re-run; printed is (change of the head) / (head range) / 1e-6, max-norm and rms.  BatchNorm re-normalises every block, so
a perturbation is amplified layer after layer: ~400 x from the first block, ~170 x from layer 4, ~100 x from layer 8,
~20 x from the 26 x 26 layers, ~6 x from the 13 x 13 layers.  This is the weight a layer's kernel error carries in the
network-level rounding budget of the forward plans (engine.Plan._apply_head_budget measures the product of the two - the
head deviation a candidate plan causes - directly on the GPU, on the live batch).
The same script prints the oracle's own noise: PyTorch-CPU float32 against float64 at the head (2.3e-5 of the range).
"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np
from oracle.darknet_ref import forward_ref
from singleshotpose_amd.darknet import Darknet
from oracle.step_check import snapshot_state, _clone, _rel
from bench import synthetic_batch
torch.manual_seed(0)
model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
x, tgt = synthetic_batch(B, 416, 416, 1000, 'cpu')
st = snapshot_state(model)
st64 = [None if e is None else {k: v.double() for k, v in e.items()} for e in st]
eps = 1e-6
with torch.no_grad():
    y32 = forward_ref(model.blocks, _clone(st), x, training=True)
    own = {}
    y64 = forward_ref(model.blocks, st64, x.double(), training=True, raws=own)
    rng = float(y64.abs().max())
    print('oracle (PyTorch-CPU float32) against float64 at the head: %.2e of the range' % _rel(y32, y64))
    g = torch.Generator().manual_seed(5)
    for l in sorted(own):
        r = own[l]
        noise = torch.randn(r.shape, generator=g, dtype=torch.float64) * eps * float(r.abs().max())
        y = forward_ref(model.blocks, st64, x.double(), training=True, raw_override={l: r + noise})
        d = (y - y64)
        print('layer %2d  raw %s  amplification: max-norm %.1f  rms %.1f' % (l, tuple(r.shape[1:]), float(d.abs().max()) / rng / eps, float(d.pow(2).mean().sqrt()) / rng / eps), flush=True)
