#!/usr/bin/env python
"""Prints the autotuned igemm plan of every conv launch of cfg/yolo-pose.cfg at a given batch / size
(plan code = tail*100000 + tile_rows*100 + ksplit*10 + ring_slots; 0 = library heuristic; 9xxxxxx = Winograd F(2x2), 8xxxxxx = Winograd F(4x4))."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from singleshotpose_amd.darknet import Darknet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 416
m = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg')).cuda().train()
x = torch.rand(B, 3, S, S, device='cuda')
m(x).sum().backward()
plan = list(m._plans.values())[0]
for ind, cs in sorted(plan.convs.items()):
    print('layer %2d  %4dx%-4d %4d->%-4d k%d  fwd %7d  dgrad %7d  wgrad %s' % (ind, cs.H, cs.W, cs.cin, cs.cout, cs.k, cs.plan_fwd, cs.plan_dgrad,
                                                                              ('winograd F(%dx%d)' % (cs.wgrad_wino, cs.wgrad_wino)) if getattr(cs, 'wgrad_wino', 0) else 'direct'))
