"""Diagnostic: per-parameter gradient error of the HIP path vs the reference's fp32 and fp64 runs (golden file)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import gold, golden_input, load_state_into, rel_err
from oracle.darknet_ref import seeded_state
from singleshotpose_amd.darknet import Darknet

tag, cfg, B, H, seed = 'full_train', os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'), 2, 416, 7
g = gold('darknet_%s.npz' % tag)
model = Darknet(cfg); state = seeded_state(model.blocks, seed); load_state_into(model, model.blocks, state)
model = model.cuda().train()
x = torch.from_numpy(golden_input(g, B, H, H)).cuda()
y = model(x)
print('head: mine-vs-32 %.2e  mine-vs-64 %.2e  32-vs-64 %.2e' % (rel_err(y.detach().cpu().numpy(), g['y_train']),
      rel_err(y.detach().cpu().numpy(), g['y_train64']), rel_err(g['y_train'], g['y_train64'])))
(y * torch.from_numpy(g['probe']).cuda()).sum().backward()
for n, p in model.named_parameters():
    gr = p.grad.detach().cpu().numpy()
    n64, n32 = float(g['g64norm/' + n][0]), float(g['gnorm/' + n][0])
    got = float(np.sqrt((gr.astype(np.float64) ** 2).sum()))
    if 'g64/' + n in g.files:
        r64, r32, mine = g['g64/' + n], g['grad/' + n], gr
    else:
        r64, r32 = g['g64slice/' + n], g['gslice/' + n]
        mine = gr.reshape(-1)[:: max(1, gr.size // 512)][:512]
    print('%-28s norm: mine %+.2e ref32 %+.2e | elem: mine %.2e ref32 %.2e' % (n, got / n64 - 1, n32 / n64 - 1, rel_err(mine, r64), rel_err(r32, r64)))
