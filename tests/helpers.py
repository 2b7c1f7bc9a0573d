"""Shared test helpers (CPU side): golden loading, seeded inputs, error metrics."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def rel_err(a, b):
    """max|a-b| / max|b| per tensor - the metric SURVEY.md section 8(d) prescribes (tolerance 1e-4)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / denom)


def golden_input(rec, B, H, W):
    if 'x' in rec.files:
        return rec['x']
    rs = np.random.RandomState(int(rec['x_seed'][0]))
    return rs.uniform(0, 1, (B, 3, H, W)).astype(np.float32)


def load_state_into(model, blocks, state):
    """Copy an oracle.darknet_ref.seeded_state into a (product or reference-shaped) Darknet module tree."""
    with torch.no_grad():
        for ind, e in enumerate(state):
            if e is None:
                continue
            seq = model.models[ind]
            seq[0].weight.copy_(e['weight'])
            if 'bn_weight' in e:
                seq[1].weight.copy_(e['bn_weight'])
                seq[1].bias.copy_(e['bn_bias'])
                seq[1].running_mean.copy_(e['running_mean'])
                seq[1].running_var.copy_(e['running_var'])
            else:
                seq[0].bias.copy_(e['bias'])


def clone_state(state, requires_grad=False):
    out = []
    for e in state:
        if e is None:
            out.append(None)
            continue
        d = {}
        for k, v in e.items():
            t = v.clone()
            if requires_grad and not k.startswith('running'):
                t.requires_grad_(True)
            d[k] = t
        out.append(d)
    return out


def make_targets(rs, nB, ngt, multi=False, dtype=np.float64):
    t = np.zeros((nB, 50, 21), dtype=dtype)
    for b in range(nB):
        for k in range(ngt[b]):
            t[b, k, 0] = rs.randint(0, 13) if multi else 0
            c = rs.uniform(0.2, 0.8, 2)
            t[b, k, 1:3] = c
            t[b, k, 3:19] = (c[None, :] + rs.uniform(-0.12, 0.12, (8, 2))).reshape(-1)
            t[b, k, 19:21] = rs.uniform(0.1, 0.4, 2)
    return t.reshape(nB, -1)
