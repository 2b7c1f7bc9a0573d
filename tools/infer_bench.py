#!/usr/bin/env python
"""Latency-bound pieces of the path (SURVEY.md section 8(d) config 4 and the 'report us per call' rows): eval-mode
forward at 672x672 (valid.py's test size), decode, batched PnP, batched pose errors, RegionLoss, fused SGD.
Prints one JSON object; numbers go to DESIGN.md section 3."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    from singleshotpose_amd import utils as U
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.region_loss import RegionLoss
    torch.manual_seed(0)
    dev = torch.device('cuda', 0)
    model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg')).to(dev).eval()
    res = {}
    with torch.no_grad():
        for B, iters in ((1, 50), (8, 20), (64, 5)):
            x = torch.rand(B, 3, 672, 672, device=dev)
            model(x)
            dt = timed(lambda: model(x), iters)
            res['eval_forward_672_b%d' % B] = {'ms': round(dt * 1e3, 3), 'images_per_s': round(B / dt, 1)}
        x = torch.rand(64, 3, 672, 672, device=dev)
        out = model(x)
        dt = timed(lambda: U.region_boxes_batched(out, 1, 9), 50)
        res['decode_argmax_b64_21x21'] = {'us': round(dt * 1e6, 1)}
        dt = timed(lambda: U.get_region_boxes(out, 1, 9), 10)
        res['get_region_boxes_b64_21x21_incl_host_list'] = {'us': round(dt * 1e6, 1)}
    # PnP + pose errors on synthetic poses (ape-sized box, LINEMOD intrinsics)
    rs = np.random.RandomState(0)
    n = 64
    half = np.array([0.038, 0.039, 0.046])
    corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) * half
    obj = np.concatenate((np.zeros((1, 3)), corners), axis=0)
    K = np.array([[572.4114, 0.0, 325.2611], [0.0, 573.5704, 242.0489], [0.0, 0.0, 1.0]])
    R = np.stack([np.linalg.qr(rs.standard_normal((3, 3)))[0] for _ in range(n)])
    R *= np.sign(np.linalg.det(R))[:, None, None]
    t = np.stack([np.array([[rs.uniform(-.1, .1)], [rs.uniform(-.1, .1)], [rs.uniform(.6, 1.2)]]) for _ in range(n)])
    cam = np.einsum('ij,njk->nik', K, np.einsum('nij,kj->nik', R, obj) + t)
    uv = (cam[:, :2] / cam[:, 2:3]).transpose(0, 2, 1)
    objs = np.broadcast_to(obj, (n, 9, 3))
    dt = timed(lambda: U.pnp_batched(objs, uv, K), 20)
    res['pnp_batched_64_incl_h2d_d2h'] = {'us': round(dt * 1e6, 1)}
    verts = rs.uniform(-1, 1, (5841, 3)) * half          # the ape mesh has 5841 vertices
    Rg, tg = U.pnp_batched(objs, uv, K)
    dt = timed(lambda: U.pose_errors_batched(verts.T, Rg, tg, R, t, K), 20)
    res['pose_errors_batched_64x5841_incl_h2d_d2h'] = {'us': round(dt * 1e6, 1)}
    dt = timed(lambda: U.calc_pts_diameter_gpu(verts), 10)
    res['pts_diameter_5841_incl_h2d_d2h'] = {'us': round(dt * 1e6, 1)}
    # RegionLoss alone (B=64, 13x13), labels already on the device
    crit = RegionLoss()
    crit.verbose = False
    head = torch.randn(64, 20, 13, 13, device=dev, requires_grad=True)
    tgt = torch.zeros(64, 50 * 21, dtype=torch.float64)
    tgt[:, 1:19] = torch.rand(64, 18, dtype=torch.float64) * 0.5 + 0.25
    tgt[:, 19:21] = 0.2
    tgt_dev = tgt.to(dev)
    dt = timed(lambda: crit(head, tgt_dev, 20), 50)
    res['region_loss_fwd_b64_device_labels'] = {'us': round(dt * 1e6, 1)}
    dt = timed(lambda: crit(head, tgt, 20), 50)
    res['region_loss_fwd_b64_host_f64_labels'] = {'us': round(dt * 1e6, 1)}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
