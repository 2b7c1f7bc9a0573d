#!/usr/bin/env python
"""Builds an SSP_TUNE_CACHE json that pins a given plan set (layer, fwd code, dgrad code) for a cfg / batch / size, so
a plan set printed by a failing run can be replayed deterministically.  No GPU needed.
   python tools/make_tune_cache.py out.json "[(4, 6413, 0), (5, 0, 12813), ...]" [--cfg cfg/yolo-pose.cfg --batch 64 --size 416]"""
import argparse
import ast
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('out')
    ap.add_argument('plans')
    ap.add_argument('--cfg', default=os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--size', type=int, default=416)
    a = ap.parse_args()
    from singleshotpose_amd.cfg import layer_shapes, parse_cfg, resolve_layers
    blocks = parse_cfg(a.cfg)
    shapes = layer_shapes(blocks, a.size, a.size)
    plans = {i: (f, d) for i, f, d in ast.literal_eval(a.plans)}
    pad4 = lambda c: (c + 3) // 4 * 4
    # input activation (channels, ld, H, W) per conv layer, following engine.Plan's walk
    prev = (4, 4, a.size, a.size)
    outs = {}
    cache = {}
    for ind, b in enumerate(blocks[1:]):
        w, h, c = shapes[ind]
        t = b['type']
        if t == 'convolutional':
            cin, ldin, H, W = prev
            k = int(b['size'])
            bn = int(b['batch_normalize']) != 0
            cinp, coutp = pad4(cin), pad4(c)
            f, d = plans.get(ind, (None, None))
            if f is not None and cinp % 16 == 0 and c > 64:
                cache[json.dumps(['fwd', a.batch, H, W, cinp, c, k, ldin, coutp, bn])] = f
            if d is not None and ind > 0 and cin > 64 and coutp % 16 == 0 and cinp % 16 == 0:
                cache[json.dumps(['dgrad', a.batch, H, W, coutp, cin, k, coutp, ldin])] = d
            prev = (c, coutp, h, w)
        elif t == 'maxpool':
            prev = (prev[0], prev[1], h, w)
        elif t == 'reorg':
            prev = (c, c, h, w)
        elif t == 'route':
            ls = resolve_layers(b['layers'], ind)
            prev = outs[ls[0]] if len(ls) == 1 else (c, c, h, w)
        outs[ind] = prev
    json.dump(cache, open(a.out, 'w'), indent=0, sort_keys=True)
    print(len(cache), 'entries ->', a.out)


if __name__ == '__main__':
    main()
