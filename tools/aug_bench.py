#!/usr/bin/env python
"""Throughput of the GPU-side augmentation (singleshotpose_amd/image.py) at LINEMOD scale: batch of 64 samples, 640 x 480
images + masks, 500 x 375 backgrounds, 416 x 416 network shape - next to the same chain through Pillow on the host (the
calls /root/reference/image.py makes, restated; one process, one core).  Prints one JSON object."""
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pillow_chain(img, mask, bg, shape, d):
    """image.py:111-128, :46-76, :14-31 with Pillow (ImageMath.eval's a*c + b*d written as a paste through the mask)."""
    from PIL import Image
    im, mk = Image.fromarray(img, 'RGB'), Image.fromarray(mask, 'RGB')
    b = Image.fromarray(bg, 'RGB').resize(im.size).convert('RGB')
    cs = []
    for c_im, c_bg, c_mk in zip(im.split(), b.split(), mk.split()):
        cs.append(Image.composite(c_im, c_bg, c_mk.point(lambda i: 255 if i >= 128 else 0)))
    comp = Image.merge('RGB', cs)
    ow, oh = comp.size
    sw_, sh_ = ow - d['pleft'] - d['pright'], oh - d['ptop'] - d['pbot']
    cropped = comp.crop((d['pleft'], d['ptop'], d['pleft'] + sw_ - 1, d['ptop'] + sh_ - 1))
    sized = cropped.resize(shape)
    hsv = sized.convert('HSV')
    h, s, v = hsv.split()
    hue, sat, val = d['dhue'], d['dsat'], d['dexp']

    def change_hue(x):
        x += hue * 255
        if x > 255:
            x -= 255
        if x < 0:
            x += 255
        return x
    out = Image.merge('HSV', (h.point(change_hue), s.point(lambda i: i * sat), v.point(lambda i: i * val))).convert('RGB')
    return np.asarray(out)


def main():
    from singleshotpose_amd.image import DeviceAugmenter, draw_augmentation
    B, shape = 64, (416, 416)
    rs = np.random.RandomState(0)
    imgs = [rs.randint(0, 256, (480, 640, 3)).astype(np.uint8) for _ in range(B)]
    yy, xx = np.mgrid[0:480, 0:640]
    m = ((xx - 320) ** 2 + (yy - 240) ** 2 < 90 ** 2).astype(np.uint8) * 255
    masks = [np.stack([m, m, m], -1) for _ in range(B)]
    bgs = [rs.randint(0, 256, (375, 500, 3)).astype(np.uint8) for _ in range(B)]
    rows = [np.concatenate([[0], rs.uniform(0.2, 0.8, 18), [0.2, 0.3]])[None] for _ in range(B)]
    aug = DeviceAugmenter()
    draws = [draw_augmentation(640, 480, 0.2, 0.1, 1.5, 1.5, random.Random(i)) for i in range(B)]

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    res = {}
    dt = timed(lambda: aug.load_data_detection_batch(imgs, masks, bgs, rows, shape, 0.2, 0.1, 1.5, 1.5, draws=draws), 10)
    res['host_arrays_in'] = {'ms_per_batch': round(dt * 1e3, 3), 'images_per_s': round(B / dt, 1),
                             'note': 'decoded numpy arrays in: staging copy + 177 MB upload + tables + 4 launches + labels'}
    di, dm, db = [torch.from_numpy(a).cuda() for a in imgs], [torch.from_numpy(a).cuda() for a in masks], \
        [torch.from_numpy(a).cuda() for a in bgs]
    dt = timed(lambda: aug.load_data_detection_batch(di, dm, db, rows, shape, 0.2, 0.1, 1.5, 1.5, draws=draws), 20)
    res['resident_in'] = {'ms_per_batch': round(dt * 1e3, 3), 'images_per_s': round(B / dt, 1),
                          'note': 'sources already in HBM (a dataset cached on the GPU): tables + 4 launches + labels'}
    # device time of the four launches alone
    aug.time_kernels = True
    ks = []
    for _ in range(5):
        aug.load_data_detection_batch(di, dm, db, rows, shape, 0.2, 0.1, 1.5, 1.5, draws=draws)
        torch.cuda.synchronize()
        ks.append(aug.kernel_events[0].elapsed_time(aug.kernel_events[1]))
    aug.time_kernels = False
    res['four_launches_gpu_ms'] = round(float(np.median(ks)), 4)
    # algorithmic bytes per sample: bg in, tmp1 out+in, img + mask in, composite out+in (crop window ~ whole), tmp2
    # out+in, result out
    per = 375 * 500 * 3 + 2 * 375 * 640 * 3 + 2 * 480 * 640 * 3 + 2 * 480 * 640 * 3 + 2 * 480 * 416 * 3 + 416 * 416 * 3
    res['algorithmic_bytes_per_batch'] = per * B
    res['achieved_GBps'] = round(per * B / (res['four_launches_gpu_ms'] * 1e-3) / 1e9, 1)
    t0 = time.perf_counter()
    n = 8
    for i in range(n):
        pillow_chain(imgs[i], masks[i], bgs[i], shape, draws[i])
    dt = (time.perf_counter() - t0) / n
    res['pillow_one_core'] = {'ms_per_image': round(dt * 1e3, 3), 'images_per_s': round(1 / dt, 1), 'sample': '%d images' % n}
    out, _ = aug.load_data_detection_batch(imgs[:4], masks[:4], bgs[:4], rows[:4], shape, 0.2, 0.1, 1.5, 1.5, draws=draws[:4])
    res['bytes_equal_pillow'] = all(np.array_equal(out[i].cpu().numpy(), pillow_chain(imgs[i], masks[i], bgs[i], shape, draws[i]))
                                    for i in range(4))
    print(json.dumps(res))


if __name__ == '__main__':
    main()
