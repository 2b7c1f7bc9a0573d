#!/usr/bin/env python
"""Golden vectors for the training-time augmentation: the reference's OWN image.py run on seeded synthetic files.

    python oracle/gen_image_golden.py        (build container only: needs /root/reference and Pillow)

Imports /root/reference/image.py unmodified.  One environment patch: Pillow >= 12 removed ImageMath.eval (image.py:125
calls it); ImageMath.unsafe_eval is the same function under its new name.  Each case writes an image, a mask, a
background and a label file into a temporary LINEMOD-shaped tree, seeds `random`, calls load_data_detection
(image.py:130-145) and stores inputs, seed and outputs in tests/golden/image_aug.npz.
"""
import os
import random
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'


def synth(rs, h, w):
    """A smooth ramp + blocks + noise image (exercises the resampler's negative lobes and the clamps)."""
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx * 3 + yy * 5) % 256)], -1).astype(np.int64)
    img += rs.randint(-40, 41, img.shape)
    img[h // 4:h // 2, w // 3:w // 2] = rs.randint(0, 256, 3)
    img[:3, :5] = 255
    img[-4:, -6:] = 0
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    from PIL import Image, ImageMath
    if not hasattr(ImageMath, 'eval'):
        ImageMath.eval = ImageMath.unsafe_eval
    sys.path.insert(0, REF)
    import image as refimage
    cases = [  # (img w, h), (bg w, h), shape, jitter, hue, saturation, exposure, seed, labels
        ((96, 72), (80, 50), (64, 64), 0.2, 0.1, 1.5, 1.5, 11, 1),
        ((96, 72), (130, 100), (96, 64), 0.3, 0.1, 1.5, 1.5, 12, 2),
        ((160, 120), (160, 120), (96, 96), 0.2, 0.05, 1.2, 1.8, 13, 1),
        ((64, 48), (20, 30), (128, 96), 0.1, 0.1, 1.5, 1.5, 14, 3),
    ]
    out = {'n': np.array(len(cases))}
    rs = np.random.RandomState(7)
    with tempfile.TemporaryDirectory(prefix='ssp_img_') as tmp:
        for d in ('JPEGImages', 'mask', 'labels', 'bg'):
            os.makedirs(os.path.join(tmp, d))
        for ci, ((w, h), (bw, bh), shape, jit, hue, sat, exp, seed, nlab) in enumerate(cases):
            img, bg = synth(rs, h, w), synth(rs, bh, bw)
            yy, xx = np.mgrid[0:h, 0:w]
            m = (((xx - w / 2) ** 2 / (w / 3) ** 2 + (yy - h / 2) ** 2 / (h / 3) ** 2) < 1).astype(np.uint8) * 255
            mask = np.stack([m, m, m], -1)
            mask[h // 2, :, 1] = 127          # boundary values of the mask threshold, per channel
            mask[h // 2 + 1, :, 2] = 128
            name = '%06d' % (ci + 3)
            ip = os.path.join(tmp, 'JPEGImages', name + '.png')
            Image.fromarray(img, 'RGB').save(ip)
            Image.fromarray(mask, 'RGB').save(os.path.join(tmp, 'mask', name[2:] + '.png'))
            bp = os.path.join(tmp, 'bg', name + '.png')
            Image.fromarray(bg, 'RGB').save(bp)
            lab = np.zeros((nlab, 21))
            for k in range(nlab):
                lab[k, 0] = k
                lab[k, 1:19] = rs.uniform(0.2, 0.8, 18)
                lab[k, 19:21] = rs.uniform(0.1, 0.4, 2)
            np.savetxt(os.path.join(tmp, 'labels', name + '.txt'), lab.reshape(nlab, -1), fmt='%.6f')
            lab_rt = np.loadtxt(os.path.join(tmp, 'labels', name + '.txt')).reshape(-1, 21)
            # composite alone (image.py:111-128)
            comp = refimage.change_background(Image.fromarray(img, 'RGB'), Image.fromarray(mask, 'RGB'), Image.fromarray(bg, 'RGB'))
            random.seed(seed)
            res, label = refimage.load_data_detection(ip, shape, jit, hue, sat, exp, bp, 9, 50)
            pre = 'c%d_' % ci
            out[pre + 'img'], out[pre + 'mask'], out[pre + 'bg'] = img, mask, bg
            out[pre + 'labels'] = lab_rt
            out[pre + 'params'] = np.array([shape[0], shape[1], jit, hue, sat, exp, seed], np.float64)
            out[pre + 'composite'] = np.asarray(comp)
            out[pre + 'out'] = np.asarray(res)
            out[pre + 'label'] = np.asarray(label)
            print('case %d: out %s label nonzero %d' % (ci, np.asarray(res).shape, int((np.asarray(label) != 0).sum())))
    path = os.path.join(ROOT, 'tests', 'golden', 'image_aug.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
