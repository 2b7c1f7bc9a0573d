"""The augmentation oracle (oracle/image_ref.py) pinned: against Pillow itself - which is all the reference's image.py
calls - and against tests/golden/image_aug.npz, the outputs of the reference's own load_data_detection / change_background
(oracle/gen_image_golden.py).  CPU only."""
import os
import random

import numpy as np
import pytest

from helpers import GOLD

from oracle import image_ref as R

PIL = pytest.importorskip('PIL')
from PIL import Image  # noqa: E402


def _all_triples():
    a = np.arange(256, dtype=np.uint8)
    x, y, z = np.meshgrid(a, a, a, indexing='ij')
    return np.stack([x, y, z], -1).reshape(4096, 4096, 3)


def test_rgb_to_hsv_matches_pillow_on_all_2_24_inputs():
    rgb = _all_triples()
    ref = np.asarray(Image.fromarray(rgb, 'RGB').convert('HSV'))
    assert np.array_equal(R.rgb_to_hsv(rgb), ref)


def test_hsv_to_rgb_matches_pillow_on_all_2_24_inputs():
    hsv = _all_triples()
    ref = np.asarray(Image.fromarray(hsv, 'HSV').convert('RGB'))
    assert np.array_equal(R.hsv_to_rgb(hsv), ref)


@pytest.mark.parametrize("w,h,ow,oh", [(640, 480, 416, 416), (500, 375, 640, 480), (37, 29, 64, 64), (64, 64, 37, 29),
                                       (640, 480, 640, 416), (100, 80, 100, 33), (13, 7, 5, 3), (5, 3, 13, 7)])
def test_resize_matches_pillow_default_filter(w, h, ow, oh):
    rs = np.random.RandomState(w * 7 + oh)
    img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    assert np.array_equal(R.resize_bicubic(img, ow, oh), np.asarray(Image.fromarray(img, 'RGB').resize((ow, oh))))


def test_point_tables_crop_and_distort_match_pillow():
    ramp = Image.fromarray(np.arange(256, dtype=np.uint8).reshape(16, 16))
    for v in (0.5, 1.5, 0.7, 1.3, 1 / 1.27, 1.4999):
        assert np.array_equal(np.asarray(ramp.point(lambda i: i * v)).reshape(-1), R.point_lut(lambda i: i * v))
    rs = np.random.RandomState(3)
    img = rs.randint(0, 256, (48, 64, 3)).astype(np.uint8)
    for box in ((-5, -3, 70, 50), (3, 4, 30, 20), (-10, 5, 40, 60), (10, -8, 64, 48)):
        assert np.array_equal(np.asarray(Image.fromarray(img, 'RGB').crop(box)), R.crop_zero(img, *box))
    # distort_image (image.py:14-31) restated with Pillow calls, against the numpy chain
    for hue, sat, val in ((-0.1, 1.5, 0.7), (0.07, 1 / 1.3, 1.2), (0.0, 1.0, 1.0)):
        im = Image.fromarray(img, 'RGB').convert('HSV')
        cs = list(im.split())
        cs[1] = cs[1].point(lambda i: i * sat)
        cs[2] = cs[2].point(lambda i: i * val)

        def change_hue(x):
            x += hue * 255
            if x > 255:
                x -= 255
            if x < 0:
                x += 255
            return x
        cs[0] = cs[0].point(change_hue)
        ref = np.asarray(Image.merge(im.mode, tuple(cs)).convert('RGB'))
        assert np.array_equal(R.distort_image(img, hue, sat, val), ref)


def test_whole_chain_matches_the_reference_golden():
    """change_background and load_data_detection of /root/reference/image.py, byte for byte, from the recorded seed."""
    g = np.load(os.path.join(GOLD, 'image_aug.npz'))
    for ci in range(int(g['n'])):
        pre = 'c%d_' % ci
        img, mask, bg = g[pre + 'img'], g[pre + 'mask'], g[pre + 'bg']
        sw, sh, jit, hue, sat, exp, seed = g[pre + 'params']
        comp = R.change_background(img, mask, bg)
        assert np.array_equal(comp, g[pre + 'composite']), ci
        rng = random.Random(int(seed))
        d = R.draw_augmentation(rng, img.shape[1], img.shape[0], jit, hue, sat, exp)
        out, flip, dx, dy, sx, sy = R.data_augmentation(comp, (int(sw), int(sh)), d)
        assert np.array_equal(out, g[pre + 'out']), ci
        label = R.fill_truth_detection(g[pre + 'labels'], flip, dx, dy, 1. / sx, 1. / sy, 9, 50)
        assert np.array_equal(label, g[pre + 'label']), ci


def test_product_host_tables_match_the_oracle():
    """Host side of singleshotpose_amd/image.py (no GPU needed): the batched coefficient tables, the distort tables and
    the random-draw order against the oracle restatement (which the tests above pin to Pillow and to the reference)."""
    from singleshotpose_amd import image as P
    for out in (416, 64, 37, 640, 5):
        ins = [640, 480, 500, 375, 37, 29, 64, 13, 7, 3, 833, out]
        ks, bnd, kk = P.resample_coeffs(ins, out)
        for j, n in enumerate(ins):
            k1, b1, c1 = R.resample_coeffs(n, out)
            assert k1 <= ks and np.array_equal(bnd[j], b1), (n, out)
            assert np.array_equal(kk[j][:, :k1], c1) and not kk[j][:, k1:].any(), (n, out)
    for hue, sat, val in ((-0.1, 1.5, 0.7), (0.0999, 1 / 1.37, 1.21), (0.0, 1.0, 1.0)):
        lh, ls, lv = R.distort_luts(hue, sat, val)
        assert np.array_equal(P.distort_tables(hue, sat, val), np.concatenate([lh, ls, lv]))
    a, b = random.Random(5), random.Random(5)
    for _ in range(20):
        assert P.draw_augmentation(640, 480, 0.2, 0.1, 1.5, 1.5, a) == R.draw_augmentation(b, 640, 480, 0.2, 0.1, 1.5, 1.5)
    g = np.load(os.path.join(GOLD, 'image_aug.npz'))
    rows = g['c1_labels']
    assert np.array_equal(P.fill_truth_detection(rows, 96, 64, 0, 0.05, -0.02, 1.1, 0.9, 9, 50),
                          R.fill_truth_detection(rows, 0, 0.05, -0.02, 1.1, 0.9, 9, 50))
