"""GPU-side training augmentation (singleshotpose_amd/image.py, csrc/image_aug.hip) against the oracle restatement of
/root/reference/image.py's Pillow arithmetic (oracle/image_ref.py, pinned on the CPU by tests/test_oracle_image.py) and
against the reference's own outputs (tests/golden/image_aug.npz).  Byte-exact everywhere."""
import os
import random

import numpy as np
import pytest
import torch

from helpers import GOLD

pytestmark = pytest.mark.gpu


def _triples():
    a = np.arange(256, dtype=np.uint8)
    x, y, z = np.meshgrid(a, a, a, indexing='ij')
    return np.stack([x, y, z], -1).reshape(-1, 3)


def test_colour_conversions_match_on_all_2_24_inputs():
    """Image.convert('HSV') / convert('RGB') as the kernels evaluate them (ssp_distort_u8 modes 1 and 2) over every
    possible pixel value."""
    from gpu_util import dev, stream
    from oracle import image_ref as R
    from singleshotpose_amd import _lib
    t = _triples()
    x = torch.from_numpy(t).to(dev())
    out = torch.empty_like(x)
    _lib.call('ssp_distort_u8', x.data_ptr(), out.data_ptr(), t.shape[0], None, 1, stream())
    assert np.array_equal(out.cpu().numpy(), R.rgb_to_hsv(t))
    _lib.call('ssp_distort_u8', x.data_ptr(), out.data_ptr(), t.shape[0], None, 2, stream())
    assert np.array_equal(out.cpu().numpy(), R.hsv_to_rgb(t))


def test_distort_image_matches_oracle():
    from oracle import image_ref as R
    from singleshotpose_amd import image as P
    rs = np.random.RandomState(2)
    img = rs.randint(0, 256, (120, 160, 3)).astype(np.uint8)
    for hue, sat, val in ((-0.1, 1.5, 0.7), (0.0999, 1 / 1.37, 1.21), (0.0, 1.0, 1.0)):
        got = P.distort_image(torch.from_numpy(img).cuda(), hue, sat, val).cpu().numpy()
        assert np.array_equal(got, R.distort_image(img, hue, sat, val))


def test_whole_chain_reproduces_the_reference_golden():
    """load_data_detection of /root/reference/image.py (change_background -> jitter crop -> resize -> HSV jitter, and
    the label transform), from the recorded seed, byte for byte - all four golden cases as ONE batch (different image,
    background and crop sizes per sample; the network shape is per case, so cases of one shape share a launch)."""
    from singleshotpose_amd.image import DeviceAugmenter
    g = np.load(os.path.join(GOLD, 'image_aug.npz'))
    aug = DeviceAugmenter()
    for ci in range(int(g['n'])):
        pre = 'c%d_' % ci
        sw, sh, jit, hue, sat, exp, seed = g[pre + 'params']
        out, lab = aug.load_data_detection_batch([g[pre + 'img']], [g[pre + 'mask']], [g[pre + 'bg']], [g[pre + 'labels']],
                                                 (int(sw), int(sh)), jit, hue, sat, exp, 9, 50, rng=random.Random(int(seed)))
        assert out.shape == (1, int(sh), int(sw), 3) and out.dtype == torch.uint8 and out.is_cuda
        assert np.array_equal(out[0].cpu().numpy(), g[pre + 'out']), ci
        assert lab.dtype == torch.float64 and np.array_equal(lab[0].numpy(), g[pre + 'label']), ci


def test_batch_of_mixed_sizes_matches_oracle_at_linemod_scale():
    """640 x 480 images, VOC-like backgrounds of assorted sizes, 416 x 416 network shape (dataset.py's first shape), a
    batch of 6 with independent draws; image / mask of sample 0 already resident on the GPU."""
    from oracle import image_ref as R
    from singleshotpose_amd.image import DeviceAugmenter
    rs = np.random.RandomState(5)
    B, shape = 6, (416, 416)
    imgs = [rs.randint(0, 256, (480, 640, 3)).astype(np.uint8) for _ in range(B)]
    yy, xx = np.mgrid[0:480, 0:640]
    masks = []
    for i in range(B):
        m = ((xx - 320 - 20 * i) ** 2 + (yy - 240) ** 2 < (60 + 15 * i) ** 2).astype(np.uint8) * 255
        masks.append(np.stack([m, m, m], -1))
    bgs = [rs.randint(0, 256, (h, w, 3)).astype(np.uint8) for (w, h) in ((500, 375), (333, 500), (640, 480), (1024, 768),
                                                                            (120, 90), (500, 334))]
    rows = [np.concatenate([[0], rs.uniform(0.2, 0.8, 18), [0.2, 0.3]])[None] for _ in range(B)]
    rng_a, rng_b = random.Random(9), random.Random(9)
    aug = DeviceAugmenter()
    imgs_in = [torch.from_numpy(imgs[0]).cuda()] + imgs[1:]
    masks_in = [torch.from_numpy(masks[0]).cuda()] + masks[1:]
    out, lab = aug.load_data_detection_batch(imgs_in, masks_in, bgs, rows, shape, 0.2, 0.1, 1.5, 1.5, 9, 50, rng=rng_a)
    out = out.cpu().numpy()
    for i in range(B):
        d = R.draw_augmentation(rng_b, 640, 480, 0.2, 0.1, 1.5, 1.5)
        comp = R.change_background(imgs[i], masks[i], bgs[i])
        ref, flip, dx, dy, sx, sy = R.data_augmentation(comp, shape, d)
        assert np.array_equal(out[i], ref), i
        assert np.array_equal(lab[i].numpy(), R.fill_truth_detection(rows[i], flip, dx, dy, 1. / sx, 1. / sy, 9, 50)), i
    # the same object again (buffers reused, staging events honoured): identical draws -> identical bytes
    out2, _ = aug.load_data_detection_batch(imgs, masks, bgs, rows, shape, 0.2, 0.1, 1.5, 1.5, 9, 50, rng=random.Random(9))
    assert np.array_equal(out2.cpu().numpy(), out)


def test_augmented_bytes_feed_the_network():
    """The uint8 batch goes straight into Darknet.forward (ssp_u8hwc_to_nhwc = ToTensor's arithmetic)."""
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.image import DeviceAugmenter
    rs = np.random.RandomState(1)
    B = 2
    imgs = [rs.randint(0, 256, (96, 128, 3)).astype(np.uint8) for _ in range(B)]
    masks = [np.full((96, 128, 3), 255, np.uint8) for _ in range(B)]
    bgs = [rs.randint(0, 256, (50, 70, 3)).astype(np.uint8) for _ in range(B)]
    rows = [np.concatenate([[0], rs.uniform(0.3, 0.7, 18), [0.2, 0.2]])[None] for _ in range(B)]
    out, lab = DeviceAugmenter().load_data_detection_batch(imgs, masks, bgs, rows, (96, 96), 0.2, 0.1, 1.5, 1.5, 9, 50,
                                                           rng=random.Random(3))
    torch.manual_seed(0)
    model = Darknet(os.path.join(GOLD, 'tiny-pose.cfg')).cuda().eval()
    with torch.no_grad():
        y_u8 = model(out)
        y_f = model(out.cpu().permute(0, 3, 1, 2).float().div(255).cuda())      # ToTensor's division, done on the host
    assert torch.equal(y_u8, y_f)
