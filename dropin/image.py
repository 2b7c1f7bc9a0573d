"""Drop-in for the names of the reference's image.py that callers use, on the GPU kernels of singleshotpose_amd.image.

The batched form - what dropin/dataset.py feeds, one pass per DataLoader batch - is the product path; the functions here
keep the reference's per-sample signatures (PIL images in and out) for callers that have their own Dataset around them:
each call is the same four launches on a batch of one plus a device-to-host copy of the result.

    load_data_detection   image.py:130-145   composite + jitter crop + bicubic resize + HSV distort, and the labels
    distort_image         image.py:14-31     HSV scaling of a PIL RGB image
    random_distort_image  image.py:39-44
    rand_scale            image.py:33-37     (host: two draws from `random`)
    fill_truth_detection  image.py:78-109    (host: 21 numbers per object)
change_background (image.py:111-128) and data_augmentation (:46-76) are two stages INSIDE the fused pass (the composite
is the epilogue of the background's vertical resample, the crop is the window of the next resample's descriptor) and are
not offered on their own.  No CPU fallback: without the HIP library every function here raises.
"""
import random

import numpy as np
import torch
from PIL import Image

from singleshotpose_amd import image as _image
from singleshotpose_amd.image import fill_truth_detection, rand_scale  # noqa: F401

_AUG = {}


def _augmenter():
    dev = torch.device('cuda', torch.cuda.current_device())
    a = _AUG.get(dev)
    if a is None:
        a = _AUG[dev] = _image.DeviceAugmenter(dev)
    return a


def _rgb(pil_or_path):
    im = Image.open(pil_or_path) if isinstance(pil_or_path, str) else pil_or_path
    return np.array(im.convert('RGB'), dtype=np.uint8)


def distort_image(im, hue, sat, val):
    x = torch.from_numpy(_rgb(im)).cuda()
    return Image.fromarray(_image.distort_image(x, hue, sat, val).cpu().numpy())


def random_distort_image(im, hue, saturation, exposure):
    dhue = random.uniform(-hue, hue)
    dsat = rand_scale(saturation)
    dexp = rand_scale(exposure)
    return distort_image(im, dhue, dsat, dexp)


def load_data_detection(imgpath, shape, jitter, hue, saturation, exposure, bgpath, num_keypoints, max_num_gt):
    labpath = imgpath.replace('images', 'labels').replace('JPEGImages', 'labels').replace('.jpg', '.txt').replace('.png', '.txt')
    maskpath = imgpath.replace('JPEGImages', 'mask').replace('/00', '/').replace('.jpg', '.png')
    out, label = _augmenter().load_data_detection_batch([_rgb(imgpath)], [_rgb(maskpath)], [_rgb(bgpath)], [labpath], shape,
                                                        jitter, hue, saturation, exposure, num_keypoints, max_num_gt)
    return Image.fromarray(out[0].cpu().numpy()), label[0].numpy()
