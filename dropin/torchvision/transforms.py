"""`Compose` and `ToTensor` with torchvision's semantics for what dataset.py:113-131 feeds them (PIL RGB images).

ToTensor: (H, W, C) uint8 -> (C, H, W) float32 in [0, 1] by true division by 255 - the arithmetic
singleshotpose_amd's ssp_u8hwc_to_nhwc kernel reproduces bit for bit when the caller hands over image bytes instead.
"""
import numpy as np
import torch


class Compose(object):
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img

    def __repr__(self):
        return 'Compose(%s)' % ', '.join(repr(t) for t in self.transforms)


class ToTensor(object):
    def __call__(self, pic):
        if isinstance(pic, np.ndarray):
            arr = pic[:, :, None] if pic.ndim == 2 else pic
            t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
            return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t
        mode = getattr(pic, 'mode', None)
        if mode is None:
            raise TypeError('pic should be PIL Image or ndarray. Got %s' % type(pic))
        if mode == 'I':
            t = torch.from_numpy(np.array(pic, np.int32, copy=True))
        elif mode == 'I;16':
            t = torch.from_numpy(np.array(pic, np.int16, copy=True))
        elif mode == 'F':
            t = torch.from_numpy(np.array(pic, np.float32, copy=True))
        elif mode == '1':
            t = 255 * torch.from_numpy(np.array(pic, np.uint8, copy=True))
        else:
            t = torch.from_numpy(np.array(pic, np.uint8, copy=True))
        t = t.view(pic.size[1], pic.size[0], len(pic.getbands())).permute(2, 0, 1).contiguous()
        return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t

    def __repr__(self):
        return 'ToTensor()'
