"""Drop-in for the reference's dataset.py: train.py's `dataset.listDataset(...)` inside its own `torch.utils.data.DataLoader`
(train.py:56-65), with the per-sample Pillow augmentation (dataset.py:92-107 -> image.py:130-145) replaced by ONE batched
GPU pass (singleshotpose_amd.image.DeviceAugmenter) - the unmodified training script
reaches it without an edit.

How an unmodified DataLoader gets a batched GPU augmentation:
  * `listDataset.__getitem__` (train=True) does what has to stay on the host, in the reference's order: the multi-scale
    shape draw of the first sample of a batch (dataset.py:66-90), the background draw (:100-101), the file decodes (PIL:
    out of scope, runs in the DataLoader's workers as before), the augmentation draws (image.py:46-64, :39-44 - from the
    global `random`, so a seeded run draws what the reference draws) and the label arithmetic (image.py:78-109).  It
    returns (RawSample, label): the decoded bytes and the draws instead of the augmented tensor.
  * torch's default collate has a registry for element types (`default_collate_fn_map`); this module registers RawSample
    there, so the DataLoader's own `default_collate` - in the main process or in a worker - turns the samples of a batch
    into a RawBatch (and stacks the labels as it always did).  RawBatch has `pin_memory()`, which is what the
    DataLoader's pinning thread looks for on objects that are not tensors.
  * train.py:82-83 `data = data.cuda()` is where the reference moves the batch to the GPU; RawBatch.cuda() uploads the
    decoded bytes and runs the four augmentation launches there.  The result is the (B, H, W, 3) uint8 batch that
    Darknet.forward takes as it is (`/255` on the GPU, bit-exact with ToTensor): byte for byte the pixels the reference's
    Pillow pipeline produces for the same draws (tests/test_gpu_dropin.py runs both over a seeded epoch).
`transform` may be None or the reference's `Compose([ToTensor()])`: ToTensor's division is what Darknet.forward does with
the uint8 batch.  SSP_DATASET_FLOAT=1 makes `.cuda()` return ToTensor's own (B, 3, H, W) float32 batch instead (one
device-side conversion) for callers that do arithmetic on `data` themselves.  Any other transform is refused.
train=False (valid.py:120-123, train.py:369-374): decode, resize, labels and the caller's transform on the host, as the
reference has them - that path is one image per step and is not on the hot path.
No CPU fallback: RawBatch has no tensor behaviour before `.cuda()`.
"""
import os
import random

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset
from torch.utils.data._utils.collate import default_collate_fn_map

from singleshotpose_amd import image as _image
from utils import read_truths_args

# augmentation strengths of the training branch (dataset.py:93-97)
JITTER, HUE, SATURATION, EXPOSURE = 0.2, 0.1, 1.5, 1.5


def _u8(pil):
    return torch.from_numpy(np.array(pil.convert('RGB'), dtype=np.uint8))


class RawSample(object):
    """What one training sample is before the GPU pass: decoded image / mask / background bytes ((h, w, 3) uint8 tensors:
    they cross from a worker process in shared memory), the network input shape of its batch and its augmentation draws."""
    __slots__ = ('img', 'mask', 'bg', 'shape', 'draws')

    def __init__(self, img, mask, bg, shape, draws):
        self.img, self.mask, self.bg, self.shape, self.draws = img, mask, bg, shape, draws


class RawBatch(object):
    """The samples of one DataLoader batch; `.cuda()` is the augmentation.

    Built by the collate function - in the worker process when the DataLoader has workers - which packs the decoded bytes
    of all samples (image, mask, background each) into ONE contiguous uint8 tensor plus an offset table: the batch crosses
    the worker boundary as one shared-memory segment (one file descriptor, not 3 x batch of them: 64 samples x 10 workers x
    prefetch 2 was ~3.9 k descriptors, over the usual 1024 limit), `pin_memory()` pins one block, and `.cuda()` is one
    host-to-device copy followed by the four augmentation launches on views of the device copy."""

    def __init__(self, samples):
        shapes = set(s.shape for s in samples)
        if len(shapes) != 1:
            raise RuntimeError("samples of one batch carry different network shapes %s: the DataLoader's batch_size and "
                               "listDataset's batch_size must agree (dataset.py:66 draws the shape every batch_size "
                               "samples)" % sorted(shapes))
        self.shape = samples[0].shape
        self.draws = [s.draws for s in samples]
        self.layout, off = [], 0                  # per sample: ((offset, h, w) of image, mask, background)
        for s in samples:
            row = []
            for t in (s.img, s.mask, s.bg):
                if not (t.dtype == torch.uint8 and t.dim() == 3 and t.size(2) == 3):
                    raise ValueError("decoded images are (h, w, 3) uint8 tensors")
                row.append((off, int(t.size(0)), int(t.size(1))))
                off += (t.numel() + 15) // 16 * 16
            self.layout.append(tuple(row))
        self.blob = torch.empty(off, dtype=torch.uint8)
        for s, row in zip(samples, self.layout):
            for t, (o, h, w) in zip((s.img, s.mask, s.bg), row):
                self.blob[o:o + h * w * 3] = t.reshape(-1)

    def _views(self, blob):
        return [tuple(blob[o:o + h * w * 3].view(h, w, 3) for o, h, w in row) for row in self.layout]

    @property
    def samples(self):
        """The batch as RawSample views of the packed bytes (introspection / tests)."""
        return [RawSample(i, m, b, self.shape, d) for (i, m, b), d in zip(self._views(self.blob), self.draws)]

    def __len__(self):
        return len(self.layout)

    def size(self, dim=None):
        w, h = self.shape
        full = (len(self.layout), h, w, 3)
        return full if dim is None else full[dim]

    def pin_memory(self, device=None):
        self.blob = self.blob.pin_memory()        # one pinned block; .cuda() copies straight out of it
        return self

    def cuda(self, device=None, non_blocking=False):
        if device is None or isinstance(device, int):
            idx = torch.cuda.current_device() if device is None else device
        else:
            idx = torch.device(device).index
            idx = torch.cuda.current_device() if idx is None else idx
        dev = torch.device('cuda', idx)
        aug = _augmenter(dev)
        none = np.zeros((0, 1))      # the labels were filled sample by sample in __getitem__
        with torch.cuda.device(dev):
            views = self._views(self.blob.to(dev, non_blocking=True))      # ONE upload (asynchronous when the blob is pinned)
            out, _ = aug.load_data_detection_batch([v[0] for v in views], [v[1] for v in views], [v[2] for v in views],
                                                   [none] * len(views), self.shape, JITTER, HUE, SATURATION, EXPOSURE,
                                                   draws=self.draws)
        if os.environ.get('SSP_DATASET_FLOAT', '0') == '1':      # ToTensor's own layout and arithmetic
            return out.permute(0, 3, 1, 2).to(torch.float32).div(255).contiguous()
        return out

    def to(self, *args, **kwargs):
        dev = kwargs.get('device', args[0] if args else None)
        if dev is None or torch.device(dev).type != 'cuda':
            raise RuntimeError("a dataset.RawBatch becomes pixels on the GPU only: .cuda() / .to('cuda') (no CPU fallback)")
        return self.cuda(dev)

    def __getattr__(self, name):      # anything a tensor would answer
        if name in ('blob', 'layout', 'draws', 'shape'):      # (un-pickling looks attributes up before __init__ ran)
            raise AttributeError(name)
        raise AttributeError("dataset.RawBatch has no %r: it is the un-augmented batch - call .cuda() first "
                             "(train.py:82-83), the augmentation runs on the GPU (no CPU fallback)" % name)


_AUGMENTERS = {}


def _augmenter(dev):
    a = _AUGMENTERS.get(dev)
    if a is None:
        a = _AUGMENTERS[dev] = _image.DeviceAugmenter(dev)
    return a


def _collate_raw(batch, *, collate_fn_map=None):
    return RawBatch(batch)


default_collate_fn_map[RawSample] = _collate_raw


def _totensor_only(transform):
    """None, ToTensor() or Compose([ToTensor()]) (torchvision's or dropin/torchvision's: matched by class name)."""
    if transform is None:
        return True
    name = type(transform).__name__
    if name == 'ToTensor':
        return True
    return name == 'Compose' and all(type(t).__name__ == 'ToTensor' for t in getattr(transform, 'transforms', [None]))


def multiscale_width(seen, nbatches, batch_size, rng=random):
    """Network input width, in cells, for the batch that starts after `seen` samples (dataset.py:66-90): 13 for the first
    ten epochs, then every ten epochs the range widens by one cell at each end, down to 7..26 from epoch 70 on."""
    period = 10 * nbatches * batch_size
    stage = 7 if period == 0 else min(seen // period, 7)
    if stage == 0:
        return 13
    return rng.randint(0, 5 + 2 * stage) + 14 - stage


def label_path(imgpath):
    return imgpath.replace('images', 'labels').replace('JPEGImages', 'labels').replace('.jpg', '.txt').replace('.png', '.txt')


def mask_path(imgpath):
    return imgpath.replace('JPEGImages', 'mask').replace('/00', '/').replace('.jpg', '.png')      # image.py:132


class listDataset(Dataset):
    """Same constructor, attributes and sample order as the reference's class (dataset.py:14-50)."""

    def __init__(self, root, shape=None, shuffle=True, transform=None, target_transform=None, train=False, seen=0,
                 batch_size=64, num_workers=4, cell_size=32, bg_file_names=None, num_keypoints=9, max_num_gt=50):
        with open(root, 'r') as f:
            self.lines = f.readlines()
        if shuffle:
            random.shuffle(self.lines)
        self.nSamples = len(self.lines)
        self.transform = transform
        self.target_transform = target_transform
        self.train = train
        self.shape = shape
        self.seen = seen
        self.batch_size = batch_size
        self.num_workers = num_workers
        self.bg_file_names = bg_file_names
        self.cell_size = cell_size
        self.nbatches = self.nSamples // self.batch_size
        self.num_keypoints = num_keypoints
        self.max_num_gt = max_num_gt
        if train and not _totensor_only(transform):
            raise TypeError("dropin dataset.listDataset(train=True) takes transform=None or Compose([ToTensor()]) (what "
                            "train.py:59 passes): the batch is augmented on the GPU and Darknet.forward does ToTensor's "
                            "/255 there; got %r" % (transform,))

    def __len__(self):
        return self.nSamples

    def __getitem__(self, index):
        assert index <= len(self), 'index range error'      # (dataset.py:57 has <=, kept: index == len raises IndexError below)
        imgpath = self.lines[index].rstrip()
        if self.train and index % self.batch_size == 0:
            width = multiscale_width(self.seen, self.nbatches, self.batch_size) * self.cell_size
            self.shape = (width, width)
        if self.train:
            bgpath = self.bg_file_names[random.randint(0, len(self.bg_file_names) - 1)]
            img = _u8(Image.open(imgpath))
            mask = _u8(Image.open(mask_path(imgpath)))
            bg = _u8(Image.open(bgpath))
            oh, ow = img.shape[0], img.shape[1]
            d = _image.draw_augmentation(ow, oh, JITTER, HUE, SATURATION, EXPOSURE)
            sx = float(ow - d['pleft'] - d['pright']) / ow
            sy = float(oh - d['ptop'] - d['pbot']) / oh
            dx = (float(d['pleft']) / ow) / sx
            dy = (float(d['ptop']) / oh) / sy
            label = torch.from_numpy(_image.fill_truth_detection(label_path(imgpath), self.shape[0], self.shape[1], d['flip'],
                                                                 dx, dy, 1. / sx, 1. / sy, self.num_keypoints,
                                                                 self.max_num_gt))
            img = RawSample(img, mask, bg, (int(self.shape[0]), int(self.shape[1])), d)
        else:
            img = Image.open(imgpath).convert('RGB')
            if self.shape:
                img = img.resize(self.shape)
            num_labels = 2 * self.num_keypoints + 3
            cap = self.max_num_gt * num_labels
            label = torch.zeros(cap)
            labpath = label_path(imgpath)
            if os.path.getsize(labpath):
                tmp = torch.from_numpy(read_truths_args(labpath)).view(-1)
                if tmp.numel() > cap:
                    label = tmp[0:cap]
                elif tmp.numel() > 0:
                    label[0:tmp.numel()] = tmp
            if self.transform is not None:
                img = self.transform(img)
        if self.target_transform is not None:
            label = self.target_transform(label)
        self.seen = self.seen + self.num_workers
        return (img, label)
