"""RegionLoss on one fused HIP kernel - host-side mirror of the reference's region_loss.py.

Same surface as /root/reference/region_loss.py:80-175: RegionLoss(num_keypoints=9, num_classes=1, anchors=[],
num_anchors=1, pretrain_num_epochs=15) with mutable attributes coord_scale / noobject_scale / object_scale /
class_scale / thresh / seen / pretrain_num_epochs (and anchors / anchor_step / iter poked from outside,
darknet.py:231-243, train.py:341); forward(output, target, epoch) -> 0-dim loss tensor supporting .backward()
and .data, and printing the reference's per-iteration status line (region_loss.py:173).

The reference moves every prediction to the host, builds targets in Python loops (build_targets,
region_loss.py:9-78) and uploads 23 mask/target tensors.  Here ssp_region_loss does decode + targets + masks +
loss + dL/d(output) in one launch per call; the only host traffic is the (<= 269 KB) label upload and - when
`verbose` - one 32-byte read of the scalars for the status line.
"""
import os
import time

import numpy as np
import torch
import torch.nn as nn

from . import _lib


class _RegionLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, mod, target, epoch):
        out = output.detach()
        if not out.is_cuda:
            raise RuntimeError("RegionLoss runs on the MI355X HIP kernel only: got a %s tensor (no CPU fallback)" % out.device)
        out = out.to(torch.float32).contiguous()
        nB, nH, nW = out.size(0), out.size(2), out.size(3)
        nA, nC, K = mod.num_anchors, mod.num_classes, mod.num_keypoints
        if out.size(1) != nA * (2 * K + 1 + nC):
            raise ValueError("output has %d channels, expected num_anchors*(2*num_keypoints+1+num_classes) = %d"
                             % (out.size(1), nA * (2 * K + 1 + nC)))
        tgt = target.detach()
        if tgt.dtype not in (torch.float32, torch.float64):
            tgt = tgt.to(torch.float32)
        if not tgt.is_cuda:
            tgt = mod._upload(tgt.contiguous(), out.device)
        tgt = tgt.contiguous().view(nB, -1)
        if tgt.size(1) != 50 * (2 * K + 3):
            raise ValueError("target must hold 50 x (2*num_keypoints+3) numbers per image, got %d" % tgt.size(1))
        grad = torch.empty_like(out)
        partials = torch.empty(nB * 8, dtype=torch.float32, device=out.device)
        stats = torch.empty(8, dtype=torch.float32, device=out.device)
        anchors = None
        if mod._multi:
            # uploaded once per anchor set: a torch.tensor(list, device=...) per call is a pageable host->device copy, which
            # blocks the host until the work queued before it has drained
            akey = (tuple(float(a) for a in mod.anchors), str(out.device))
            cache = mod.__dict__.setdefault('_anchor_cache', {})
            anchors = cache.get(akey)
            if anchors is None:
                cache.clear()
                anchors = cache[akey] = torch.tensor(akey[0], dtype=torch.float32, device=out.device)
            step = len(mod.anchors) // nA
        else:
            step = 0
        conf_on = 1 if epoch > mod.pretrain_num_epochs else 0
        st = torch.cuda.current_stream().cuda_stream
        _lib.call('ssp_region_loss', out.data_ptr(), tgt.data_ptr(), 1 if tgt.dtype == torch.float64 else 0,
                  grad.data_ptr(), partials.data_ptr(), stats.data_ptr(), nB, nA, nC, nH, nW, K,
                  float(mod.noobject_scale), float(mod.object_scale), float(mod.coord_scale), float(mod.class_scale),
                  float(mod.thresh), conf_on, 1 if mod._multi else 0,
                  anchors.data_ptr() if anchors is not None else None, step, st)
        for slot in mod.__dict__.get('_pin_ring', {}).values():
            for ev, stream in slot.pop('release', ()):
                ev.record(stream)           # this slot's pinned and device label buffers are free once the kernel has run
        ctx.save_for_backward(grad)
        mod._last_stats = stats
        return stats[4].clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


class _RegionLossBase(nn.Module):
    _multi = False

    def __init__(self, num_keypoints, num_classes, anchors, num_anchors, pretrain_num_epochs):
        super(_RegionLossBase, self).__init__()
        self.num_classes = num_classes
        self.anchors = anchors
        self.num_anchors = num_anchors
        self.anchor_step = len(anchors) // num_anchors if num_anchors else 0
        self.num_keypoints = num_keypoints
        self.coord_scale = 1
        self.noobject_scale = 1
        self.object_scale = 5
        self.class_scale = 1
        self.thresh = 0.6
        self.seen = 0
        self.pretrain_num_epochs = pretrain_num_epochs
        self.verbose = True      # print the reference's status line (one 32-byte device read per call)
        # host labels reach the kernel through a pinned ring: 'copy' = asynchronous copy into a device twin, 'mapped' = the
        # kernel reads the pinned buffer in place (zero copy)
        self.label_upload = os.environ.get('SSP_LABEL_UPLOAD', 'copy')
        self._last_stats = None

    def _upload(self, host_tensor, device):
        """Host labels -> device without stalling the host: a `.to(device)` from pageable memory blocks the Python
        thread until the forward pass queued before it has drained (the reference pays exactly that, train.py:83-97).
        The labels are staged through a ring of 4 pinned host buffers, each paired with its own device buffer and event,
        all created once per (shape, dtype, device): a call is one host memcpy into the pinned slot, one asynchronous
        copy in stream order, one event record - no allocation (host, device or event) after the first lap of the ring.
        `upload_host_us` keeps the host time of the last calls (tools/label_upload_probe.py, tests/test_gpu_head.py)."""
        t0 = time.perf_counter()
        key = (tuple(host_tensor.shape), host_tensor.dtype, str(device))
        ring = self.__dict__.setdefault('_pin_ring', {})
        slot = ring.get(key)
        if slot is None:
            n = 4
            slot = {'pin': [torch.empty(host_tensor.shape, dtype=host_tensor.dtype).pin_memory() for _ in range(n)],
                    'dev': [torch.empty(host_tensor.shape, dtype=host_tensor.dtype, device=device) for _ in range(n)],
                    'events': [torch.cuda.Event() for _ in range(n)], 'used': [False] * n, 'next': 0}
            ring[key] = slot
        i = slot['next']
        slot['next'] = (i + 1) % len(slot['pin'])
        stream = torch.cuda.current_stream(device)
        if slot['used'][i]:
            # the copy that last read this pinned buffer (and the kernel that last read its device twin) were queued 4
            # calls ago: normally long finished, so this returns at once; a caller hopping streams is ordered by the wait
            slot['events'][i].synchronize()
        t1 = time.perf_counter()
        # numpy's memcpy, not Tensor.copy_: ATen splits a 537 KB host-to-host copy over its intra-op thread pool, and on a
        # 256-thread host one call in ~60-100 then waits 80-95 ms for a parked worker (tools/label_upload_probe.py,
        # profiles/r03_label_upload.json: the whole stall sits in this one statement) - one thread copies it in ~30 us
        np_pin = slot.get('pin_np')
        if np_pin is None:
            np_pin = slot['pin_np'] = [t.numpy() for t in slot['pin']]
        if getattr(self, '_probe_aten_staging', False):      # tools/label_upload_probe.py: the round-2 form, for the record
            slot['pin'][i].copy_(host_tensor)
        else:
            np.copyto(np_pin[i], host_tensor.numpy())
        t2 = time.perf_counter()
        if self.label_upload == 'mapped':
            # pinned host memory is mapped into the GPU's address space: the kernel reads the labels (<= 269 KB, one pass)
            # straight over PCIe - no copy engine, no asynchronous memcpy call on the host
            out = slot['pin'][i]
        else:
            slot['dev'][i].copy_(slot['pin'][i], non_blocking=True)
            out = slot['dev'][i]
        if not getattr(self, '_probe_no_events', False):      # (tools/label_upload_probe.py's diagnostic mode drops the events)
            slot['used'][i] = True
            # recorded now - behind the copy that reads the pinned buffer - and AGAIN by forward() behind the kernel that
            # reads the device twin (a re-record moves the event later).  Recording here as well keeps the slot's event valid
            # when no kernel follows this upload (an exception between the two, a second upload of the same shape): the
            # reuse wait four calls later then still orders itself after the copy instead of returning on a stale event.
            slot['events'][i].record(stream)
            slot.setdefault('release', []).append((slot['events'][i], stream))
        t3 = time.perf_counter()
        hist = self.__dict__.setdefault('upload_host_us', [])
        hist.append(((t3 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))   # total, ring wait, host copy, H2D issue
        if len(hist) > 4096:
            del hist[:2048]
        return out

    def last_stats(self):
        """Device tensor [loss_x, loss_y, loss_conf, loss_cls, total, nGT, nCorrect, nProposals] of the last call."""
        return self._last_stats

    def forward(self, output, target, epoch):
        loss = _RegionLossFn.apply(output, self, target, epoch)
        if self.verbose:
            s = self._last_stats.tolist()
            if self._multi:
                print('%d: nGT %d, recall %d, proposals %d, loss: x %f, y %f, conf %f, cls %f, total %f' %
                      (self.seen, int(s[5]), int(s[6]), int(s[7]), s[0], s[1], s[2], s[3], s[4]))
            else:
                print('%d: nGT %d, recall %d, proposals %d, loss: x %f, y %f, conf %f, total %f' %
                      (self.seen, int(s[5]), int(s[6]), int(s[7]), s[0], s[1], s[2], s[4]))
        return loss


class RegionLoss(_RegionLossBase):
    """Single-object loss (region_loss.py:80-175): one trivial anchor, no class term."""
    _multi = False

    def __init__(self, num_keypoints=9, num_classes=1, anchors=[], num_anchors=1, pretrain_num_epochs=15):
        super(RegionLoss, self).__init__(num_keypoints, num_classes, anchors, num_anchors, pretrain_num_epochs)


class RegionLossMulti(_RegionLossBase):
    """Multi-object loss (multi_obj_pose_estimation/region_loss_multi.py:94-189): anchor pick by IoU, class CE."""
    _multi = True

    def __init__(self, num_keypoints=9, num_classes=13, anchors=[], num_anchors=5, pretrain_num_epochs=15):
        super(RegionLossMulti, self).__init__(num_keypoints, num_classes, anchors, num_anchors, pretrain_num_epochs)
