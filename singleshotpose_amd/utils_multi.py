"""Multi-object decode helpers - host-side mirror of multi_obj_pose_estimation/utils_multi.py.

Everything utils.py offers, plus get_multi_region_boxes (utils_multi.py:266-382), bbox_iou (:125-156) and `nms`
(:223-241: no caller on the pose path - valid_multi.py / train_multi.py never invoke it - kept as a small host helper
because the reference's scripts do `from utils_multi import *`).  The per-cell decode (sigmoid, grid offsets, softmax, arg-max) runs in ssp_region_decode_all; the
variable-length box lists are assembled on the host from that one device->host copy, with the reference's rules:
threshold on det_conf (only_objectness) or det_conf*cls_max_conf; a fallback box of `correspondingclass` when no kept
box has that class; `max_cls_conf` is NOT reset per image.
"""
import sys

import numpy as np
import torch

from . import _lib
from .utils import *  # noqa: F401,F403


def _span(lo_a, hi_a, lo_b, hi_b):
    """Length of the overlap of two intervals (<= 0 when they are disjoint)."""
    return min(hi_a, hi_b) - max(lo_a, lo_b)


def bbox_iou(box1, box2, x1y1x2y2=False):
    """Intersection over union of two boxes given as corners (x1, y1, x2, y2) or as centre + size (x, y, w, h) - the
    helper region_loss_multi.py:74 calls on [0, 0, anchor_w, anchor_h] boxes (utils_multi.py:125-156); on the hot path
    the anchor pick runs inside ssp_region_loss, this host version serves callers and tests."""
    if x1y1x2y2:
        ax0, ay0, ax1, ay1 = box1[0], box1[1], box1[2], box1[3]
        bx0, by0, bx1, by1 = box2[0], box2[1], box2[2], box2[3]
    else:
        ax0, ax1 = box1[0] - box1[2] / 2.0, box1[0] + box1[2] / 2.0
        ay0, ay1 = box1[1] - box1[3] / 2.0, box1[1] + box1[3] / 2.0
        bx0, bx1 = box2[0] - box2[2] / 2.0, box2[0] + box2[2] / 2.0
        by0, by1 = box2[1] - box2[3] / 2.0, box2[1] + box2[3] / 2.0
    ow, oh = _span(ax0, ax1, bx0, bx1), _span(ay0, ay1, by0, by1)
    if ow <= 0 or oh <= 0:
        return 0.0
    inter = ow * oh
    return inter / ((ax1 - ax0) * (ay1 - ay0) + (bx1 - bx0) * (by1 - by0) - inter)


def nms(boxes, nms_thresh):
    """Greedy non-maximum suppression over (x, y, w, h, det_conf, ...) boxes (utils_multi.py:223-241): boxes are visited
    by descending det_conf; a visited box with det_conf > 0 is kept and zeroes the det_conf (IN PLACE, as the reference
    does) of every later box whose centre-size IoU with it exceeds `nms_thresh`."""
    if len(boxes) == 0:
        return boxes
    order = np.argsort(np.asarray([1.0 - float(b[4]) for b in boxes], dtype=np.float32), kind='stable')
    kept = []
    for pos, i in enumerate(order):
        cur = boxes[i]
        if not cur[4] > 0:
            continue
        kept.append(cur)
        for j in order[pos + 1:]:
            if bbox_iou(cur, boxes[j], x1y1x2y2=False) > nms_thresh:
                boxes[j][4] = 0
    return kept


def region_rows(output, num_classes, num_keypoints, num_anchors):
    """(B, nA*h*w, 2K+3+nC) float32 CPU array of decoded cells in the reference's (cy, cx, anchor) scan order."""
    if output.dim() == 3:
        output = output.unsqueeze(0)
    if not output.is_cuda:
        raise RuntimeError("get_multi_region_boxes runs on the MI355X HIP kernel only: got a %s tensor (no CPU fallback)" % output.device)
    assert output.size(1) == (2 * num_keypoints + 1 + num_classes) * num_anchors
    out = output.detach().to(torch.float32).contiguous()
    B, h, w = out.size(0), out.size(2), out.size(3)
    rows = torch.empty(B, num_anchors * h * w, 2 * num_keypoints + 3 + num_classes, dtype=torch.float32, device=out.device)
    _lib.call('ssp_region_decode_all', out.data_ptr(), rows.data_ptr(), B, num_anchors, num_classes, h, w,
              num_keypoints, torch.cuda.current_stream().cuda_stream)
    return rows.cpu().numpy()


def get_multi_region_boxes(output, conf_thresh, num_classes, num_keypoints, anchors, num_anchors, correspondingclass,
                           only_objectness=1, validation=False):
    K = num_keypoints
    rows = region_rows(output, num_classes, K, num_anchors)
    all_boxes = []
    max_cls_conf = -sys.maxsize          # persists across images, as in the reference
    max_ind = None                       # (image, cell) of the running fallback candidate - also persists
    for b in range(rows.shape[0]):
        r = rows[b]
        det, cmax, cid = r[:, 2 * K], r[:, 2 * K + 1], r[:, 2 * K + 2]
        ccorr = r[:, 2 * K + 3 + correspondingclass]
        conf = det if only_objectness else det * cmax
        # running arg-max used by the fallback box: strict improvements of BOTH det_conf and the class confidence
        max_conf = -1
        for ind in range(r.shape[0]):
            if det[ind] > max_conf and ccorr[ind] > max_cls_conf:
                max_conf, max_cls_conf, max_ind = det[ind], ccorr[ind], (b, ind)
        boxes = []
        for ind in np.nonzero(conf > conf_thresh)[0]:
            box = [float(v) for v in r[ind, :2 * K]] + [float(det[ind]), float(cmax[ind]), int(cid[ind])]
            if (not only_objectness) and validation:
                for c in range(num_classes):
                    tmp = r[ind, 2 * K + 3 + c]
                    if c != int(cid[ind]) and det[ind] * tmp > conf_thresh:
                        box += [float(tmp), c]
            boxes.append(box)
        if len(boxes) == 0 or correspondingclass not in [bx[2 * K + 2] for bx in boxes]:
            if max_ind is None:
                raise UnboundLocalError("max_ind")      # the reference fails the same way when no cell ever qualified
            boxes.append([float(v) for v in rows[max_ind[0]][max_ind[1], :2 * K]] +
                         [float(max_conf), float(max_cls_conf), correspondingclass])
        all_boxes.append(boxes)
    return all_boxes
