"""Multi-object decode helpers - host-side mirror of multi_obj_pose_estimation/utils_multi.py.

Everything utils.py offers, plus get_multi_region_boxes (utils_multi.py:266-382), bbox_iou (:125-156) and nms
(:223-241).  The per-cell decode (sigmoid, grid offsets, softmax, arg-max) runs in ssp_region_decode_all; the
variable-length box lists are assembled on the host from that one device->host copy, with the reference's rules:
threshold on det_conf (only_objectness) or det_conf*cls_max_conf; a fallback box of `correspondingclass` when no kept
box has that class; `max_cls_conf` is NOT reset per image.
"""
import sys

import numpy as np
import torch

from . import _lib
from .utils import *  # noqa: F401,F403


def bbox_iou(box1, box2, x1y1x2y2=False):
    if x1y1x2y2:
        mx, Mx = min(box1[0], box2[0]), max(box1[2], box2[2])
        my, My = min(box1[1], box2[1]), max(box1[3], box2[3])
        w1, h1, w2, h2 = box1[2] - box1[0], box1[3] - box1[1], box2[2] - box2[0], box2[3] - box2[1]
    else:
        mx = min(box1[0] - box1[2] / 2.0, box2[0] - box2[2] / 2.0)
        Mx = max(box1[0] + box1[2] / 2.0, box2[0] + box2[2] / 2.0)
        my = min(box1[1] - box1[3] / 2.0, box2[1] - box2[3] / 2.0)
        My = max(box1[1] + box1[3] / 2.0, box2[1] + box2[3] / 2.0)
        w1, h1, w2, h2 = box1[2], box1[3], box2[2], box2[3]
    cw, ch = w1 + w2 - (Mx - mx), h1 + h2 - (My - my)
    if cw <= 0 or ch <= 0:
        return 0.0
    carea = cw * ch
    return carea / (w1 * h1 + w2 * h2 - carea)


def nms(boxes, nms_thresh):
    if len(boxes) == 0:
        return boxes
    det_confs = torch.zeros(len(boxes))
    for i in range(len(boxes)):
        det_confs[i] = 1 - boxes[i][4]
    _, order = torch.sort(det_confs)
    out = []
    for i in range(len(boxes)):
        box_i = boxes[order[i]]
        if box_i[4] > 0:
            out.append(box_i)
            for j in range(i + 1, len(boxes)):
                box_j = boxes[order[j]]
                if bbox_iou(box_i, box_j, x1y1x2y2=False) > nms_thresh:
                    box_j[4] = 0
    return out


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
