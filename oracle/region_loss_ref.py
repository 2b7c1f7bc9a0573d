"""CPU restatement of RegionLoss / build_targets / get_region_boxes - TEST INFRASTRUCTURE ONLY.

Follows /root/reference/region_loss.py:9-78 (build_targets), :95-161 (loss), /root/reference/utils.py:138-187
(corner_confidences / corner_confidence), :216-296 (get_region_boxes) and the multi-object twins
/root/reference/multi_obj_pose_estimation/region_loss_multi.py:9-92,110-176, utils_multi.py:125-156 (bbox_iou).
float32 torch-CPU arithmetic in the reference's operation order; gradients by CPU autograd of the same masked-SSE
expression.  Pinned by tests/golden/region_*.npz, produced by oracle/gen_golden.py from the reference code itself.

Where the single-object reference cannot run (0 or >= 2 labels per image make region_loss.py:39-40 raise on a shape
mismatch) this restatement uses the multi-object file's well-defined form of the same lines (max over GTs).
"""
import math

import numpy as np
import torch


def corner_confidences_ref(gt, pr):
    """gt, pr: (2K, n) float32 -> (n,) (utils.py:138-165); normaliser exp(2)-1."""
    n = gt.shape[1]
    dist = (gt - pr).t().contiguous().view(n, -1, 2).clone()
    dist[:, :, 0] = dist[:, :, 0] * 640
    dist[:, :, 1] = dist[:, :, 1] * 480
    d = torch.sqrt(torch.sum(dist ** 2, dim=2))
    th = torch.full_like(d, 80.0)
    mask = (d < th).float()
    conf = torch.exp(2 * (1 - d / th)) - 1
    conf0 = torch.exp(2 * (1 - torch.zeros(n, 1))) - 1
    conf = mask * (conf / conf0)
    return torch.mean(conf, dim=1)


def corner_confidence_ref(gt, pr):
    """gt, pr: (2K,) float32 -> scalar (utils.py:167-187); normaliser exp(2)-1+1e-5."""
    dist = (gt - pr).view(-1, 2).clone()
    dist[:, 0] = dist[:, 0] * 640
    dist[:, 1] = dist[:, 1] * 480
    d = torch.sqrt(torch.sum(dist ** 2, dim=1))
    mask = (d < 80).float()
    conf = torch.exp(2 * (1.0 - d / 80)) - 1
    conf0 = torch.exp(torch.FloatTensor([2])) - 1 + 1e-5
    return torch.mean(mask * (conf / conf0))


def bbox_iou_centered(aw, ah, gw, gh):
    """utils_multi.py:125-156 with x1y1x2y2=False on boxes [0,0,w,h]."""
    uw = max(aw / 2.0, gw / 2.0) - min(-aw / 2.0, -gw / 2.0)
    uh = max(ah / 2.0, gh / 2.0) - min(-ah / 2.0, -gh / 2.0)
    cw, ch = aw + gw - uw, ah + gh - uh
    if cw <= 0 or ch <= 0:
        return 0.0
    carea = cw * ch
    return carea / (aw * ah + gw * gh - carea)


def region_loss_ref(output, target, epoch, num_keypoints=9, num_classes=1, num_anchors=1, anchors=(),
                    coord_scale=1, noobject_scale=1, object_scale=5, class_scale=1, thresh=0.6,
                    pretrain_num_epochs=15, multi=False):
    """output (nB, nA*(2K+1+nC), nH, nW) float32 CPU; target (nB, 50*(2K+3)) float32/float64 CPU.

    Returns dict: loss, loss_x, loss_y, loss_conf, loss_cls, nGT, nCorrect, nProposals (python numbers) and grad
    (dL/d output, same shape).
    """
    K, nA, nC = num_keypoints, num_anchors, num_classes
    out = output.detach().clone().float().requires_grad_(True)
    nB, nH, nW = out.size(0), out.size(2), out.size(3)
    o = out.view(nB, nA, 2 * K + 1 + nC, nH, nW)
    x = [torch.sigmoid(o[:, :, 0])] + [o[:, :, 2 * i] for i in range(1, K)]
    y = [torch.sigmoid(o[:, :, 1])] + [o[:, :, 2 * i + 1] for i in range(1, K)]
    conf = torch.sigmoid(o[:, :, 2 * K])
    cls = o[:, :, 2 * K + 1:2 * K + 1 + nC]                       # nB nA nC nH nW

    grid_x = torch.linspace(0, nW - 1, nW).repeat(nH, 1).repeat(nB * nA, 1, 1).view(nB, nA, nH, nW)
    grid_y = torch.linspace(0, nH - 1, nH).repeat(nW, 1).t().repeat(nB * nA, 1, 1).view(nB, nA, nH, nW)
    pred = torch.zeros(2 * K, nB * nA * nH * nW)
    for i in range(K):
        pred[2 * i] = ((x[i].detach() + grid_x) / nW).reshape(-1)
        pred[2 * i + 1] = ((y[i].detach() + grid_y) / nH).reshape(-1)
    pred = pred.t().contiguous()                                   # (cells, 2K), cell = ((b*nA+a)*nH+j)*nW+i

    NL = 2 * K + 3
    nAnch, nPix = nA * nH * nW, nH * nW
    conf_mask = torch.ones(nB, nA, nH, nW) * noobject_scale
    coord_mask = torch.zeros(nB, nA, nH, nW)
    cls_mask = torch.zeros(nB, nA, nH, nW)
    tx = [torch.zeros(nB, nA, nH, nW) for _ in range(K)]
    ty = [torch.zeros(nB, nA, nH, nW) for _ in range(K)]
    tconf = torch.zeros(nB, nA, nH, nW)
    tcls = torch.zeros(nB, nA, nH, nW)
    tgt = target.detach()

    for b in range(nB):
        cur_pred = pred[b * nAnch:(b + 1) * nAnch].t()
        cur = torch.zeros(nAnch)
        for t in range(50):
            if tgt[b][t * NL + 1] == 0:
                break
            g = torch.FloatTensor([float(tgt[b][t * NL + 1 + k]) for k in range(2 * K)])
            cur = torch.max(cur, corner_confidences_ref(cur_pred, g.repeat(nAnch, 1).t()))
        conf_mask[b][cur.view(nA, nH, nW) > thresh] = 0

    nGT = nCorrect = 0
    for b in range(nB):
        for t in range(50):
            if tgt[b][t * NL + 1] == 0:
                break
            nGT += 1
            gx = [tgt[b][t * NL + 2 * i + 1] * nW for i in range(K)]
            gy = [tgt[b][t * NL + 2 * i + 2] * nH for i in range(K)]
            gi0, gj0 = int(gx[0]), int(gy[0])
            gt_box = torch.FloatTensor([float(tgt[b][t * NL + 1 + k]) for k in range(2 * K)])
            if multi:
                # region_loss_multi.py:51,63: best_n is still -1 here -> Python negative indexing
                pred_box = pred[b * nAnch + (-1) * nPix + gj0 * nW + gi0]
                gw, gh = float(tgt[b][t * NL + NL - 2]) * nW, float(tgt[b][t * NL + NL - 1]) * nH
                step = len(anchors) // nA
                best_iou, best_n = 0.0, -1
                for n in range(nA):
                    iou = bbox_iou_centered(anchors[step * n], anchors[step * n + 1], gw, gh)
                    if iou > best_iou:
                        best_iou, best_n = iou, n
            else:
                best_n = 0
                pred_box = pred[b * nAnch + best_n * nPix + gj0 * nW + gi0]
            c = corner_confidence_ref(gt_box, pred_box)
            coord_mask[b][best_n][gj0][gi0] = 1
            cls_mask[b][best_n][gj0][gi0] = 1
            conf_mask[b][best_n][gj0][gi0] = object_scale
            for i in range(K):
                tx[i][b][best_n][gj0][gi0] = gx[i] - gi0
                ty[i][b][best_n][gj0][gi0] = gy[i] - gj0
            tconf[b][best_n][gj0][gi0] = c
            tcls[b][best_n][gj0][gi0] = tgt[b][t * NL]
            if c > 0.5:
                nCorrect += 1

    nProposals = int((conf > 0.25).sum().item())
    cm = conf_mask.sqrt()
    sse = lambda a, bb: ((a - bb) ** 2).sum()
    loss_x = sum(coord_scale * sse(x[i] * coord_mask, tx[i] * coord_mask) / 2.0 for i in range(K))
    loss_y = sum(coord_scale * sse(y[i] * coord_mask, ty[i] * coord_mask) / 2.0 for i in range(K))
    loss_conf = sse(conf * cm, tconf * cm) / 2.0
    loss_cls = torch.zeros(())
    if multi:
        m = cls_mask == 1
        logits = cls.permute(0, 1, 3, 4, 2)[m]                     # (n, nC)
        if logits.numel():
            loss_cls = class_scale * torch.nn.functional.cross_entropy(logits, tcls[m].long(), reduction='sum')
    loss = loss_x + loss_y + (loss_cls if multi else 0)
    if epoch > pretrain_num_epochs:
        loss = loss + loss_conf
    loss.backward()
    return dict(loss=float(loss), loss_x=float(loss_x), loss_y=float(loss_y), loss_conf=float(loss_conf),
                loss_cls=float(loss_cls), nGT=nGT, nCorrect=nCorrect, nProposals=nProposals,
                grad=out.grad.detach().clone())


def get_region_boxes_ref(output, num_classes, num_keypoints, only_objectness=1):
    """utils.py:216-296 restated: the single best cell of the batch, first maximum in (b, cy, cx) order.

    Returns a list of 2K+3 python floats (coords, det_conf, cls_max_conf, cls_max_id).
    """
    K = num_keypoints
    if output.dim() == 3:
        output = output.unsqueeze(0)
    B, h, w = output.size(0), output.size(2), output.size(3)
    o = output.float().view(B, 2 * K + 1 + num_classes, h * w).transpose(0, 1).contiguous().view(2 * K + 1 + num_classes, B * h * w)
    gx = torch.linspace(0, w - 1, w).repeat(h, 1).repeat(B, 1, 1).view(B * h * w)
    gy = torch.linspace(0, h - 1, h).repeat(w, 1).t().repeat(B, 1, 1).view(B * h * w)
    xs = [torch.sigmoid(o[0]) + gx] + [o[2 * j] + gx for j in range(1, K)]
    ys = [torch.sigmoid(o[1]) + gy] + [o[2 * j + 1] + gy for j in range(1, K)]
    det = torch.sigmoid(o[2 * K])
    cls_conf = torch.softmax(o[2 * K + 1:2 * K + 1 + num_classes].transpose(0, 1), dim=1)
    cmax, cid = torch.max(cls_conf, 1)
    conf = det if only_objectness else det * cmax
    best, ind = -float('inf'), 0
    for i in range(B * h * w):
        if conf[i] > best:
            best, ind = float(conf[i]), i
    box = []
    for j in range(K):
        box.append(float(xs[j][ind] / w))
        box.append(float(ys[j][ind] / h))
    return box + [float(det[ind]), float(cmax[ind]), int(cid[ind])]


def get_multi_region_boxes_ref(output, conf_thresh, num_classes, num_keypoints, num_anchors, correspondingclass,
                               only_objectness=1):
    """utils_multi.py:266-382 restated (validation=False): per image, every (cy, cx, anchor) whose confidence exceeds
    the threshold, plus a fallback box of `correspondingclass` when none of the kept boxes has that class.
    max_cls_conf and max_ind persist across images (utils_multi.py:281,316-330)."""
    K, nA, nC = num_keypoints, num_anchors, num_classes
    if output.dim() == 3:
        output = output.unsqueeze(0)
    B, h, w = output.size(0), output.size(2), output.size(3)
    o = output.float().view(B * nA, 2 * K + 1 + nC, h * w).transpose(0, 1).contiguous().view(2 * K + 1 + nC, B * nA * h * w)
    gx = torch.linspace(0, w - 1, w).repeat(h, 1).repeat(B * nA, 1, 1).view(-1)
    gy = torch.linspace(0, h - 1, h).repeat(w, 1).t().repeat(B * nA, 1, 1).view(-1)
    xs = [torch.sigmoid(o[0]) + gx] + [o[2 * j] + gx for j in range(1, K)]
    ys = [torch.sigmoid(o[1]) + gy] + [o[2 * j + 1] + gy for j in range(1, K)]
    det = torch.sigmoid(o[2 * K])
    cls_conf = torch.softmax(o[2 * K + 1:2 * K + 1 + nC].transpose(0, 1), dim=1)
    cmax, cid = torch.max(cls_conf, 1)
    sz_hw, sz_hwa = h * w, h * w * nA
    max_cls_conf, max_ind = -float('inf'), None
    all_boxes = []

    def make(ind, det_conf, cls_max_conf, cls_max_id):
        box = []
        for j in range(K):
            box.append(float(xs[j][ind] / w))
            box.append(float(ys[j][ind] / h))
        return box + [float(det_conf), float(cls_max_conf), int(cls_max_id)]

    for b in range(B):
        boxes, max_conf = [], -1
        for cy in range(h):
            for cx in range(w):
                for i in range(nA):
                    ind = b * sz_hwa + i * sz_hw + cy * w + cx
                    conf = det[ind] if only_objectness else det[ind] * cmax[ind]
                    if det[ind] > max_conf and cls_conf[ind, correspondingclass] > max_cls_conf:
                        max_conf, max_cls_conf, max_ind = det[ind], cls_conf[ind, correspondingclass], ind
                    if conf > conf_thresh:
                        boxes.append(make(ind, det[ind], cmax[ind], cid[ind]))
        if len(boxes) == 0 or correspondingclass not in [bx[2 * K + 2] for bx in boxes]:
            boxes.append(make(max_ind, max_conf, max_cls_conf, correspondingclass))
        all_boxes.append(boxes)
    return all_boxes
