#!/usr/bin/env python
"""What the REFERENCE network predicts for every (evaluated object, test image) of the OCCLUSION-shaped fixture - golden
generation only (build container, run through oracle/run_reference_cpu.py so that every import is the reference's file):

    cd <fixture>/multi && python oracle/run_reference_cpu.py oracle/multi_predictions.py cfg/yolo-pose-multi.cfg init.weights out.json

Follows valid_multi.py:18-120 up to the choice of `box_pr` (the box of the truth's class with the highest confidence) and
writes its 9 predicted points in pixels.  oracle/gen_dropin_golden.py turns them into the labels_occlusion rows of the
fixture (tests/fixture_occlusion.py explains why).
"""
import json
import sys

import numpy as np
import torch
from torch.autograd import Variable
from torchvision import transforms

import dataset_multi
from cfg import parse_cfg
from darknet_multi import Darknet
from utils_multi import get_multi_region_boxes, read_data_cfg


def main(cfgfile, weightfile, out):
    net_options, loss_options = parse_cfg(cfgfile)[0], parse_cfg(cfgfile)[-1]
    conf_thresh = float(net_options['conf_thresh'])
    K = int(net_options['num_keypoints'])
    nC, nA = int(loss_options['classes']), int(loss_options['num'])
    anchors = [float(a) for a in loss_options['anchors'].split(',')]
    model = Darknet(cfgfile)
    model.load_weights(weightfile)
    model.eval()
    res = {}
    for obj in ('ape', 'can', 'cat', 'duck', 'glue', 'holepuncher'):
        opts = read_data_cfg('cfg/%s_occlusion.data' % obj)
        ds = dataset_multi.listDataset(opts['valid'], shape=(model.width, model.height), shuffle=False, objclass=obj,
                                       transform=transforms.Compose([transforms.ToTensor(), ]))
        res[obj] = {}
        for i in range(len(ds)):
            data, target = ds[i]
            name = ds.lines[i].rstrip().split('/')[-1][:-4]
            output = model(Variable(data.unsqueeze(0))).data
            cls = int(target.view(-1, 2 * K + 3)[0][0])
            boxes = get_multi_region_boxes(output, conf_thresh, nC, K, anchors, nA, cls, only_objectness=0)[0]
            best, box_pr = -sys.maxsize, None
            for b in boxes:
                if (b[2 * K] > best) and (b[2 * K + 2] == cls):
                    best, box_pr = b[2 * K], b
            pts = np.array([float(v) for v in box_pr[:2 * K]]).reshape(K, 2) * np.array([int(opts['im_width']), int(opts['im_height'])])
            res[obj][name] = dict(cls=cls, points_px=pts.tolist(), det_conf=float(best), n_boxes=len(boxes))
    json.dump(res, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:4])
