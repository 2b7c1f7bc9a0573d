#!/usr/bin/env python
"""Writes the Darknet-19 pose network definitions under cfg/ from a layer table.

The .cfg text format is the boundary the reference's train.py / valid.py pass around (--modelcfg); the files written
here describe the same networks as the reference's cfg/yolo-pose.cfg and
multi_obj_pose_estimation/cfg/yolo-pose-multi.cfg, regenerated from the table below rather
than copied.  Run:  python tools/make_cfgs.py
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (kind, filters, size) - 'c' conv+BN+leaky, 'm' maxpool 2/2
TRUNK = [('c', 32, 3), ('m',), ('c', 64, 3), ('m',),
         ('c', 128, 3), ('c', 64, 1), ('c', 128, 3), ('m',),
         ('c', 256, 3), ('c', 128, 1), ('c', 256, 3), ('m',),
         ('c', 512, 3), ('c', 256, 1), ('c', 512, 3), ('c', 256, 1), ('c', 512, 3), ('m',),
         ('c', 1024, 3), ('c', 512, 1), ('c', 1024, 3), ('c', 512, 1), ('c', 1024, 3),
         ('c', 1024, 3), ('c', 1024, 3)]


def conv(filters, size, bn=True, act='leaky'):
    lines = ['[convolutional]']
    if bn:
        lines.append('batch_normalize=1')
    lines += ['filters=%d' % filters, 'size=%d' % size, 'stride=1', 'pad=1', 'activation=%s' % act, '']
    return lines


def network(net_opts, head_filters, region_opts):
    out = ['[net]'] + ['%s=%s' % kv for kv in net_opts] + ['']
    for item in TRUNK:
        if item[0] == 'c':
            out += conv(item[1], item[2])
        else:
            out += ['[maxpool]', 'size=2', 'stride=2', '']
    # passthrough: 26x26x512 -> 1x1x64 -> reorg -> concat with the 13x13x1024 trunk
    out += ['[route]', 'layers=-9', '']
    out += conv(64, 1)
    out += ['[reorg]', 'stride=2', '']
    out += ['[route]', 'layers=-1,-4', '']
    out += conv(1024, 3)
    out += conv(head_filters, 1, bn=False, act='linear')
    out += ['[region]'] + ['%s=%s' % kv for kv in region_opts] + ['']
    return '\n'.join(out)


COMMON_REGION = [('bias_match', 1), ('coords', 18), ('softmax', 1), ('jitter', '.3'), ('rescore', 1),
                 ('object_scale', 5), ('noobject_scale', '0.1'), ('class_scale', 1), ('coord_scale', 1),
                 ('absolute', 1), ('thresh', '.6'), ('random', 1)]

SINGLE_NET = [('batch', 8), ('height', 416), ('width', 416), ('channels', 3), ('num_keypoints', 9),
              ('momentum', 0.9), ('decay', 0.0005), ('angle', 0), ('burn_in', 1000), ('max_batches', 80200),
              ('policy', 'steps'), ('max_epochs', 500), ('learning_rate', 0.001), ('steps', '-1,80,160'),
              ('scales', '0.1,0.1,0.1'), ('conf_thresh', 0.1), ('test_width', 672), ('test_height', 672),
              ('saturation', 1.5), ('exposure', 1.5), ('hue', '.1')]

MULTI_NET = [('batch', 32), ('subdivisions', 8), ('height', 416), ('width', 416), ('channels', 3),
             ('num_keypoints', 9), ('momentum', 0.9), ('decay', 0.0005), ('angle', 0), ('saturation', 1.5),
             ('exposure', 1.5), ('hue', '.1'), ('learning_rate', 0.001), ('burn_in', 1000), ('max_batches', 80200),
             ('policy', 'steps'), ('steps', '-1,100,20000,30000'), ('scales', '0.1,10,.1,.1'), ('conf_thresh', 0.05),
             ('max_epochs', 500)]

MULTI_ANCHORS = '1.4820, 2.2412, 2.0501, 3.1265, 2.3946, 4.6891, 3.1018, 3.9910, 3.4879, 5.8851'


def main():
    os.makedirs(os.path.join(ROOT, 'cfg'), exist_ok=True)
    single = network(SINGLE_NET, 20, [('anchors', ''), ('classes', 1), ('num', 1)] + COMMON_REGION)
    multi = network(MULTI_NET, 160, [('anchors', MULTI_ANCHORS), ('classes', 13), ('num', 5)] + COMMON_REGION)
    with open(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'), 'w') as f:
        f.write(single)
    with open(os.path.join(ROOT, 'cfg', 'yolo-pose-multi.cfg'), 'w') as f:
        f.write(multi)
    # camera / dataset description in the reference's .data format (values of LINEMOD "ape", cfg/ape.data)
    with open(os.path.join(ROOT, 'cfg', 'ape.data'), 'w') as f:
        f.write('\n'.join(['train  = LINEMOD/ape/train.txt', 'valid  = LINEMOD/ape/test.txt', 'backup = backup/ape',
                           'mesh = LINEMOD/ape/ape.ply', 'tr_range = LINEMOD/ape/training_range.txt', 'name = ape',
                           'diam = 0.103', 'gpus = 0', 'width = 640', 'height = 480', 'fx = 572.4114',
                           'fy = 573.5704', 'u0 = 325.2611', 'v0 = 242.0489', '']))


if __name__ == '__main__':
    main()
