#!/usr/bin/env python
"""B = 1 eval forward at 672 x 672 (valid.py's operating point) under `rocprofv3 --kernel-trace`: run N forwards, the
trace's last forward is the steady state.  With --print <kernel_trace.csv> lists the launches of the last forward."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(B, n):
    import torch
    from singleshotpose_amd.darknet import Darknet
    torch.manual_seed(0)
    dev = torch.device('cuda', 0)
    model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg')).to(dev).eval()
    x = torch.rand(B, 3, 672, 672, device=dev)
    with torch.no_grad():
        for _ in range(n):
            model(x)
    torch.cuda.synchronize()


def show(path):
    rows = list(csv.DictReader(open(path)))
    ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', '')))
                for r in rows)
    starts = [i for i, k in enumerate(ks) if k[2].startswith('nchw_to_nhwc_kernel')]
    ends = [i for i, k in enumerate(ks) if k[2].startswith('nhwc_to_nchw_kernel')]
    i0, i1 = starts[-1], ends[-1]
    t0 = ks[i0][0]
    busy = 0
    for k in ks[i0:i1 + 1]:
        busy += k[1] - k[0]
        print('%8.1f %7.1f us g%-8s %s' % ((k[0] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[3], k[2].replace('void ', '')[:90]))
    print('forward wall %.1f us, kernels %.1f us, %d launches' % ((ks[i1][1] - t0) / 1e3, busy / 1e3, i1 - i0 + 1))


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--print':
        show(sys.argv[2])
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 12)
