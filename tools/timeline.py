#!/usr/bin/env python
"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: per-queue busy time, overlap, idle gaps.

  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -o t -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline
  python tools/timeline.py out/**/t_kernel_trace.csv
"""
import csv
import sys
from collections import defaultdict


def main(path, verbose):
    rows = list(csv.DictReader(open(path)))
    ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0')) for r in rows]
    ks.sort()
    # last step: from the last nchw_to_nhwc launch that is followed by a region_loss kernel to the last sgd kernel
    starts = [i for i, k in enumerate(ks) if k[2].startswith('nchw_to_nhwc_kernel')]
    sgd = [i for i, k in enumerate(ks) if k[2].startswith('sgd_kernel')]
    i1 = sgd[-1]
    i0 = max(i for i in starts if i < i1 and not any(j for j in sgd if i < j < i1))
    # the forward's first nchw_to_nhwc (the backward uses the same kernel for grad_out)
    cand = [i for i in starts if i < i1 and (not sgd[:-1] or i > sgd[-2])]
    i0 = cand[0]
    step = ks[i0:i1 + 1]
    t0, t1 = step[0][0], step[-1][1]
    loss_i = next(i for i, k in enumerate(step) if 'region_loss' in k[2])
    tl = step[loss_i][0]
    print('step wall %.3f ms: forward %.3f ms, backward+opt %.3f ms, %d launches' %
          ((t1 - t0) / 1e6, (tl - t0) / 1e6, (t1 - tl) / 1e6, len(step)))
    for name, seg in (('forward', [k for k in step if k[0] < tl]), ('backward', [k for k in step if k[0] >= tl])):
        a, b = seg[0][0], max(k[1] for k in seg)
        ev = []
        for s, e, n, q in seg:
            ev.append((s, 1)); ev.append((e, -1))
        ev.sort()
        busy = {0: 0, 1: 0, 2: 0}
        depth, last = 0, a
        for t, d in ev:
            busy[min(depth, 2)] += t - last
            depth += d; last = t
        fam = defaultdict(float)
        for s, e, n, q in seg:
            key = n.split('(')[0][:60]
            fam[key] += (e - s) / 1e6
        qs = defaultdict(float)
        for s, e, n, q in seg:
            qs[q] += (e - s) / 1e6
        print('%s: wall %.3f ms | idle %.3f | one kernel %.3f | >=2 kernels %.3f | per-queue busy %s' %
              (name, (b - a) / 1e6, busy[0] / 1e6, busy[1] / 1e6, busy[2] / 1e6, {q: round(v, 2) for q, v in qs.items()}))
        for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:14]:
            print('    %-62s %7.3f ms' % (k, v))
    if verbose:
        for s, e, n, q in step:
            print('%9.3f %8.3f q%s %s' % ((s - t0) / 1e6, (e - s) / 1e6, q, n[:70]))


if __name__ == '__main__':
    main(sys.argv[1], len(sys.argv) > 2)
