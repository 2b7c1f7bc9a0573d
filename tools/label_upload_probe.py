#!/usr/bin/env python
"""Host time of RegionLoss.forward with HOST float64 labels (what train.py:83-97 hands over), call by call.

Round 2 committed one 8.3 ms average for this call (profiles/r02_infer.json) next to 84-95 us on two other visits.  This
probe times every call on the host (perf_counter around the call, no synchronize inside the loop, a synchronize between
rounds), 4 rounds x 500 calls per mode, and reports per round the median / p99 / max of (a) the whole call and (b) the label
staging alone (RegionLoss.upload_host_us: ring wait, host copy, H2D issue), the sub-steps of the slowest call, and the
calls above 0.5 ms.  Modes:
  copy            the product path: pinned ring, single-thread host copy, asynchronous H2D copy, same label tensor each call
  mapped          the kernel reads the pinned buffer in place (no H2D copy)
  device          labels already on the device (no staging at all)
  no_events       copy without the ring's events (diagnostic: reuse protection off)
  aten_staging    copy, but the host copy done by Tensor.copy_ - the round-2 form: ATen splits the 537 KB copy over its
                  intra-op thread pool and, on the 256-thread host, about one call in 100 then waits 84 / 94 ms for a parked
                  worker thread (the stall sits entirely in that one statement: 'host_copy_us' of the worst call)
  caller_clones   copy, but the CALLER makes a fresh label tensor with Tensor.clone() every 7th call inside the timed region
                  - the same ATen parallel host copy, this time on the caller's side (what a DataLoader's collate does)
Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from singleshotpose_amd.region_loss import RegionLoss, RegionLossMulti
    dev = torch.device('cuda', 0)
    rounds, calls = 4, 500
    out = {}
    anchors = [1.4820, 2.2412, 2.0501, 3.1265, 2.3946, 4.6891, 3.1018, 3.9910, 3.4879, 5.8851]
    for name, crit, ch, nlab, mode in (('single_copy', RegionLoss(), 20, 1, 'copy'), ('single_mapped', RegionLoss(), 20, 1, 'mapped'),
                                       ('single_device_labels', RegionLoss(), 20, 1, 'device'),
                                       ('single_copy_no_events', RegionLoss(), 20, 1, 'noevent'),
                                       ('single_aten_staging', RegionLoss(), 20, 1, 'aten'),
                                       ('single_caller_clones', RegionLoss(), 20, 1, 'clones'),
                                       ('multi_copy', RegionLossMulti(anchors=anchors), 160, 8, 'copy')):
        crit.verbose = False
        crit.label_upload = 'mapped' if mode == 'mapped' else 'copy'
        crit._probe_no_events = mode == 'noevent'      # diagnostic only: the ring's reuse protection is off
        crit._probe_aten_staging = mode == 'aten'
        head = torch.randn(64, ch, 13, 13, device=dev, requires_grad=True)
        g = torch.Generator().manual_seed(0)
        t = torch.zeros(64, 50, 21, dtype=torch.float64)
        for k in range(nlab):
            t[:, k, 0] = float(k % 13)
            t[:, k, 1:19] = torch.rand(64, 18, generator=g, dtype=torch.float64) * 0.5 + 0.25
            t[:, k, 19:21] = 0.2
        tgt = t.view(64, -1)
        if mode == 'device':
            tgt = tgt.to(dev)
        rec = []
        for r in range(rounds):
            torch.cuda.synchronize()
            crit.upload_host_us = []
            ts = np.empty(calls)
            for i in range(calls):
                t0 = time.perf_counter()
                crit(head, tgt.clone() if (i % 7 == 0 and mode == 'clones') else tgt, 20)
                ts[i] = (time.perf_counter() - t0) * 1e6
                if mode == 'copy32' and i % 32 == 31:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            det = np.asarray(crit.upload_host_us[-calls:]) if mode != 'device' else np.zeros((calls, 4))
            up = det[:, 0]
            worst = int(np.argmax(ts))
            rec.append(dict(call_us=dict(median=round(float(np.median(ts)), 1), p99=round(float(np.percentile(ts, 99)), 1),
                                         max=round(float(ts.max()), 1)),
                            upload_us=dict(median=round(float(np.median(up)), 1), p99=round(float(np.percentile(up, 99)), 1),
                                           max=round(float(up.max()), 1)),
                            worst_call=dict(index=worst, call_us=round(float(ts[worst]), 1), ring_wait_us=round(float(det[worst, 1]), 1),
                                            host_copy_us=round(float(det[worst, 2]), 1), h2d_issue_us=round(float(det[worst, 3]), 1)),
                            slow_calls=[(int(i), round(float(ts[i]), 1)) for i in np.nonzero(ts > 500.0)[0][:8]]))
        out[name] = rec
    print(json.dumps(out))


if __name__ == '__main__':
    main()
