#!/usr/bin/env python
"""PCIe-inclusive step rate: the headline step with its input batch handed over as a HOST buffer each
step, as train.py:82-83 does - pinned float32 NCHW (133 MB), pinned uint8 NHWC (33 MB, what Darknet.forward also takes),
pageable float32 (what `data.cuda()` of an un-pinned DataLoader batch is) - copied on the step's own stream (serial) or on a
copy stream one step ahead (what a pin_memory DataLoader + non_blocking upload gives).  bench.py's `value` never includes
any of this: its inputs are resident.

  python tools/h2d_probe.py [steps=20] [json out]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_batch  # noqa: E402
from singleshotpose_amd.darknet import Darknet  # noqa: E402
from singleshotpose_amd.optim import SGD  # noqa: E402
from singleshotpose_amd.region_loss import RegionLoss  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    B, S = 64, 416
    torch.manual_seed(0)
    m = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg')).cuda().train()
    crit = RegionLoss()
    crit.verbose = False
    opt = SGD(m.parameters(), lr=1e-4 / B, momentum=0.9, weight_decay=0.0005 * B)
    x_dev, tgt = synthetic_batch(B, S, S, 0, 'cuda')
    x_f32 = x_dev.cpu()
    x_u8 = (x_f32 * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    hosts = {'resident': None, 'pinned_f32_nchw_133MB': x_f32.pin_memory(), 'pinned_u8_nhwc_33MB': x_u8.pin_memory(),
             'pageable_f32_nchw_133MB': x_f32}
    copy_stream = torch.cuda.Stream()

    def step(x):
        opt.zero_grad(set_to_none=True)
        loss = crit(m(x), tgt, 20)
        loss.backward()
        opt.step()
        return loss

    def run(host, ahead):
        main_s = torch.cuda.current_stream()
        for it in range(3 + steps):
            if it == 3:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            if host is None:
                x = x_dev
            elif not ahead:
                x = host.cuda(non_blocking=True)
            else:
                if it == 0:
                    with torch.cuda.stream(copy_stream):
                        nxt = host.cuda(non_blocking=True)
                main_s.wait_stream(copy_stream)
                x = nxt
                x.record_stream(main_s)
                with torch.cuda.stream(copy_stream):      # the next batch travels while this one is computed
                    nxt = host.cuda(non_blocking=True)
            step(x)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    res = {}
    step(x_dev)
    step(x_u8.cuda())
    torch.cuda.synchronize()
    for name, host in hosts.items():
        for ahead in ((False,) if host is None else (False, True)):
            ms = run(host, ahead)
            key = name + ('' if host is None else ('_copy_stream_one_ahead' if ahead else '_same_stream'))
            res[key] = dict(ms_per_step=round(ms, 3), images_per_s=round(B / ms * 1e3, 1))
            print(key, res[key], flush=True)
    # the bare copies
    for name, host in hosts.items():
        if host is None:
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            host.cuda(non_blocking=True)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res['copy_' + name] = dict(ms=round(ms, 3), gb_per_s=round(host.numel() * host.element_size() / ms / 1e6, 1))
        print('copy_' + name, res['copy_' + name], flush=True)
    if out_path:
        json.dump(dict(what='headline training step (B=64, 416x416) with the input batch handed over as a host buffer each step',
                       steps=steps, results=res), open(out_path, 'w'), indent=1)


if __name__ == '__main__':
    main()
