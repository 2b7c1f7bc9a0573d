#!/usr/bin/env python
"""Hot-path benchmark: yolo-pose.cfg training step (zero_grad, forward, RegionLoss, backward, grad all-reduce, SGD) on
synthetic 416x416 batches, 64 images per GPU (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `value` = images/s of the whole job (all ranks), max-over-ranks time of exactly K
steps between barrier + synchronize pairs, inputs resident in HBM.  `roofline` is for the dominant kernel
(the implicit-GEMM conv kernel): algorithmic conv FLOPs of its forward launches / their HIP-event time, against the
fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md), from HIP events around exactly those launches inside the timed
region; the backward conv launches overlap on two streams and are reported together in `roofline_bwd`, which - like
`kernel_ms_per_step` - comes from a second, untimed pass with every launch bracketed by events.  `cpu_baseline` times the CPU oracle of the same step
(oracle/: the reference's PyTorch-CPU semantics) on this host for a bounded batch - rank 0, N=1 only.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz


def synthetic_batch(B, H, W, seed, device):
    """images ~ U[0,1) (what ToTensor yields), one label per image: class 0, 9 keypoints in (0.25,0.75), ranges 0.2."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, H, W, generator=g)
    t = torch.zeros(B, 50, 21, dtype=torch.float64)
    t[:, 0, 1:19] = torch.rand(B, 18, generator=g, dtype=torch.float64) * 0.5 + 0.25
    t[:, 0, 19:21] = 0.2
    return x.to(device), t.view(B, -1)   # labels stay on the host (float64), as train.py:83 leaves them


def cpu_baseline(cfgfile, B, H, W, budget_s=25.0):
    """The same step on the CPU oracle (reference semantics, PyTorch-CPU kernels): forward, RegionLoss, backward."""
    from oracle.darknet_ref import forward_ref, seeded_state
    from oracle.region_loss_ref import region_loss_ref
    from singleshotpose_amd.cfg import parse_cfg
    blocks = parse_cfg(cfgfile)
    state = seeded_state(blocks, 0)
    for e in state:
        if e is not None:
            for k, v in e.items():
                if not k.startswith('running'):
                    v.requires_grad_(True)
    x, tgt = synthetic_batch(B, H, W, 0, 'cpu')
    times = []
    t_begin = time.time()
    for it in range(4):
        t0 = time.time()
        y = forward_ref(blocks, state, x, training=True)
        r = region_loss_ref(y.detach(), tgt, 20)
        y.backward(r['grad'])
        for e in state:
            if e is not None:
                for v in e.values():
                    v.grad = None
        dt = time.time() - t0
        if it > 0:
            times.append(dt)
        if time.time() - t_begin > budget_s and times:
            break
    med = float(np.median(times))
    return {"value": round(B / med, 3), "unit": "images/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": "%d steps of fwd+RegionLoss+bwd at batch %d, %dx%d, after 1 warm-up (median)" % (len(times), B, H, W),
            "host_cpus": os.cpu_count()}


def forward_traffic_per_launch():
    """HBM bytes per forward launch of the conv kernel, from the PMC passes committed under profiles/ (FETCH_SIZE
    doubled per MI355X_MICROARCH.md + WRITE_SIZE; tools/gpu_check.sh, tools/traffic_summary.py).  PMC counters cannot
    be read from inside this process, so the figure is that of the last committed profile of this same command."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')))
    if not files:
        return None, None
    data = json.load(open(files[-1]))
    tot, n = 0.0, 0
    for name, v in data.items():
        # conv_igemm_dma_kernel<BM, BN, PASS, ...> and conv_igemm_kernel<BM, BN, WM, WN, BK, ABL, PASS> (first layer):
        # PASS 0 = forward launches, 1 = data-gradient launches of the same code
        fwd = (re.search(r'conv_igemm_dma_kernel<\d+, \d+, 0[,>]', name) is not None or
               re.search(r'conv_igemm_kernel<(\d+, ){6}0>', name) is not None)
        if fwd:
            tot += v['launches'] * (v['fetch_bytes_per_launch_corrected'] + v['write_bytes_per_launch_reported'])
            n += v['launches']
    return (round(tot / n) if n else None), os.path.relpath(files[-1], ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64, help='images per GPU')
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--cfg', default=os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--timers', default='conv', choices=['conv', 'all', 'none'],
                    help='launches bracketed by HIP events INSIDE the timed region: conv = forward conv launches only '
                         '(what the roofline needs), all = every launch, none')
    ap.add_argument('--opt', default='', help='kernel experiment knobs, name=value,... (ssp_set_option); default: none')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)

    from singleshotpose_amd import _lib
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.dist import GradReducer, init_distributed
    from singleshotpose_amd.optim import SGD
    from singleshotpose_amd.region_loss import RegionLoss

    if world > 1:
        init_distributed()
    for kv in filter(None, args.opt.split(',')):
        name, val = kv.split('=')
        _lib.call('ssp_set_option', name.encode(), int(val))

    torch.manual_seed(0)                       # identical initial weights on every rank
    model = Darknet(args.cfg).to(device).train()
    crit = RegionLoss()
    crit.verbose = False                        # status line (one host sync per step) off inside the timed region
    B, H, W = args.batch, args.size, args.size
    global_batch = B * world
    # the reference's sum-loss convention (train.py:45,388): lr / batch, decay * batch with the GLOBAL batch
    # torch.optim.SGD's update rule (train.py:388) as one fused launch over the flat parameter/gradient/momentum buffers
    opt = SGD(model.parameters(), lr=1e-3 / global_batch, momentum=0.9, dampening=0, weight_decay=0.0005 * global_batch)
    reducer = GradReducer(model, world)
    x, tgt = synthetic_batch(B, H, W, 1000 + rank, device)

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(x)
        loss = crit(out, tgt, 20)               # epoch 20 > pretrain: confidence term active
        loss.backward()
        reducer.all_reduce()                    # RCCL SUM over ranks (no-op for world 1)
        opt.step()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    import ctypes
    nk = _lib.query('ssp_prof_nkinds')

    def collect():
        ms = (ctypes.c_double * nk)()
        work = (ctypes.c_double * nk)()
        cnt = (ctypes.c_int64 * nk)()
        _lib.call('ssp_prof_collect', ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(work, ctypes.c_void_p),
                  ctypes.cast(cnt, ctypes.c_void_p))
        return list(ms), list(work), list(cnt)

    # Timed region: HIP events only around the FORWARD launches of the dominant kernel (what `roofline` needs; every
    # timed launch adds two event packets to its stream: all ~240 launches of a step timed cost 0.9 ms, these 23 cost
    # <0.1 ms).  --timers all / none change that.
    _lib.call('ssp_prof_enable', {'conv': 0b001, 'all': -1, 'none': 0}[args.timers])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    _lib.call('ssp_prof_enable', 0)
    ms, work, cnt = collect()
    # Per-family breakdown (kernel_ms_per_step, roofline_bwd): a separate, untimed pass of the same steps with every
    # launch bracketed by events.
    nb = min(args.steps, 5)
    _lib.call('ssp_prof_enable', -1)
    for _ in range(nb):
        loss = step()
    barrier()
    _lib.call('ssp_prof_enable', 0)
    bms, bwork, bcnt = collect()
    final_loss = float(loss)

    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        kinds = _lib.PROF_KINDS
        prof = {kinds[k]: {"ms_per_step": bms[k] / nb, "launches_per_step": bcnt[k] / nb,
                           "work_per_step": bwork[k] / nb} for k in range(nk)}
        # Dominant kernel = the implicit-GEMM conv kernel.  Its FORWARD launches run alone on the GPU, so their HIP-event
        # durations are kernel-exclusive; the same kernel's data-gradient launches overlap the filter-gradient kernel
        # on a second stream (Plan.backward), which stretches both their event durations - they are reported apart.
        ig_ms, ig_flop, ig_n, ig_steps = ms[0], work[0], cnt[0], args.steps
        if ig_ms <= 0:      # --timers none: take the forward launches of the breakdown pass
            ig_ms, ig_flop, ig_n, ig_steps = bms[0], bwork[0], bcnt[0], nb
        achieved = ig_flop / (ig_ms * 1e-3) / 1e12 if ig_ms > 0 else 0.0
        bwd_ms = max(bms[1], bms[2])   # the two streams run concurrently: wall time of the conv backward ~ the longer one
        bwd_tf = (bwork[1] + bwork[2]) / (bwd_ms * 1e-3) / 1e12 if bwd_ms > 0 else 0.0
        traffic, traffic_src = forward_traffic_per_launch()
        images_per_s = global_batch * args.steps / dt
        res = {
            "metric": "images/sec (fwd+bwd) yolo-pose 416x416 bs=64/GPU",
            "value": round(images_per_s, 2),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "cfg/yolo-pose.cfg train step (zero_grad+fwd+RegionLoss+bwd+grad all-reduce+SGD), "
                                   "%dx%d, batch %d/GPU, random-init weights, 1 label/image" % (H, W, B),
                       "global_batch": global_batch, "parallelism": "dp%d" % world},
            "roofline": {"bound": "mfma", "kernel": "conv_igemm_dma_kernel<*,*,0,...> / conv_igemm_kernel forward launches",
                         "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "avg_launch_ms": round(ig_ms / max(ig_n, 1), 4),
                         "launches_per_step": ig_n / ig_steps,
                         "flop_per_launch_avg": ig_flop / max(ig_n, 1)},
            "roofline_bwd": {"bound": "mfma", "kernel": "conv dgrad (stream 1) overlapped with conv_wgrad_kernel (stream 2)",
                             "achieved": round(bwd_tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(bwd_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                             "note": "algorithmic dgrad+wgrad FLOPs / max(sum of dgrad event times, sum of wgrad event times); "
                                     "from a separate untimed pass with every launch timed, as kernel_ms_per_step"},
            "step_conv_flop_frac_of_peak": round(images_per_s / world * 87.673e9 / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4),
            "kernel_ms_per_step": {k: round(v["ms_per_step"], 3) for k, v in prof.items()},
            "final_loss": final_loss,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.cfg, args.cpu_batch, H, W)
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
