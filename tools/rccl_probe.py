#!/usr/bin/env python
"""What does a live RCCL process group cost the HBM-bound kernels of the SAME process?  (round 4: the single-rank rehearsal
of bench.py ran every HBM-bound family ~1.9x slower - forward pass included, where no collective is in flight.)
Measures a plain device copy (GB/s) and one HBM-bound library kernel at each stage of bringing the group up."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from singleshotpose_amd import _lib


def bw(tag, a, b, x, y, sc, sh):
    torch.cuda.synchronize()
    res = []
    for fn, nbytes in ((lambda: b.copy_(a), 2 * a.numel() * 4),
                       (lambda: _lib.call('ssp_bn_act_fwd', x.data_ptr(), 256, y.data_ptr(), 256, sc.data_ptr(), sh.data_ptr(), 256,
                                          64, 104, 104, 0, 0.1, torch.cuda.current_stream().cuda_stream), 2 * x.numel() * 4)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        e1.synchronize()
        res.append(nbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    print('%-46s copy %7.0f GB/s   bn_act_fwd %7.0f GB/s' % (tag, res[0], res[1]), flush=True)


def main():
    dev = torch.device('cuda', 0)
    a = torch.rand(64 << 20, device=dev); b = torch.empty_like(a)
    x = torch.rand(64 * 104 * 104 * 256, device=dev); y = torch.empty_like(x)
    sc = torch.ones(256, device=dev); sh = torch.zeros(256, device=dev)
    bw('before torch.distributed', a, b, x, y, sc, sh)
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29511')
    dist.init_process_group('nccl', rank=0, world_size=1)
    bw('after init_process_group', a, b, x, y, sc, sh)
    t = torch.ones(1024, device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    bw('after the first (small) all_reduce', a, b, x, y, sc, sh)
    big = torch.ones(48 << 20, device=dev)
    w = dist.all_reduce(big, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    bw('after a 200 MB async all_reduce', a, b, x, y, sc, sh)
    side = torch.cuda.Stream(device=dev, priority=-1)
    with torch.cuda.stream(side):
        w = dist.all_reduce(big, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    bw('after an all_reduce from a priority stream', a, b, x, y, sc, sh)
    dist.barrier()
    bw('after barrier()', a, b, x, y, sc, sh)
    dist.destroy_process_group()
    bw('after destroy_process_group', a, b, x, y, sc, sh)


if __name__ == '__main__':
    main()
