"""Data-parallel gradient exchange on CPU: 2 processes, gloo, 127.0.0.1 (the N>1 path of bench.py minus the GPU)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from singleshotpose_amd.dist import GradReducer, init_distributed
    import torch.distributed as dist
    init_distributed('gloo')
    assert dist.get_world_size() == world
    # layer gradients arrive in flat-buffer order (reverse layer order), sizes like a small conv net
    sizes = [4000, 12, 900, 30000, 64, 64, 70000, 5]
    offs = np.concatenate([[0], np.cumsum([(s + 3) // 4 * 4 for s in sizes])])
    total = int(offs[-1])
    red = GradReducer(None, world, bucket_bytes=40000 * 4, tail_bytes=0)
    results = []
    flat = torch.empty(total)
    for step in range(4):                                   # the SAME flat buffer every backward, as Plan.backward reuses it
        g = torch.Generator().manual_seed(100 * step + rank)
        flat.copy_(torch.randn(total, generator=g))
        local = flat.clone()
        red.tail_elems = 71000 if step == 1 else 0          # step 1: the tail rule closes a bucket early
        red.bucket_elems = 10 ** 9 if step == 3 else 40000  # step 3: nothing reaches a bucket - only an OPEN bucket is left
        red.begin(flat)
        for i in range(len(sizes)):
            red.layer_done(flat, int(offs[i]), int(offs[i + 1]))
        if step < 2:
            red.all_reduce()
        else:                                               # a backward that is never joined: the next begin() joins it
            if step == 3:
                assert not red._pending and not red.launched    # (no launched bucket to remind it: the open range must)
            red.begin(flat)
            assert not red._pending
            red.all_reduce()
        # numpy, not tensors: a tensor crosses the queue as a file descriptor that dies with this process
        results.append((local.numpy().copy(), flat.numpy().copy(), list(red.launched)))
    q.put((rank, results))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_all_reduce_sum_two_ranks():
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step in range(4):
        expected = out[0][step][0] + out[1][step][0]        # SUM, not mean (reference loss is a batch sum)
        for r in range(world):
            assert np.allclose(out[r][step][1], expected, rtol=0, atol=1e-6)
        if step >= 2:
            continue                                        # joined by begin(): the bucket list was reset
        buckets = out[0][step][2]
        assert buckets == out[1][step][2]
        # contiguous, ordered, covering the whole buffer
        assert buckets[0][0] == 0 and buckets[-1][1] == expected.size
        assert all(b[1] == c[0] for b, c in zip(buckets, buckets[1:]))
        if step == 0:     # size rule only: every bucket but the last reaches the threshold
            assert all(b[1] - b[0] >= 40000 for b in buckets[:-1]) and len(buckets) == 2
        else:             # tail rule: the first bucket closes as soon as <= 71000 elements are still to come
            assert buckets == [(0, 34912), (34912, 105040), (105040, 105048)], buckets


def test_reducer_is_inert_for_one_rank():
    from singleshotpose_amd.dist import GradReducer
    red = GradReducer(None, 1)
    flat = torch.ones(16)
    red.layer_done(flat, 0, 16)
    red.all_reduce()
    assert torch.equal(flat, torch.ones(16)) and not red.active
