"""RCCL path on one GPU: process group "nccl" (RCCL on ROCm) with world_size 1, reducer forced active, so the bucketed
all-reduce is really issued from Plan.backward's side stream.  (Multi-rank semantics: tests/test_dist_cpu.py.)"""
import os
import socket

import numpy as np
import pytest
import torch

from helpers import GOLD, load_state_into, make_targets, rel_err

pytestmark = pytest.mark.gpu


def test_rccl_bucketed_all_reduce_world1():
    import torch.distributed as dist
    from oracle.darknet_ref import seeded_state
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.dist import GradReducer, init_distributed
    from singleshotpose_amd.region_loss import RegionLoss
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
    init_distributed('nccl')
    try:
        def run(with_reducer):
            model = Darknet(os.path.join(GOLD, 'tiny-pose.cfg'))
            load_state_into(model, model.blocks, seeded_state(model.blocks, 41))
            model = model.cuda().train()
            red = GradReducer(model, 1, bucket_bytes=16 << 10, force=True) if with_reducer else None
            rs = np.random.RandomState(2)
            x = torch.from_numpy(rs.uniform(0, 1, (2, 3, 96, 96)).astype(np.float32)).cuda()
            tgt = torch.from_numpy(make_targets(rs, 2, [1, 1]))
            crit = RegionLoss(); crit.verbose = False
            loss = crit(model(x), tgt, 20)
            loss.backward()
            if red is not None:
                red.all_reduce()
                assert len(red.launched) >= 2 and red.launched[0][0] == 0
            torch.cuda.synchronize()
            return [p.grad.clone() for p in model.parameters()]
        g0, g1 = run(False), run(True)
        for a, b in zip(g0, g1):
            # SUM over one rank = identity; run-to-run differences are the fp32 atomics' summation order
            assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < 1e-4
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
# Model-level data parallelism, world size 2 (SURVEY.md section 4 level 6 / section 8(d) config 3).  Both ranks share
# cuda:0 (the test box has one GPU; RCCL refuses two ranks on one device), so the process group is gloo - its HIP
# build stages device tensors through the host - while everything else is the product path: Darknet on the HIP plan,
# Plan.backward notifying the GradReducer from the filter-gradient stream, bucketed asynchronous all-reduce(SUM) of
# the flat gradient buffer, the fused optimizer consuming the reduced buffer with global-batch lr / decay.
def _dp_worker(rank, world, port, q, backend='gloo'):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: what RCCL needs on this driver
    torch.cuda.set_device(rank if backend == 'nccl' else 0)       # RCCL: one GPU per rank; gloo: both ranks share cuda:0
    from oracle.darknet_ref import seeded_state
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.dist import GradReducer, init_distributed
    from singleshotpose_amd.optim import SGD
    from singleshotpose_amd.region_loss import RegionLoss
    init_distributed(backend)
    per_rank = 2
    global_batch = per_rank * world
    cfg = os.path.join(GOLD, 'tiny-pose.cfg')
    rs = np.random.RandomState(100 + rank)                      # rank-specific data
    x = torch.from_numpy(rs.uniform(0, 1, (per_rank, 3, 96, 96)).astype(np.float32)).cuda()
    tgt = torch.from_numpy(make_targets(rs, per_rank, [1] * per_rank))
    crit = RegionLoss()
    crit.verbose = False
    out = {}

    def build():
        model = Darknet(cfg)
        load_state_into(model, model.blocks, seeded_state(model.blocks, 77))      # identical init on every rank
        return model.cuda().train()

    # (a) one bucket, launched by all_reduce(): the local gradient is complete and can be read before the exchange
    model = build()
    red = GradReducer(model, world, bucket_bytes=1 << 40, tail_bytes=0)
    crit(model(x), tgt, 20).backward()
    torch.cuda.synchronize()
    plan = list(model._plans.values())[0]
    flat = plan.last_flat_grad
    out['local_flat'] = flat.cpu().clone()
    red.all_reduce()
    torch.cuda.synchronize()
    out['reduced_flat_one_bucket'] = flat.cpu().clone()
    assert red.launched == [(0, flat.numel())]

    # (b) the training step as bench.py runs it: small buckets issued from the backward's side stream as layers finish,
    # then the fused SGD step on the reduced buffer
    model = build()
    p0 = [p.detach().cpu().clone() for p in model.parameters()]
    red = GradReducer(model, world, bucket_bytes=16 << 10)
    opt = SGD(model.parameters(), lr=1e-3 / global_batch, momentum=0.9, dampening=0, weight_decay=0.0005 * global_batch)
    opt.zero_grad(set_to_none=True)
    crit(model(x), tgt, 20).backward()
    red.all_reduce()
    torch.cuda.synchronize()
    plan = list(model._plans.values())[0]
    out['buckets'] = list(red.launched)
    out['reduced_flat'] = plan.last_flat_grad.cpu().clone()
    out['reduced_grads'] = [p.grad.detach().cpu().clone() for p in model.parameters()]
    opt.step()
    torch.cuda.synchronize()
    assert opt.fused_steps == 1
    out['params_before'] = p0
    out['params_after'] = [p.detach().cpu().clone() for p in model.parameters()]
    out['running'] = [b.detach().cpu().clone() for n, b in model.named_buffers() if 'running' in n]

    # (c) the same rank-local step with no reducer at all: BatchNorm statistics are per replica (as under DataParallel)
    model = build()
    crit(model(x), tgt, 20).backward()
    torch.cuda.synchronize()
    out['running_single'] = [b.detach().cpu().clone() for n, b in model.named_buffers() if 'running' in n]
    out['local_grads_single'] = [p.grad.detach().cpu().clone() for p in model.parameters()]
    # plain numpy through the queue (torch tensors travel as shared-memory handles that die with this process)
    def np_tree(v):
        if torch.is_tensor(v):
            return v.contiguous().numpy().copy()
        if isinstance(v, list):
            return [np_tree(u) for u in v]
        return v
    q.put((rank, {k: np_tree(v) for k, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_two_ranks_model_level():
    _run_two_ranks('gloo')


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: fewer than 2 GPUs visible")
def test_data_parallel_two_ranks_rccl():
    """The same model-level check under RCCL ("nccl" backend) over xGMI, one GPU per rank - runs wherever two GPUs are
    visible (the driver's multi-GPU node); the single-GPU test box runs the gloo form above."""
    _run_two_ranks('nccl')


def _run_two_ranks(backend):
    import torch.multiprocessing as mp
    world = 2
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    def t_tree(v):
        if isinstance(v, np.ndarray):
            return torch.from_numpy(v)
        if isinstance(v, list) and v and isinstance(v[0], np.ndarray):
            return [torch.from_numpy(u) for u in v]
        return v
    a, b = ({k: t_tree(v) for k, v in res[r].items()} for r in (0, 1))
    # (a) SUM of the two local flat gradients, exactly (one fp32 addition per element), identical on both ranks
    want = a['local_flat'] + b['local_flat']
    assert torch.equal(a['reduced_flat_one_bucket'], want) and torch.equal(b['reduced_flat_one_bucket'], want)
    assert not torch.equal(a['local_flat'], b['local_flat'])                 # rank-specific data
    # (b) bucketed / overlapped exchange: same buckets on both ranks, several of them, covering the buffer in order;
    # the reduced buffer is identical on both ranks and equals the sum of the single-process gradients (<= 1e-5: the
    # filter-gradient atomics make two runs of one rank differ in the last bits)
    assert a['buckets'] == b['buckets'] and len(a['buckets']) >= 2
    assert a['buckets'][0][0] == 0 and a['buckets'][-1][1] == want.numel()
    assert all(u[1] == v[0] for u, v in zip(a['buckets'], a['buckets'][1:]))
    assert torch.equal(a['reduced_flat'], b['reduced_flat'])
    for ga, gs0, gs1 in zip(a['reduced_grads'], a['local_grads_single'], b['local_grads_single']):
        assert rel_err(ga.numpy(), (gs0 + gs1).numpy()) < 1e-5
    # per-replica BatchNorm: running statistics differ between the ranks and equal each rank's single-process run
    assert any(not torch.equal(u, v) for u, v in zip(a['running'], b['running']))
    for r in (a, b):
        for u, v in zip(r['running'], r['running_single']):
            assert rel_err(u.numpy(), v.numpy()) < 1e-6
    # identical parameters after the step on both ranks, equal to ONE torch.optim.SGD step fed the summed gradient with
    # the global-batch lr / decay (train.py:45,388: lr / batch, decay * batch)
    for u, v in zip(a['params_after'], b['params_after']):
        assert torch.equal(u, v)
    shadow = [torch.nn.Parameter(p.clone()) for p in a['params_before']]
    sopt = torch.optim.SGD(shadow, lr=1e-3 / 4, momentum=0.9, dampening=0, weight_decay=0.0005 * 4)
    for sp, g in zip(shadow, a['reduced_grads']):
        sp.grad = g.clone()
    sopt.step()
    for sp, got in zip(shadow, a['params_after']):
        assert rel_err(got.numpy(), sp.detach().numpy()) < 1e-6


def _sync_worker(rank, world, port, q):
    """Two ranks on cuda:0 (gloo): rank 1 runs with the autotuner OFF (every plan code 0), rank 0 tunes.  With
    dist.sync_plans installed both must end up on rank 0's codes - forward, data gradient and the filter gradients' choice."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if rank == 1:
        os.environ['SSP_AUTOTUNE'] = '0'
    torch.cuda.set_device(0)
    from oracle.darknet_ref import seeded_state
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.dist import GradReducer, init_distributed, sync_plans
    from singleshotpose_amd.region_loss import RegionLoss
    init_distributed('gloo')
    from helpers import ROOT
    # the full network (the tiny cfg's layers are below the tuner's thresholds: nothing to agree on), small input
    model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
    load_state_into(model, model.blocks, seeded_state(model.blocks, 77))
    model = model.cuda().train()
    red = GradReducer(model, world)
    assert sync_plans(model) is not None
    rs = np.random.RandomState(100 + rank)
    x = torch.from_numpy(rs.uniform(0, 1, (4, 3, 160, 160)).astype(np.float32)).cuda()
    tgt = torch.from_numpy(make_targets(rs, 4, [1] * 4))
    crit = RegionLoss()
    crit.verbose = False
    crit(model(x), tgt, 20).backward()
    red.all_reduce()
    torch.cuda.synchronize()
    plan = list(model._plans.values())[0]
    codes = [(i, cs.plan_fwd, cs.plan_dgrad, getattr(cs, 'wgrad_wino', 0)) for i, cs in sorted(plan.convs.items())]
    flat = plan.last_flat_grad.cpu().numpy().copy()
    q.put((rank, codes, bool(np.isfinite(flat).all()), flat))
    dist.barrier()
    dist.destroy_process_group()


def test_sync_plans_every_rank_runs_rank0_plan_set():
    import torch.multiprocessing as mp
    world = 2
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (c, ok, f) for r, c, ok, f in (q.get(timeout=600) for _ in range(world))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert any(f or d for _, f, d, _w in res[1][0]), "rank 0 tuned nothing: the test would pass vacuously"
    assert res[0][1] and res[1][1]
    assert np.array_equal(res[0][2], res[1][2])          # the reduced gradient, identical on both ranks
