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


# ---- world size 8: the bucket boundaries and the tail rule of the REAL model's gradient layout (bench.py --gpus 8) ----
def _yolo_pose_layout():
    """(lo, hi) of every conv block's slice of the flat gradient buffer, in backward order - engine.Plan.grad_layout's rule
    (weight | bias  or  weight | bn.weight | bn.bias, each padded to 4 floats) applied to cfg/yolo-pose.cfg."""
    from helpers import ROOT
    from singleshotpose_amd.cfg import layer_shapes, parse_cfg
    blocks = parse_cfg(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
    shapes = layer_shapes(blocks, 416, 416)
    pad4 = lambda n: (n + 3) // 4 * 4
    sizes = []
    cin = 3
    chans = []
    for ind, b in enumerate(blocks[1:]):
        w, h, c = shapes[ind]
        if b['type'] == 'convolutional':
            prev = 3 if ind == 0 else chans[ind - 1]
            k = int(b['size'])
            n = pad4(c * prev * k * k) + (pad4(c) * 2 if int(b['batch_normalize']) else pad4(c))
            sizes.append((ind, n))
        chans.append(c)
    out, off = [], 0
    for ind, n in sorted(sizes, reverse=True):
        out.append((off, off + n))
        off += n
    return out, off


def _worker8(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from singleshotpose_amd.dist import GradReducer, init_distributed
    import torch.distributed as dist
    init_distributed('gloo')
    layout, total = _yolo_pose_layout()
    scale = 64                                   # the test moves 1/64 of every slice (0.8 M floats per rank, not 50 M)
    lay = [(lo // scale // 4 * 4, hi // scale // 4 * 4) for lo, hi in layout]
    lay = [(lay[i - 1][1] if i else 0, hi) for i, (lo, hi) in enumerate(lay)]
    n = lay[-1][1]
    red = GradReducer(None, world, bucket_bytes=(32 << 20) // scale, tail_bytes=(4 << 20) // scale)
    flat = torch.full((n,), float(rank + 1))
    flat[::7] = float(rank) * 0.5
    local = flat.clone()
    red.begin(flat)
    for lo, hi in lay:
        red.layer_done(flat, lo, hi)
    red.all_reduce()
    q.put((rank, local.numpy().copy(), flat.numpy().copy(), list(red.launched), n, total))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_all_reduce_eight_ranks_bucket_boundaries_of_the_real_layout():
    """SURVEY.md section 8(e) at the world size the metric is quoted on: 8 ranks (gloo, CPU), yolo-pose.cfg's own gradient
    layout (scaled down 64 x so that the test moves megabytes, not 1.6 GB): SUM over the 8 ranks, identical buckets on every
    rank, contiguous and covering the buffer, six of them - five closed by the 32 MB rule while the 13 x 13 and 26 x 26
    layers finish, the small one for layers 0 - 10 by the tail rule (47 / 38 / 38 / 40 / 36 / 3 MB)."""
    world = 8
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda g: g[0])
    expected = sum(g[1].astype(np.float64) for g in got)
    assert got[0][5] >= 50547764                 # the real buffer: every parameter of yolo-pose.cfg, slices padded to 4
    for g in got:
        assert np.allclose(g[2], expected, rtol=0, atol=1e-5)
        assert g[3] == got[0][3]
    buckets, n = got[0][3], got[0][4]
    assert buckets[0][0] == 0 and buckets[-1][1] == n and all(b[1] == c[0] for b, c in zip(buckets, buckets[1:]))
    sizes_mb = [(hi - lo) * 4 * 64 / 2 ** 20 for lo, hi in buckets]
    assert len(buckets) == 6, sizes_mb
    assert all(s >= 31.5 for s in sizes_mb[:-1]) and sizes_mb[-1] <= 4.2, sizes_mb


def _worker_sync(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from singleshotpose_amd.dist import init_distributed, sync_plans
    import torch.distributed as dist
    init_distributed('gloo')

    class M(object):
        pass
    m = M()
    fn = sync_plans(m)
    assert m._plan_sync is fn
    out = fn([64, 416, 416, 0, 3, 8006413 + rank, 6413, 12813 * (rank + 1)])
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_sync_plans_broadcasts_rank0_codes():
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_sync, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0] == out[1] == [64, 416, 416, 0, 3, 8006413, 6413, 12813]


def test_sync_plans_is_a_noop_without_a_process_group():
    from singleshotpose_amd.dist import sync_plans

    class M(object):
        pass
    m = M()
    assert sync_plans(m) is None and m._plan_sync is None


def test_bench_refuses_to_measure_one_rank_as_many():
    """`python bench.py --gpus 8` without a launcher: on a node without 8 GPUs nothing is measured and the status is non-zero
    (with them it re-executes itself under torch.distributed.run, one rank per GPU)."""
    import subprocess
    import sys
    from helpers import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0'], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        return
    assert p.returncode != 0 and 'nothing measured' in p.stdout, p.stdout[-500:]
    assert '"metric"' not in p.stdout


def test_launch_dp_script_starts_one_process_per_rank_with_the_queue_setting(tmp_path):
    """tools/launch_dp.sh N script: N ranks through torch.distributed.run on 127.0.0.1, GPU_MAX_HW_QUEUES=8 and the dmabuf IPC
    mode exported BEFORE the ranks start."""
    import subprocess
    import sys
    from helpers import ROOT
    probe = tmp_path / 'probe.py'
    # (one file per rank: two processes printing to one pipe interleave their lines)
    probe.write_text("import os\nopen(os.path.join(%r, 'rank' + os.environ['RANK']), 'w').write(' '.join([os.environ['RANK'], "
                     "os.environ['WORLD_SIZE'], os.environ['MASTER_ADDR'], str(os.environ.get('GPU_MAX_HW_QUEUES')), "
                     "str(os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))]))\n" % str(tmp_path))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'GPU_MAX_HW_QUEUES', 'MASTER_PORT')}
    p = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'launch_dp.sh'), '2', str(probe)], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1000:]
    rows = [open(str(tmp_path / ('rank%d' % r))).read().split() for r in range(2)]
    assert rows == [['0', '2', '127.0.0.1', '8', '0'], ['1', '2', '127.0.0.1', '8', '0']], rows


def _worker_plan_sync(rank, world, port, q):
    """Both ranks build the engine's plan of the full network (CPU buffers: no launches), rank 0 carries a tuned-looking code
    set, rank 1 the library defaults; after Plan._sync_codes both hold rank 0's - and the sizes derived from the codes."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import ROOT
    from singleshotpose_amd import engine
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.dist import init_distributed, sync_plans
    import torch.distributed as dist
    init_distributed('gloo')
    model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
    assert sync_plans(model) is not None
    plan = engine.Plan(model, 4, 160, 160, torch.device('cpu'))
    for cs in plan.convs.values():
        cs.plan_dgrad, cs.wgrad_wino = 0, 0
    before = {i: (cs.tile_m, cs.ntile, cs.ws_fwd, cs.stats.numel() if cs.bn else 0) for i, cs in plan.convs.items()}
    if rank == 0:
        plan.convs[8].plan_fwd = 8006413            # F(4x4), 64-row GEMM tiles
        plan.convs[12].plan_fwd = 9006413           # F(2x2)
        plan.convs[13].plan_fwd = 12813             # a direct tile choice
        plan.convs[8].plan_dgrad, plan.convs[8].wgrad_wino = 8006413, 4
        plan.convs[18].plan_dgrad, plan.convs[18].wgrad_wino = 6413, 2
        for i in (8, 12, 13):
            plan._size_layer(plan.convs[i])
        plan.convs[8].wino_ws_floats = 123
    plan._sync_codes(0)
    plan._sync_codes(1)
    out = {i: (cs.plan_fwd, cs.plan_dgrad, cs.wgrad_wino, cs.tile_m, cs.ntile, cs.ws_fwd, cs.stats.numel() if cs.bn else 0,
               getattr(cs, 'wino_ws_floats', None) if cs.wgrad_wino else None) for i, cs in plan.convs.items()}
    q.put((rank, out, before, plan.ws_floats))
    dist.barrier()
    dist.destroy_process_group()


def test_plan_sync_adopts_rank0_codes_and_resizes():
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_plan_sync, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (o, b, w) for r, o, b, w in (q.get(timeout=240) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res[0][0], res[1][0]
    for i in a:      # same codes, same derived sizes (rank 0's own filter-gradient workspace figure aside: it was a dummy)
        assert a[i][:7] == b[i][:7], (i, a[i], b[i])
    assert (a[8][0], a[8][1], a[8][2]) == (8006413, 8006413, 4) and a[12][0] == 9006413 and a[13][0] == 12813
    assert a[18][1:3] == (6413, 2)
    assert b[8][3] == 0 and b[12][3] == 0                      # Winograd plans: the counted statistics format (tile_m 0)
    assert b[8][5] > res[1][1][8][2] and res[1][2] >= b[8][5]   # ... a workspace for V | M, and the shared buffer grew with it
    assert b[8][7] and b[8][7] != 123 and b[18][7]             # the filter-gradient workspaces were sized by the adopting rank


def _worker_plan_sync_refused(rank, world, port, q):
    """Rank 1 runs with SSP_WINOGRAD=0 (its launch script switched the Winograd plans off): it cannot adopt rank 0's Winograd
    codes, so NO rank does - every rank falls back to the library's heuristic plans (code 0), identically."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if rank == 1:
        os.environ['SSP_WINOGRAD'] = '0'
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import ROOT
    from singleshotpose_amd import engine
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.dist import init_distributed, sync_plans
    import torch.distributed as dist
    init_distributed('gloo')
    model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
    fn = sync_plans(model)
    assert fn is not None and hasattr(fn, 'all_ok')
    plan = engine.Plan(model, 4, 160, 160, torch.device('cpu'))
    for cs in plan.convs.values():
        cs.plan_dgrad, cs.wgrad_wino = 0, 0
    if rank == 0:
        plan.convs[8].plan_fwd = 8006413
        plan.convs[4].plan_fwd = engine.WINOF
        plan.convs[13].plan_fwd = 12813
        plan.convs[8].plan_dgrad, plan.convs[8].wgrad_wino = engine.WINOF, engine.WGRAD_FUSED
        for i in (4, 8, 13):
            plan._size_layer(plan.convs[i])
    else:
        plan.convs[13].plan_fwd = 6414      # this rank's own (direct) choice is dropped as well: same plans everywhere
        plan._size_layer(plan.convs[13])
    plan._sync_codes(0)
    plan._sync_codes(1)
    # a message of another SHAPE (rank 1's plan is for another input size) must neither hang nor be adopted
    other = engine.Plan(model, 4, 160 if rank == 0 else 192, 160, torch.device('cpu'))
    for cs in other.convs.values():
        cs.plan_dgrad, cs.wgrad_wino = 0, 0
    if rank == 0:
        other.convs[13].plan_fwd = 12813
    other._sync_codes(0)
    q.put((rank, {i: (cs.plan_fwd, cs.plan_dgrad, cs.wgrad_wino) for i, cs in plan.convs.items()},
           {i: cs.plan_fwd for i, cs in other.convs.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_plan_sync_is_all_or_none_and_survives_a_shape_mismatch():
    """ADVICE (round 5): the broadcast has a fixed size, the adoption is agreed by a MIN all-reduce - a rank that cannot run rank
    0's codes makes every rank fall back to plan 0; ranks whose plans differ in shape pair up without a size mismatch."""
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_plan_sync_refused, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (a, b) for r, a, b in (q.get(timeout=240) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert all(v == (0, 0, 0) for v in res[r][0].values()), (r, res[r][0])
        assert all(v == 0 for v in res[r][1].values()), (r, res[r][1])
