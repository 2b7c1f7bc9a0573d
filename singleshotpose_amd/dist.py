"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce(SUM) over xGMI.

The reference has no distributed code beyond nn.DataParallel(device_ids=[0]) (train_multi.py:387).  Its loss is a SUM
over the batch (MSELoss(size_average=False), region_loss.py:150-152) compensated by lr/batch and decay*batch
(train.py:45,388), so the sharded form must all-reduce gradients with SUM - not DDP's mean - and use the GLOBAL batch
in the lr/decay formula; BatchNorm statistics stay per replica, as under DataParallel.

Gradients of one backward live in ONE flat fp32 buffer laid out in reverse layer order (engine.Plan.grad_layout), so a
bucket is a contiguous slice: as soon as enough trailing layers have finished, its all-reduce is issued asynchronously
(torch.distributed's RCCL stream orders itself after the work already queued on the compute stream) and overlaps the
remaining wgrad/dgrad kernels.  xGMI is point-to-point: a ring all-reduce of the 202 MB gradient is bound by one
~153 GB/s link (~2.3 ms) against a ~27 ms step, so a handful of large buckets is the right shape - with a SMALL last one,
because only the last bucket cannot overlap anything (GradReducer's tail rule).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment (torch.distributed.run).

    Call order: this function first - it binds the process to ITS GPU (LOCAL_RANK) before
    anything creates a HIP context or a stream, then creates the engine's second compute stream, then the process group.
    A process that reaches it with another current device than its LOCAL_RANK's (somebody already called
    torch.cuda.set_device) keeps that choice; one that never chose gets cuda:LOCAL_RANK, not cuda:0 for every rank."""
    if dist.is_initialized():
        return
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'   # "nccl" is RCCL on ROCm
    if backend == 'nccl' and torch.cuda.is_available():
        local = int(os.environ.get('LOCAL_RANK', '0'))
        if local < torch.cuda.device_count() and (not torch.cuda.is_initialized() or torch.cuda.current_device() == 0):
            torch.cuda.set_device(local)
        elif torch.cuda.current_device() != local:
            import warnings
            warnings.warn("singleshotpose_amd.dist: LOCAL_RANK is %d but the current device is cuda:%d - the side stream and the "
                          "process group are created on cuda:%d" % (local, torch.cuda.current_device(), torch.cuda.current_device()))
        # the engine's second compute stream first: HIP hands its streams to a few hardware queues in creation order, and
        # behind RCCL's own streams it would share the main stream's queue (engine._side_stream)
        from .engine import _side_stream
        _side_stream(torch.device('cuda', torch.cuda.current_device()))
    dist.init_process_group(backend=backend, rank=int(os.environ.get('RANK', '0')),
                            world_size=int(os.environ.get('WORLD_SIZE', '1')))


def sync_plans(model, group=None):
    """Every rank runs the SAME kernel plan set: rank 0's.

    The autotuner times candidates per process, and a near-tie can fall either way on two GPUs: ranks would then run
    different tile / split / Winograd choices - different fp32 rounding in the forward pass and in the BatchNorm statistics
    (gradients agree after the all-reduce either way), and different step times, which in weak scaling show up as
    all-reduce wait on the faster ranks.  With this hook installed every plan, right after its own tuning (forward codes
    after the head-error budget; data- and filter-gradient codes after theirs), takes part in ONE broadcast of rank 0's
    codes (a fixed-size message, so that ranks whose plans differ still pair up) and adopts them - all ranks together, or,
    when any rank cannot (another plan shape, a code its environment switched off), all ranks fall back to the library's
    deterministic heuristic plans for that stage.
    Requirement: all ranks build their plans in the same order (same sequence of input shapes) - bench.py and a training
    loop whose multi-scale schedule is seeded identically on every rank; a loop whose ranks draw shapes independently
    must not install it (the broadcasts would pair up wrongly)."""
    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        model._plan_sync = None
        return None

    MAXN = 1023      # FIXED message size: ranks whose plans differ in shape / conv count still pair up (no size mismatch
                     # inside the collective: NCCL would hang or corrupt before any guard could run)

    def fn(values):
        """Broadcast rank 0's list (at most MAXN integers): returns rank 0's values, whatever this rank passed in."""
        vals = [int(v) for v in values]
        if len(vals) > MAXN:
            raise ValueError("sync_plans: %d values exceed the fixed message size %d" % (len(vals), MAXN))
        dev = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
        t = torch.zeros(MAXN + 1, dtype=torch.int64, device=dev)
        t[0] = len(vals)
        if vals:
            t[1:1 + len(vals)] = torch.tensor(vals, dtype=torch.int64, device=dev)
        dist.broadcast(t, src=0, group=group)
        got = [int(v) for v in t.tolist()]
        n = max(0, min(got[0], MAXN))
        return got[1:1 + n]

    def all_ok(flag):
        """True only when EVERY rank passed a true flag (one MIN all-reduce): ranks adopt rank 0's codes all together, or none."""
        dev = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
        t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return bool(int(t.item()))
    fn.all_ok = all_ok
    model._plan_sync = fn
    return fn


class GradReducer(object):
    """Bucketed all-reduce(SUM) of the flat gradient buffer, fed by Plan.backward as layers finish.

    Buckets are contiguous slices in backward (reverse layer) order.  A bucket closes when it holds >= bucket_bytes, and
    - the tail rule - as soon as what is still to come is <= tail_bytes: the last all-reduce can only start when the
    backward pass ends, so it is kept small (yolo-pose.cfg: 47 / 38 / 38 / 40 / 36 MB buckets while the 13 x 13 and 26 x 26
    layers run, then 3 MB for layers 0-10 instead of a 60 MB bucket exposed after the last wgrad)."""

    def __init__(self, model=None, world_size=None, bucket_bytes=32 << 20, group=None, force=False, tail_bytes=4 << 20):
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.tail_elems = max(0, tail_bytes // 4)
        self.group = group
        self.force = force      # exercise the collective path even on one rank (tests)
        self._pending = []
        self._flat = None
        self._lo = self._hi = 0
        self._tail_closed = False
        self.launched = []     # (lo, hi) of every bucket of the last backward (introspection / tests)
        self.enabled = True    # False: the reducer stands aside (bench.py's untimed local-gradient reference step)
        self.profile = False   # record HIP events per bucket (bench.py's untimed diagnostic steps)
        self._events = []      # (lo, hi, issue event, done event) of the last profiled step
        self._join = None      # (backward-complete event, all-buckets-complete event) of the last profiled step
        self._probe = None     # profiling only: a stream that waits on each collective alone and records its completion
        if model is not None:
            model._reducer = self
            for plan in getattr(model, '_plans', {}).values():
                plan.reducer = self if self.active else None

    @property
    def active(self):
        return self.enabled and (self.world > 1 or self.force)

    def _launch(self, lo, hi):
        if hi <= lo:
            return
        self.launched.append((lo, hi))
        ev = done = None
        if self.profile:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()            # on the stream the bucket's last filter gradient was queued on
        work = dist.all_reduce(self._flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if self.profile and self._flat.is_cuda:
            # the collective's OWN completion: a probe stream that waits for this work alone (not the compute stream's
            # join at the end of backward, which would stamp every bucket with the end of the pass)
            if self._probe is None:
                self._probe = torch.cuda.Stream(device=self._flat.device)
            with torch.cuda.stream(self._probe):
                work.wait()
                done = torch.cuda.Event(enable_timing=True)
                done.record()
        self._pending.append((work, lo, hi, ev, done))

    def begin(self, flat):
        """A backward pass starts writing gradients into `flat` (Plan.backward)."""
        if not self.active:
            return
        # a previous backward's buckets were never joined (no all_reduce() call) - launched ones or just an open one (a
        # model smaller than a bucket, gradient accumulation over two backwards): join them now
        if self._pending or (self._flat is not None and self._lo is not None and self._hi > self._lo):
            self.all_reduce()
        self._flat, self._lo, self._hi, self._tail_closed = flat, None, None, False
        self.launched = []

    def layer_done(self, flat, lo, hi):
        """Gradients flat[lo:hi] are complete (queued on the current stream); ranges arrive in increasing order."""
        if not self.active:
            return
        if self._flat is not flat or self._lo is None:
            if self._flat is not flat:
                self.begin(flat)
            self._lo = self._hi = lo
        assert lo == self._hi, "layers must finish in flat-buffer order"
        self._hi = hi
        remaining = flat.numel() - hi
        close_tail = (not self._tail_closed) and 0 < remaining <= self.tail_elems
        if self._hi - self._lo >= self.bucket_elems or close_tail:
            self._launch(self._lo, self._hi)
            self._lo = self._hi
            self._tail_closed = self._tail_closed or close_tail

    def all_reduce(self):
        """Flush the open bucket and make the current stream wait for every outstanding all-reduce."""
        if not self.active or self._flat is None:
            return
        if self._lo is not None:
            self._launch(self._lo, self._hi)
            self._lo = self._hi
        t0 = None
        if self.profile:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record()            # the current stream has joined the backward pass: everything before this is compute
            self._events = []
        t1 = None
        for work, lo, hi, ev, done in self._pending:
            work.wait()
            if self.profile:
                self._events.append((lo, hi, ev, done))
        if self.profile:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record()            # the compute stream has every collective behind it
            self._join = (t0, t1)
        self._pending = []
        self._flat = None

    def report(self):
        """Diagnostics of the last profiled step (after a synchronize): per bucket the bytes and the time from its issue
        (last filter gradient of the bucket queued) to its completion as seen by the compute stream - queueing behind
        earlier buckets included - and the exposed tail: how long the compute stream sat waiting for collectives after
        the backward pass had ended."""
        out = {"backend": dist.get_backend(self.group) if dist.is_initialized() else None, "ranks": self.world,
               # read back from the process group itself (not from the launcher's environment)
               "group_ranks": dist.get_world_size(self.group) if dist.is_initialized() else None,
               "bucket_bytes": self.bucket_elems * 4, "tail_bytes": self.tail_elems * 4,
               "buckets_bytes": [(hi - lo) * 4 for lo, hi in self.launched]}
        if self._events and self._join is not None:
            torch.cuda.synchronize()
            # issue = the bucket's last filter gradient queued; done = that collective complete (probe stream): queueing
            # behind the bucket before it included, the rest of the backward pass not
            out["bucket_issue_to_done_ms"] = [round(ev.elapsed_time(done), 3) if ev is not None and done is not None else None
                                              for _, _, ev, done in self._events]
            out["exposed_tail_ms"] = round(self._join[0].elapsed_time(self._join[1]), 3)
        return out
