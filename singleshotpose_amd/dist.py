"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce(SUM) over xGMI.

The reference has no distributed code beyond nn.DataParallel(device_ids=[0]) (train_multi.py:387).  Its loss is a SUM
over the batch (MSELoss(size_average=False), region_loss.py:150-152) compensated by lr/batch and decay*batch
(train.py:45,388), so the sharded form must all-reduce gradients with SUM - not DDP's mean - and use the GLOBAL batch
in the lr/decay formula; BatchNorm statistics stay per replica, as under DataParallel.

Gradients of one backward live in ONE flat fp32 buffer laid out in reverse layer order (engine.Plan.grad_layout), so a
bucket is a contiguous slice: as soon as enough trailing layers have finished, its all-reduce is issued asynchronously
(torch.distributed's RCCL stream orders itself after the work already queued on the compute stream) and overlaps the
remaining wgrad/dgrad kernels.  xGMI is point-to-point: a ring all-reduce of the 202 MB gradient is bound by one
~153 GB/s link (~2.3 ms) against a ~50 ms step, so a handful of large buckets is the right shape.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment (torch.distributed.run)."""
    if dist.is_initialized():
        return
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'   # "nccl" is RCCL on ROCm
    dist.init_process_group(backend=backend, rank=int(os.environ.get('RANK', '0')),
                            world_size=int(os.environ.get('WORLD_SIZE', '1')))


class GradReducer(object):
    """Bucketed all-reduce(SUM) of the flat gradient buffer, fed by Plan.backward as layers finish."""

    def __init__(self, model=None, world_size=None, bucket_bytes=48 << 20, group=None, force=False):
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.group = group
        self.force = force      # exercise the collective path even on one rank (tests)
        self._pending = []
        self._flat = None
        self._lo = self._hi = 0
        self.launched = []     # (lo, hi) of every bucket of the last backward (introspection / tests)
        if model is not None:
            model._reducer = self
            for plan in getattr(model, '_plans', {}).values():
                plan.reducer = self if self.active else None

    @property
    def active(self):
        return self.world > 1 or self.force

    def _launch(self, lo, hi):
        if hi <= lo:
            return
        self.launched.append((lo, hi))
        self._pending.append(dist.all_reduce(self._flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def layer_done(self, flat, lo, hi):
        """Gradients flat[lo:hi] are complete (queued on the current stream); ranges arrive in increasing order."""
        if not self.active:
            return
        if self._flat is not flat:
            self._flat, self._lo, self._hi = flat, lo, lo
            self.launched = []
        assert lo == self._hi, "layers must finish in flat-buffer order"
        self._hi = hi
        if self._hi - self._lo >= self.bucket_elems:
            self._launch(self._lo, self._hi)
            self._lo = self._hi

    def all_reduce(self):
        """Flush the open bucket and make the current stream wait for every outstanding all-reduce."""
        if not self.active or self._flat is None:
            return
        self._launch(self._lo, self._hi)
        self._lo = self._hi
        for w in self._pending:
            w.wait()
        self._pending = []
        self._flat = None
