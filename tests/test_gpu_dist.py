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
