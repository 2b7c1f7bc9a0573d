"""Fused SGD (ssp_sgd_step / singleshotpose_amd.optim.SGD) against torch.optim.SGD on the CPU - the optimizer the
reference trains with (train.py:388,106)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLD, make_targets, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 3, 4, 1000, 4099, 1 << 20])
@pytest.mark.parametrize("momentum,dampening,wd,nesterov", [(0.9, 0.0, 0.032, False), (0.0, 0.0, 0.0, False),
                                                             (0.9, 0.0, 0.0, True), (0.8, 0.1, 0.01, False)])
def test_sgd_kernel_vs_torch(n, momentum, dampening, wd, nesterov):
    from singleshotpose_amd import _lib
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([ref], lr=0.01, momentum=momentum, dampening=dampening, weight_decay=wd, nesterov=nesterov)
    p = p0.clone().cuda()
    m = torch.zeros_like(p)
    st = torch.cuda.current_stream().cuda_stream
    for step in range(4):
        grad = torch.randn(n, generator=g)
        lr = 0.01 * (0.5 if step >= 2 else 1.0)          # train.py:44-45 rewrites lr between steps
        for grp in opt.param_groups:
            grp['lr'] = lr
        ref.grad = grad.clone()
        opt.step()
        gd = grad.cuda()
        _lib.call('ssp_sgd_step', p.data_ptr(), gd.data_ptr(), m.data_ptr() if momentum else None, n, lr, momentum,
                  dampening, wd, 1 if nesterov else 0, 1 if step == 0 else 0, st)
        torch.cuda.synchronize()
        assert rel_err(p.cpu().numpy(), ref.detach().numpy()) < 1e-6
    if momentum:
        assert rel_err(m.cpu().numpy(), opt.state[ref]['momentum_buffer'].numpy()) < 1e-6


def test_sgd_argument_errors():
    from singleshotpose_amd import _lib
    p = torch.zeros(16, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    with pytest.raises(RuntimeError, match="momentum"):
        _lib.call('ssp_sgd_step', p.data_ptr(), p.data_ptr(), None, 16, 0.1, 0.9, 0.0, 0.0, 0, 1, st)
    with pytest.raises(RuntimeError, match="aligned"):
        _lib.call('ssp_sgd_step', p.data_ptr() + 4, p.data_ptr(), p.data_ptr(), 8, 0.1, 0.9, 0.0, 0.0, 0, 1, st)


def _tiny():
    from oracle.darknet_ref import seeded_state
    from helpers import load_state_into
    from singleshotpose_amd.darknet import Darknet
    model = Darknet(os.path.join(GOLD, 'tiny-pose.cfg'))
    load_state_into(model, model.blocks, seeded_state(model.blocks, 5))
    return model


def _shadow(model, make_opt):
    """CPU copies of the parameters under torch.optim.SGD, fed the GPU run's own gradients every step: the RegionLoss
    is discontinuous (thresholds, cell assignment), so two independently trained copies drift apart chaotically, while
    the optimizer itself is linear in the gradients it is given."""
    params = [torch.nn.Parameter(p.detach().cpu().clone()) for p in model.parameters()]
    return params, make_opt(params)


def _shadow_step(model, sh_params, sh_opt):
    for p, q in zip(model.parameters(), sh_params):
        q.grad = p.grad.detach().cpu().clone()
    sh_opt.step()


def test_fused_optimizer_matches_torch_sgd_over_training_steps():
    """The tiny net trained 4 steps on the HIP path with the fused optimizer (flat layout), shadowed by torch.optim.SGD
    on the CPU.  Also: eval after a fused step must see the new weights."""
    from singleshotpose_amd.optim import SGD
    from singleshotpose_amd.region_loss import RegionLoss
    b = _tiny().cuda().train()
    kw = dict(lr=1e-3 / 4, momentum=0.9, dampening=0, weight_decay=0.0005 * 4)
    ob = SGD(b.parameters(), **kw)
    sh, osh = _shadow(b, lambda ps: torch.optim.SGD(ps, **kw))
    crit = RegionLoss()
    crit.verbose = False
    g = torch.Generator().manual_seed(3)
    for step in range(4):
        x = torch.rand(4, 3, 96, 96, generator=g).cuda()
        tgt = torch.from_numpy(make_targets(np.random.RandomState(step), 4, [1] * 4))
        if step == 2:
            for grp in osh.param_groups + ob.param_groups:
                grp['lr'] = grp['lr'] * 0.1
        ob.zero_grad()
        crit(b(x), tgt, 20).backward()
        _shadow_step(b, sh, osh)
        ob.step()
        for (n, p), q in zip(b.named_parameters(), sh):
            assert rel_err(p.detach().cpu().numpy(), q.detach().numpy()) < 1e-6, (step, n)
            assert rel_err(ob.state[p]['momentum_buffer'].cpu().numpy(), osh.state[q]['momentum_buffer'].numpy()) < 1e-6
    assert ob.fused_steps == 4
    # parameters now live in one flat buffer; the module tree still owns them
    base = ob._flat_p.data_ptr()
    assert all(base <= p.data_ptr() < base + 4 * ob._flat_p.numel() for p in b.parameters())
    # eval right after a fused step uses the updated filters (packed-filter cache invalidated)
    b.eval()
    x = torch.rand(2, 3, 96, 96, generator=g).cuda()
    with torch.no_grad():
        y0 = b(x)
    ob.zero_grad()
    b.train()
    crit(b(torch.rand(4, 3, 96, 96, generator=g).cuda()), torch.from_numpy(make_targets(np.random.RandomState(9), 4, [1] * 4)), 20).backward()
    before = [p.detach().clone() for p in b.parameters()]
    ob.step()
    assert any(not torch.equal(p0, p1) for p0, p1 in zip(before, b.parameters()))
    b.eval()
    with torch.no_grad():
        y1 = b(x)
    fresh = _tiny().cuda().eval()
    fresh.load_state_dict(b.state_dict())
    with torch.no_grad():
        y2 = fresh(x)
    assert not torch.equal(y1, y0)
    assert torch.equal(y1, y2)       # the cached packed filters were refreshed: same result as a freshly built model


def test_fused_optimizer_per_parameter_groups_and_accumulated_grads():
    """Mixed hyper-parameters (the param-group list train.py:381-387 builds) and gradients accumulated over two
    backwards take the per-parameter launches; results still match torch.optim.SGD."""
    from singleshotpose_amd.optim import SGD
    from singleshotpose_amd.region_loss import RegionLoss
    b = _tiny().cuda().train()
    names = [k for k, _ in b.named_parameters()]

    def groups(values):
        out = []
        for key, value in zip(names, values):
            wd = 0.0 if (key.find('.bn') >= 0 or key.find('.bias') >= 0) else 0.002
            out.append({'params': [value], 'weight_decay': wd})
        return out
    ob = SGD(groups(list(b.parameters())), lr=2e-4, momentum=0.9)
    sh, osh = _shadow(b, lambda ps: torch.optim.SGD(groups(ps), lr=2e-4, momentum=0.9))
    crit = RegionLoss()
    crit.verbose = False
    g = torch.Generator().manual_seed(11)
    for step in range(3):
        ob.zero_grad()
        for i in range(2):      # two backwards accumulate into .grad
            x = torch.rand(4, 3, 96, 96, generator=g).cuda()
            crit(b(x), torch.from_numpy(make_targets(np.random.RandomState(10 * step + i), 4, [1] * 4)), 20).backward()
        _shadow_step(b, sh, osh)
        ob.step()
        for (n, p), q in zip(b.named_parameters(), sh):
            assert rel_err(p.detach().cpu().numpy(), q.detach().numpy()) < 1e-6, (step, n)
    assert ob.fused_steps == 0


def test_cpu_parameters_raise():
    from singleshotpose_amd.optim import SGD
    p = torch.nn.Parameter(torch.zeros(8))
    p.grad = torch.ones(8)
    with pytest.raises(RuntimeError, match="HIP kernel only"):
        SGD([p], lr=0.1).step()


def test_fused_optimizer_honours_momentum_restored_by_load_state_dict():
    """Resume: a fresh optimizer that load_state_dict()s saved momentum must continue from it (torch.optim.SGD does),
    both when the state is loaded BEFORE the first fused step (the flat layout adopts the loaded buffers and the step
    is not a 'first' step) and when it is loaded AFTER adoption (the momentum tensors are swapped under the flat
    buffer: the layout is re-adopted)."""
    from singleshotpose_amd.optim import SGD
    from singleshotpose_amd.region_loss import RegionLoss
    kw = dict(lr=1e-3 / 4, momentum=0.9, dampening=0, weight_decay=0.0005 * 4)
    crit = RegionLoss()
    crit.verbose = False

    def one_step(model, opt, seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.rand(4, 3, 96, 96, generator=g).cuda()
        tgt = torch.from_numpy(make_targets(np.random.RandomState(seed), 4, [1] * 4))
        opt.zero_grad()
        crit(model(x), tgt, 20).backward()

    a = _tiny().cuda().train()
    oa = SGD(a.parameters(), **kw)
    for s in (1, 2):
        one_step(a, oa, s)
        oa.step()
    saved_model = {k: v.clone() for k, v in a.state_dict().items()}
    saved_opt = oa.state_dict()
    saved_mom = [oa.state[p]['momentum_buffer'].clone() for p in a.parameters()]

    for when in ('before_first_step', 'after_adoption'):
        b = _tiny().cuda().train()
        b.load_state_dict(saved_model)
        ob = SGD(b.parameters(), **kw)
        if when == 'after_adoption':
            one_step(b, ob, 7)
            ob.step()                               # adopts the flat layout with its own momentum
            b.load_state_dict(saved_model)
        ob.load_state_dict(saved_opt)
        sh, osh = _shadow(b, lambda ps: torch.optim.SGD(ps, **kw))
        for q, m in zip(sh, saved_mom):
            osh.state[q]['momentum_buffer'] = m.detach().cpu().clone()
        one_step(b, ob, 3)
        _shadow_step(b, sh, osh)
        ob.step()
        assert ob.fused_steps >= 1
        for (n, p), q in zip(b.named_parameters(), sh):
            assert rel_err(p.detach().cpu().numpy(), q.detach().numpy()) < 1e-6, (when, n)
            assert rel_err(ob.state[p]['momentum_buffer'].cpu().numpy(), osh.state[q]['momentum_buffer'].numpy()) < 1e-6, (when, n)
