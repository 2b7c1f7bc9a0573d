"""End-to-end: Darknet on the HIP plan vs golden outputs of the reference Darknet (CPU) and the oracle."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLD, ROOT, clone_state, gold, golden_input, load_state_into, make_targets, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _build(cfgfile, seed):
    from oracle.darknet_ref import seeded_state
    from singleshotpose_amd.darknet import Darknet
    model = Darknet(cfgfile)
    state = seeded_state(model.blocks, seed)
    load_state_into(model, model.blocks, state)
    return model.cuda(), state


def _grad_stats(model, g):
    """Per-parameter distances to the golden's FLOAT64 gradients: (norm error, element error) of the product and of the
    reference's own fp32 run."""
    dn_mine, dn_ref, de_mine, de_ref = [], [], [], []
    for n, p in model.named_parameters():
        gr = p.grad.detach().cpu().numpy()
        n64, n32 = float(g['g64norm/' + n][0]), float(g['gnorm/' + n][0])
        got_norm = float(np.sqrt((gr.astype(np.float64) ** 2).sum()))
        if 'g64/' + n in g.files:
            ref64, ref32, mine = g['g64/' + n], g['grad/' + n], gr
        else:
            ref64, ref32 = g['g64slice/' + n], g['gslice/' + n]
            k = len(ref64)
            mine = gr.reshape(-1)[:: max(1, gr.size // k)][:k]
        dn_mine.append(abs(got_norm / n64 - 1)); dn_ref.append(abs(n32 / n64 - 1))
        de_mine.append(rel_err(mine, ref64)); de_ref.append(rel_err(ref32, ref64))
    return dn_mine, dn_ref, de_mine, de_ref


def _check(cfgfile, tag, B, H, W, seed, with_grad, envelope=3.0, collect=None):
    g = gold('darknet_%s.npz' % tag)
    model, state = _build(cfgfile, seed)
    x = torch.from_numpy(golden_input(g, B, H, W)).cuda()
    model.eval()
    with torch.no_grad():
        y = model(x)
    assert tuple(y.shape) == tuple(g['y_eval'].shape) and y.is_contiguous()
    assert rel_err(y.cpu().numpy(), g['y_eval']) < TOL
    if not with_grad:
        return
    model.train()
    y = model(x)
    assert rel_err(y.detach().cpu().numpy(), g['y_train']) < TOL
    (y * torch.from_numpy(g['probe']).cuda()).sum().backward()
    # Whole-network gradients.  On the 31-layer net the reference's own fp32 result sits ~1e-3 (norms) / ~1e-2 (single
    # elements) away from exact arithmetic: rounding flips max-pool / leaky decisions and the flips propagate.  The
    # golden file therefore also holds the reference run in float64; the GPU path must sit inside the same envelope
    # around the float64 gradients as the reference's fp32 path does (per parameter <= 3x the reference's worst
    # parameter, and on average no worse than 2.5x the reference's average).  Per-kernel and small-net tests keep the
    # strict 1e-4 / 3e-4 bars.
    dn_mine, dn_ref, de_mine, de_ref = _grad_stats(model, g)
    for n, b in model.named_buffers():
        if 'running' in n:
            np.testing.assert_allclose(b.cpu().numpy(), g['buf/' + n], rtol=1e-4, atol=1e-5)
    if collect is not None:      # multi-seed statistic: the caller pools the batches
        collect.append((dn_mine, dn_ref, de_mine, de_ref))
        return
    if max(de_ref) < 1e-4:      # small nets: no flips, strict bar
        assert max(de_mine) < 3e-4 and max(dn_mine) < 3e-4, (max(de_mine), max(dn_mine))
    else:
        assert max(dn_mine) <= envelope * max(dn_ref) and max(de_mine) <= envelope * max(de_ref), \
            (max(dn_mine), max(dn_ref), max(de_mine), max(de_ref))
        assert np.mean(dn_mine) <= 2.5 * np.mean(dn_ref) + 1e-5 and np.mean(de_mine) <= 2.5 * np.mean(de_ref) + 1e-5, \
            (np.mean(dn_mine), np.mean(dn_ref), np.mean(de_mine), np.mean(de_ref))


def test_tiny_matches_reference():
    _check(os.path.join(GOLD, 'tiny-pose.cfg'), 'tiny', 2, 96, 96, 5, True)


def test_full_eval_matches_reference():
    _check(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'), 'full_eval', 1, 416, 416, 6, False)


FULL_TRAIN_GOLDENS = ('full_train', 'full_train_s211', 'full_train_s212', 'full_train_s213', 'full_train_s214')


@pytest.mark.parametrize("tuned", [False, True])
def test_full_train_matches_reference(monkeypatch, tuned):
    """Un-frozen whole-network gradients against the reference's own golden runs: a STATISTICAL bound - which near-tie
    max-pool / leaky decisions flip depends on every rounding upstream, and ONE flipped element that happens to carry a
    channel's gradient moves that parameter by 1e-3.  So the statistic is taken over FIVE input batches (the same seeded
    weights; oracle/gen_golden.py gen_multiseed holds the reference's float32 and float64 runs of each), pooled on both
    sides: the product's worst parameter over the five batches against 3x the reference's own worst (norms and elements),
    its average against 2.5x the reference's average - with the library's default plans (direct kernels) and with the
    autotuner's (Winograd F(4x4) on most 3x3 layers, the forward codes admitted under the head-error budget).  Round 4
    judged this on one batch and had to allow the tuned path 5x (3.1x measured on one box of three); the chunked
    accumulation of round 5 and the error budget brought the kernels' own rounding down, the five-batch statistic takes
    the single-flip noise out - the bound is 3x for both.  The rigorous check of the tuned path stays the decision-frozen
    one (tests/test_gpu_fullsize.py)."""
    if not tuned:
        monkeypatch.setenv('SSP_AUTOTUNE', '0')
    rows = []
    for tag in FULL_TRAIN_GOLDENS:
        _check(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'), tag, 2, 416, 416, 7, True, collect=rows)
    dn_mine, dn_ref, de_mine, de_ref = ([v for r in rows for v in r[k]] for k in range(4))
    per_batch = [(max(r[0]) / max(r[1]), max(r[2]) / max(r[3])) for r in rows]
    print('whole-network gradients over %d batches: worst norm error %.2e (reference %.2e), worst element error %.2e '
          '(reference %.2e); per batch (norm ratio, element ratio): %s' % (
              len(rows), max(dn_mine), max(dn_ref), max(de_mine), max(de_ref), [(round(a, 2), round(b, 2)) for a, b in per_batch]))
    assert max(dn_mine) <= 3.0 * max(dn_ref) and max(de_mine) <= 3.0 * max(de_ref), per_batch
    assert np.mean(dn_mine) <= 2.5 * np.mean(dn_ref) + 1e-5 and np.mean(de_mine) <= 2.5 * np.mean(de_ref) + 1e-5


FULL_TRAIN_B8_GOLDENS = ('full_train_b8_s221', 'full_train_b8_s222')


def test_full_train_b8_matches_reference_with_tuned_plans():
    """The same un-frozen statistic at the batch the cfg ships with (batch=8, yolo-pose.cfg:3; round-5 review, weak #2: the
    five-batch statistic above is at batch 2): the reference's float32 and float64 runs of two 8-image batches
    (oracle/gen_golden.py gen_b8), the autotuner's plans for THIS shape (on-chip Winograd on the 208 x 208 / 104 x 104
    layers, F(4x4) below), pooled: worst parameter <= 3x the reference's own worst, mean <= 2.5x its mean."""
    rows = []
    for tag in FULL_TRAIN_B8_GOLDENS:
        _check(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'), tag, 8, 416, 416, 7, True, collect=rows)
    dn_mine, dn_ref, de_mine, de_ref = ([v for r in rows for v in r[k]] for k in range(4))
    per_batch = [(max(r[0]) / max(r[1]), max(r[2]) / max(r[3])) for r in rows]
    print('whole-network gradients, batch 8, %d batches: worst norm error %.2e (reference %.2e), worst element error %.2e '
          '(reference %.2e); per batch (norm ratio, element ratio): %s' % (
              len(rows), max(dn_mine), max(dn_ref), max(de_mine), max(de_ref), [(round(a, 2), round(b, 2)) for a, b in per_batch]))
    assert max(dn_mine) <= 3.0 * max(dn_ref) and max(de_mine) <= 3.0 * max(de_ref), per_batch
    assert np.mean(dn_mine) <= 2.5 * np.mean(dn_ref) + 1e-5 and np.mean(de_mine) <= 2.5 * np.mean(de_ref) + 1e-5


def test_layerwise_vs_oracle_other_resolution():
    """Multi-scale shapes (dataset.py:66-90 draws 224..832): tiny net at 160x160 (test size) against the oracle."""
    from oracle.darknet_ref import forward_ref
    model, state = _build(os.path.join(GOLD, 'tiny-pose.cfg'), 9)
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.uniform(0, 1, (3, 3, 160, 160)).astype(np.float32))
    model.eval()
    with torch.no_grad():
        y = model(x.cuda()).cpu()
        ref = forward_ref(model.blocks, clone_state(state), x, training=False)
    assert rel_err(y.numpy(), ref.numpy()) < TOL


def test_train_step_with_region_loss_and_sgd():
    """The train.py inner loop (train.py:83-106): forward, RegionLoss, backward, SGD step - vs the same on the oracle."""
    from oracle.darknet_ref import forward_ref
    from oracle.region_loss_ref import region_loss_ref
    from singleshotpose_amd.region_loss import RegionLoss
    model, state = _build(os.path.join(GOLD, 'tiny-pose.cfg'), 12)
    rs = np.random.RandomState(4)
    B = 4
    x = torch.from_numpy(rs.uniform(0, 1, (B, 3, 96, 96)).astype(np.float32))
    tgt = torch.from_numpy(make_targets(rs, B, [1] * B))
    crit = RegionLoss()
    crit.verbose = False
    opt = torch.optim.SGD(model.parameters(), lr=1e-3 / B, momentum=0.9, dampening=0, weight_decay=0.0005 * B)
    model.train()
    opt.zero_grad()
    out = model(x.cuda())
    loss = crit(out, tgt, 20)
    loss.backward()
    # oracle
    st = clone_state(state, requires_grad=True)
    y = forward_ref(model.blocks, st, x, training=True)
    r = region_loss_ref(y.detach(), tgt, 20)
    y.backward(r['grad'])
    assert abs(float(loss) - r['loss']) <= TOL * abs(r['loss'])
    for ind, e in enumerate(st):
        if e is None:
            continue
        seq = model.models[ind]
        assert rel_err(seq[0].weight.grad.cpu().numpy(), e['weight'].grad.numpy()) < 3e-4, ind
    opt.step()
    # weights changed -> next forward must see the new filters (repack is not stale)
    out2 = model(x.cuda())
    assert not torch.equal(out2, out)


def test_weights_round_trip_and_module_tree(tmp_path):
    from singleshotpose_amd.darknet import Darknet
    model, state = _build(os.path.join(GOLD, 'tiny-pose.cfg'), 21)
    p1, p2 = str(tmp_path / 'a.weights'), str(tmp_path / 'b.weights')
    model.seen = 1234
    model.save_weights(p1)
    m2 = Darknet(os.path.join(GOLD, 'tiny-pose.cfg'))
    m2.load_weights(p1)
    assert int(m2.seen) == 1234
    m2.save_weights(p2)
    assert open(p1, 'rb').read() == open(p2, 'rb').read()


def test_cpu_input_raises():
    from singleshotpose_amd.darknet import Darknet
    m = Darknet(os.path.join(GOLD, 'tiny-pose.cfg'))
    with pytest.raises(RuntimeError, match="HIP"):
        m(torch.zeros(1, 3, 96, 96))


def test_inference_pipeline_672():
    """valid.py's path (valid.py:107-153): eval forward at the 672x672 test size, get_region_boxes, PnP."""
    from oracle.darknet_ref import forward_ref
    from oracle.pnp_ref import project, solve_pnp_ref
    from oracle.region_loss_ref import get_region_boxes_ref
    from singleshotpose_amd.utils import get_camera_intrinsic, get_region_boxes, pnp
    model, state = _build(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'), 17)
    assert (model.test_width, model.test_height) == (672, 672)
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.uniform(0, 1, (1, 3, 672, 672)).astype(np.float32))
    model.eval()
    with torch.no_grad():
        out = model(x.cuda())
        ref = forward_ref(model.blocks, clone_state(state), x, training=False)
    assert tuple(out.shape) == (1, 20, 21, 21)
    assert rel_err(out.cpu().numpy(), ref.numpy()) < TOL
    box = get_region_boxes(out, 1, 9)
    box_ref = get_region_boxes_ref(ref, 1, 9)
    np.testing.assert_allclose(np.array([float(v) for v in box]), np.array(box_ref, dtype=np.float64), rtol=2e-4, atol=2e-5)
    corners2d = np.array(np.reshape([float(v) for v in box[:18]], [9, 2]), dtype='float32')
    corners2d[:, 0] *= 640
    corners2d[:, 1] *= 480
    K = np.array(get_camera_intrinsic(325.2611, 242.0489, 572.4114, 573.5704), dtype='float32')
    X = np.concatenate([np.zeros((1, 3)), np.array([[sx * .038, sy * .039, sz * .046] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])], 0).astype('float32')
    R, t = pnp(X, corners2d, K)
    R_o, t_o = solve_pnp_ref(X, corners2d, K)
    assert R.shape == (3, 3) and t.shape == (3, 1)
    # random-weight corners are not a rigid projection: both solvers must land on the same least-squares pose
    assert np.abs(project(X, R, t, K) - project(X, R_o, t_o, K)).max() < 1e-2


def test_multi_object_train_step():
    """yolo-pose-multi style head (5 anchors x 13 classes) on the tiny trunk: forward, multi RegionLoss, backward."""
    from oracle.darknet_ref import forward_ref, seeded_state
    from oracle.region_loss_ref import region_loss_ref
    from singleshotpose_amd.darknet import DarknetMulti
    from singleshotpose_amd.region_loss import RegionLossMulti
    cfg = os.path.join(GOLD, 'tiny-pose-multi.cfg')
    model = DarknetMulti(cfg)
    state = seeded_state(model.blocks, 23)
    load_state_into(model, model.blocks, state)
    model = model.cuda().train()
    assert isinstance(model.models[-1], RegionLossMulti) and model.num_anchors == 5 and model.num_classes == 13
    rs = np.random.RandomState(6)
    B = 4
    x = torch.from_numpy(rs.uniform(0, 1, (B, 3, 96, 96)).astype(np.float32))
    tgt = torch.from_numpy(make_targets(rs, B, [2, 1, 3, 1], multi=True))
    crit = RegionLossMulti(num_keypoints=9, num_classes=13, anchors=model.anchors, num_anchors=5, pretrain_num_epochs=0)
    crit.verbose = False
    out = model(x.cuda())
    assert tuple(out.shape) == (B, 160, 3, 3)
    loss = crit(out, tgt, 1)
    loss.backward()
    st = clone_state(state, requires_grad=True)
    y = forward_ref(model.blocks, st, x, training=True)
    r = region_loss_ref(y.detach(), tgt, 1, num_classes=13, num_anchors=5, anchors=model.anchors, pretrain_num_epochs=0, multi=True)
    y.backward(r['grad'])
    assert rel_err(out.detach().cpu().numpy(), y.detach().numpy()) < TOL
    assert abs(float(loss) - r['loss']) <= TOL * abs(r['loss'])
    for ind, e in enumerate(st):
        if e is not None:
            assert rel_err(model.models[ind][0].weight.grad.cpu().numpy(), e['weight'].grad.numpy()) < 3e-4, ind


@pytest.mark.parametrize("size,B", [(64, 3), (224, 2), (416, 1), (288, 2)])
def test_multiscale_training_shapes(size, B):
    """Multi-scale training changes the input size every batch (dataset.py:66-90: 224..832 in steps of 32); each
    shape gets its own Plan.  Tiny trunk, one training step against the decision-frozen oracle at several grids
    (2x2 ... 13x13): the same 1e-4 bars as the full network (tests/test_gpu_fullsize.py runs that at 224 / 608 / 832)."""
    from oracle.step_check import check_train_step
    from singleshotpose_amd.region_loss import RegionLoss
    model, state = _build(os.path.join(GOLD, 'tiny-pose.cfg'), 30 + size)
    rs = np.random.RandomState(size)
    x = torch.from_numpy(rs.uniform(0, 1, (B, 3, size, size)).astype(np.float32))
    tgt = torch.from_numpy(make_targets(rs, B, [1] * B))
    res = check_train_step(model, RegionLoss(), x, tgt, 20)
    assert tuple(model._plans.keys())[0][:3] == (B, size, size)
    for k in ('head', 'loss', 'running', 'conv', 'grad_out'):
        assert res[k] < TOL, (k, res[k])
    for name, err in res['grad_by_param'].items():
        assert err < (5e-4 if name == '0.weight' else TOL), (name, err)


def test_plan_cache_multiscale_revisit_and_eviction():
    """Multi-scale training revisits shapes (dataset.py:66-90): plans are kept per shape, the autotuned kernel choices
    are shared process-wide (a rebuilt plan re-times nothing), and the cache is bounded by a share of device memory."""
    from singleshotpose_amd import engine
    from singleshotpose_amd.darknet import Darknet
    torch.manual_seed(0)
    model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg')).cuda().eval()
    xs = {s: torch.rand(1, 3, s, s, device='cuda') for s in (224, 256, 288)}
    with torch.no_grad():
        first = {s: model(x).clone() for s, x in xs.items()}
    assert len(model._plans) == 3 and all(p.nbytes_est > 0 for p in model._plans.values())
    tuned = dict(engine._TUNE_CACHE)
    assert any(k[0] == 'fwd' and k[2] == 224 // 32 for k in tuned) and all(isinstance(v, int) for v in tuned.values())
    with torch.no_grad():      # revisiting: same plan objects, same results, nothing re-timed
        plans = dict(model._plans)
        for s, x in xs.items():
            assert torch.equal(model(x), first[s])
        assert all(model._plans[k] is plans[k] for k in plans)
    assert engine._TUNE_CACHE == tuned
    # a second model of the same shapes re-uses the timed choices
    other = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg')).cuda().eval()
    other.load_state_dict(model.state_dict())
    with torch.no_grad():
        assert torch.equal(other(xs[224]), first[224])
    assert engine._TUNE_CACHE == tuned
    # memory bound: with no budget only the newest plan survives
    model._plan_mem_frac = 0.0
    with torch.no_grad():
        model(torch.rand(1, 3, 320, 320, device='cuda'))
    assert list(model._plans.keys()) == [(1, 320, 320, torch.cuda.current_device())]
    model._plan_mem_frac, model._max_plans = 0.5, 2
    with torch.no_grad():
        for s in (224, 256, 288):
            model(xs[s])
    assert [k[1] for k in model._plans] == [256, 288]


def test_evicted_plan_is_freed_while_the_caller_still_holds_its_loss():
    """A training loop holds the previous resolution's loss tensor while it runs the next forward.  Before its backward
    the autograd node keeps its plan (forward A, forward B, backward A works whatever was evicted); after it only the cache
    does, so an evicted plan's buffers really go back to the allocator (the multi-scale soak's peak was two plans)."""
    import gc
    import weakref
    from singleshotpose_amd.darknet import Darknet
    torch.manual_seed(0)
    m = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg')).cuda().train()
    dev = torch.cuda.current_device()
    xa, xb = torch.rand(2, 3, 160, 160, device='cuda'), torch.rand(2, 3, 224, 224, device='cuda')
    loss_a = m(xa).square().sum()
    loss_a.backward(retain_graph=True)                  # (retain_graph: autograd itself will not stop a second call below)
    m.zero_grad(set_to_none=True)
    ref_a = weakref.ref(m._plans[(2, 160, 160, dev)])
    m._plan_mem_frac = 0.0                              # every new shape evicts everything else
    pending = m(xa * 0.5).square().sum()                # forward at A again, NOT yet backpropagated ...
    out_b = m(xb)                                       # ... then shape B evicts A from the cache
    assert list(m._plans) == [(2, 224, 224, dev)]
    assert ref_a() is not None                          # pending backward: the node still owns plan A
    pending.backward()                                  # and it works
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    m.zero_grad(set_to_none=True)
    del pending
    gc.collect()
    if ref_a() is not None:
        holders = [type(r).__name__ + ':' + repr(r)[:120] for r in gc.get_referrers(ref_a())]
        assert False, "plan A outlived its eviction although both backwards had run; held by %s" % holders
    assert float(loss_a.detach()) == float(loss_a.detach()) and out_b.shape[0] == 2      # loss_a itself is still a valid tensor
    with pytest.raises(RuntimeError, match='plan'):     # its node has no plan any more
        loss_a.backward()
    # and the plain double backward on a cached plan: refused by the plan (the chain rewrites the saved conv outputs in place)
    l = m(xb).square().sum()
    l.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match='twice'):
        l.backward()


def test_uint8_image_input_equals_totensor_path():
    """Darknet accepts the decoder's uint8 (B,H,W,3) bytes directly: same output, bit for bit, as feeding the
    ToTensor'd NCHW float tensor the reference's loader builds (dataset.py:113-131); training step included."""
    model, _ = _build(os.path.join(GOLD, 'tiny-pose.cfg'), 77)
    rs = np.random.RandomState(5)
    img = torch.from_numpy(rs.randint(0, 256, (3, 96, 96, 3)).astype(np.uint8))
    x_float = img.permute(0, 3, 1, 2).to(torch.float32).div(255).contiguous()
    model.eval()
    with torch.no_grad():
        assert torch.equal(model(img.cuda()), model(x_float.cuda()))
    model.train()
    outs, grads = [], []
    for inp in (img.cuda(), x_float.cuda()):
        model.zero_grad()
        y = model(inp)
        y.square().sum().backward()
        outs.append(y.detach().clone())
        grads.append(model.models[0][0].weight.grad.clone())
    assert torch.equal(outs[0], outs[1])
    assert rel_err(grads[0].cpu().numpy(), grads[1].cpu().numpy()) < 1e-5      # wgrad sums with fp32 atomics
    with pytest.raises(ValueError):
        model(torch.zeros(1, 3, 96, 96, dtype=torch.uint8, device='cuda'))


def test_filters_are_channels_last_and_used_in_place():
    """Darknet keeps conv filters channels-last in memory (same shape / values / .weights format): the kernels use the
    parameter as the forward operand and accumulate its gradient in place.  A filter a caller replaced by a plain
    contiguous tensor takes the repack path and must give the same numbers."""
    from singleshotpose_amd.engine import _is_packed
    model, state = _build(os.path.join(GOLD, 'tiny-pose.cfg'), 5)
    convs = [m for m in model.modules() if isinstance(m, torch.nn.Conv2d)]
    assert convs[0].weight.is_contiguous() and not _is_packed(convs[0].weight, 4)     # 3 input channels: padded path
    for c in convs[1:]:
        assert c.weight.is_cuda and c.weight.permute(0, 2, 3, 1).is_contiguous()
        assert _is_packed(c.weight, c.weight.size(1))
    model.train()
    g = torch.Generator().manual_seed(1)
    x = torch.rand(3, 3, 96, 96, generator=g).cuda()
    probe = torch.randn(3, 20, 3, 3, generator=g).cuda()
    model.zero_grad()
    y1 = model(x)
    (y1 * probe).sum().backward()
    grads1 = [c.weight.grad.clone() for c in convs]
    for c in convs[1:]:
        assert c.weight.grad.stride() == c.weight.stride()           # gradient view has the parameter's layout
    base = convs[1].weight.grad.untyped_storage().data_ptr()
    assert all(c.weight.grad.untyped_storage().data_ptr() == base for c in convs)     # one flat buffer
    # now the same network with plain contiguous filters (reference layout): repack / unpack path
    for c in convs:
        c.weight.data = c.weight.data.contiguous()
        assert c.weight.is_contiguous()
    model.zero_grad()
    y2 = model(x)
    (y2 * probe).sum().backward()
    assert rel_err(y2.detach().cpu().numpy(), y1.detach().cpu().numpy()) < 1e-6
    for c, g1 in zip(convs, grads1):
        assert c.weight.grad.shape == g1.shape
        assert rel_err(c.weight.grad.cpu().numpy(), g1.cpu().numpy()) < 2e-5      # fp32 atomics: order differs


def test_graph_inference_matches_eager_and_tracks_weight_updates():
    """Eval forward replayed from a captured hipGraph: bit-identical to the eager launch chain, sees in-place parameter
    and running-statistics updates without re-capture, re-captures when a parameter tensor moves."""
    model, _ = _build(os.path.join(GOLD, 'tiny-pose.cfg'), 21)
    model.eval()
    g = torch.Generator().manual_seed(4)
    xs = [torch.rand(2, 3, 96, 96, generator=g).cuda() for _ in range(3)]
    with torch.no_grad():
        model.graph_inference = False
        eager = [model(x).clone() for x in xs]
        model.graph_inference = True
        plan = list(model._plans.values())[0]
        got = [model(x) for x in xs]
        assert plan._graph is not None and not plan._graph_failed
        for a, b in zip(eager, got):
            assert torch.equal(a, b)
        assert got[0].data_ptr() != got[1].data_ptr()          # fresh output tensors, not the graph's static buffer
        graph0 = plan._graph
        # in-place updates (what load_weights / an optimizer step do): same graph, new numbers
        convs = [m for m in model.modules() if isinstance(m, torch.nn.Conv2d)]
        bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        convs[0].weight.data.mul_(1.1)
        convs[2].weight.data.add_(0.01)
        bns[1].running_mean.add_(0.05)
        y_graph = model(xs[0])
        assert plan._graph is graph0 and not torch.equal(y_graph, eager[0])
        model.graph_inference = False
        assert torch.equal(model(xs[0]), y_graph)
        # a parameter that moved to new storage: re-capture
        model.graph_inference = True
        convs[1].weight.data = convs[1].weight.data.clone(memory_format=torch.preserve_format)
        y2 = model(xs[0])
        assert plan._graph is not graph0 and torch.equal(y2, y_graph)
        # uint8 input goes through its own capture
        img = (xs[0].permute(0, 2, 3, 1) * 255).to(torch.uint8).contiguous()
        yu = model(img)
        model.graph_inference = False
        assert torch.equal(model(img), yu)
    # training-mode and autograd forwards never take the graph path
    model.graph_inference = True
    model.train()
    y = model(xs[0])
    y.sum().backward()
    assert convs[0].weight.grad is not None


def test_eval_caches_follow_weight_loading(tmp_path):
    """Inference keeps BN constants and the first layer's padded filter pack between calls; load_weights (which writes
    through .data, as the reference's cfg.py does) and in-place tensor updates must refresh them."""
    from oracle.darknet_ref import seeded_state
    model, _ = _build(os.path.join(GOLD, 'tiny-pose.cfg'), 40)
    other, _ = _build(os.path.join(GOLD, 'tiny-pose.cfg'), 41)
    wfile = str(tmp_path / 'other.weights')
    other.save_weights(wfile)
    x = torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(0)).cuda()
    model.eval(); other.eval()
    with torch.no_grad():
        y_a = model(x).clone()
        y_b = other(x).clone()
        assert not torch.equal(y_a, y_b)
        model.load_weights(wfile)
        assert torch.equal(model(x), y_b)
        bn = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)][0]
        bn.running_var.mul_(2.0)                # in-place on the buffer itself: version counter bumps
        bn_o = [m for m in other.modules() if isinstance(m, torch.nn.BatchNorm2d)][0]
        bn_o.running_var.mul_(2.0)
        assert torch.equal(model(x), other(x))
        assert not torch.equal(model(x), y_b)


def test_eval_train_eval_without_optimizer_step_refreshes_bn_constants():
    """eval -> train-mode forward (no optimizer step in between: BN recalibration, gradient accumulation with a
    periodic eval) -> eval on the same shape: the second eval must use the UPDATED running statistics, not the cached
    constants of the first eval nor the training batch's scale/shift left in the shared vectors; a second plan (other
    shape) must notice too."""
    from oracle.darknet_ref import forward_ref
    from oracle.step_check import snapshot_state
    model, state = _build(os.path.join(GOLD, 'tiny-pose.cfg'), 55)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, 96, 96, generator=g)
    x2 = torch.rand(1, 3, 128, 128, generator=g)
    xt = torch.rand(2, 3, 96, 96, generator=g) * 3.0 + 1.0          # shifts the running statistics visibly
    model.eval()
    with torch.no_grad():
        y0 = model(x.cuda()).clone()
        z0 = model(x2.cuda()).clone()
        model.train()
        model(xt.cuda())
        model.eval()
        y1 = model(x.cuda()).clone()
        z1 = model(x2.cuda()).clone()
    assert not torch.equal(y0, y1) and not torch.equal(z0, z1)
    st = snapshot_state(model)                                     # running statistics as they are now
    with torch.no_grad():
        ref_y = forward_ref(model.blocks, st, x, training=False)
        ref_z = forward_ref(model.blocks, st, x2, training=False)
    assert rel_err(y1.cpu().numpy(), ref_y.numpy()) < TOL
    assert rel_err(z1.cpu().numpy(), ref_z.numpy()) < TOL


def test_eval_forward_with_autograd_on_takes_the_inference_chain_and_can_still_backprop():
    """valid.py:113 `Variable(data, volatile=True)` is a no-op on current torch: the unchanged callers run eval forwards
    with grad mode on.  That must not allocate / repack the 202 MB data-gradient operands nor skip the fused inference
    launches - and a backward through it (frozen-BN fine-tuning) must still be right."""
    from oracle.darknet_ref import forward_ref
    model, state = _build(os.path.join(GOLD, 'tiny-pose.cfg'), 56)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 96, 96, generator=g)
    probe = torch.randn(2, 20, 3, 3, generator=g)
    model.eval()
    with torch.no_grad():
        y_ng = model(x.cuda()).clone()
    y = model(x.cuda())                       # grad mode on, eval mode
    plan = list(model._plans.values())[0]
    assert y.requires_grad and torch.equal(y.detach(), y_ng)
    assert plan._dpack is None and plan.dgrad_ready is None      # no data-gradient operands were built
    (y * probe.cuda()).sum().backward()
    st = clone_state(state, requires_grad=True)
    ref = forward_ref(model.blocks, st, x, training=False)
    (ref * probe).sum().backward()
    for ind, e in enumerate(st):
        if e is not None:
            assert rel_err(model.models[ind][0].weight.grad.cpu().numpy(), e['weight'].grad.numpy()) < 3e-4, ind
            if 'bn_weight' in e:
                assert rel_err(model.models[ind][1].weight.grad.cpu().numpy(), e['bn_weight'].grad.numpy()) < 3e-4, ind


def test_training_trajectory_tracks_oracle():
    """train.py's inner loop for 4 batches - forward, RegionLoss, backward, torch.optim.SGD.step exactly as train.py:388
    builds it (momentum 0.9, decay * batch, lr / batch), new weights picked up by the next forward - against the oracle
    trained with the same optimizer on the same batches.  Tiny net: every sum is well conditioned, so the two
    trajectories must coincide (loss 1e-5, parameters 1e-5 of their scale) - the full net's first-layer filter
    gradient is not (tests/test_gpu_dropin.py explains and measures that drift)."""
    from oracle.darknet_ref import forward_ref
    from oracle.region_loss_ref import region_loss_ref
    from singleshotpose_amd.region_loss import RegionLoss
    model, state = _build(os.path.join(GOLD, 'tiny-pose.cfg'), 3)
    model.train()
    crit = RegionLoss()
    crit.verbose = False
    B = 8
    kw = dict(lr=1e-3 / B, momentum=0.9, dampening=0, weight_decay=0.0005 * B)
    opt = torch.optim.SGD(model.parameters(), **kw)
    cpu_params = []
    for e in state:
        if e is not None:
            for k in ('weight', 'bias', 'bn_weight', 'bn_bias'):      # module order: conv.weight, conv.bias | bn.weight, bn.bias
                if k in e:
                    cpu_params.append(e[k].requires_grad_(True))
    opt_c = torch.optim.SGD(cpu_params, **kw)
    assert [tuple(p.shape) for p in model.parameters()] == [tuple(q.shape) for q in cpu_params]
    rs = np.random.RandomState(0)
    for step in range(4):
        x = torch.from_numpy(rs.uniform(0, 1, (B, 3, 96, 96)).astype(np.float32))
        tgt = torch.from_numpy(make_targets(rs, B, [1] * B))
        opt.zero_grad()
        loss = crit(model(x.cuda()), tgt, 20 if step % 2 else 0)
        loss.backward()
        opt.step()
        opt_c.zero_grad()
        y = forward_ref(model.blocks, state, x, training=True)
        r = region_loss_ref(y.detach(), tgt, 20 if step % 2 else 0)
        y.backward(r['grad'])
        opt_c.step()
        assert abs(float(loss) - r['loss']) <= 1e-5 * abs(r['loss']), (step, float(loss), r['loss'])
        for p, q in zip(model.parameters(), cpu_params):
            assert rel_err(p.detach().cpu().numpy(), q.detach().numpy()) < 1e-5, step


