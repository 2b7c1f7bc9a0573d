"""The configurations the headline numbers are quoted on, end to end, with the autotuned plans active.

BASELINE.json config 2 (cfg/yolo-pose.cfg, B = 64, 416 x 416, train step) and config 5 (cfg/yolo-pose-multi.cfg, full
trunk, 1-8 labels per image) against the CPU oracle through oracle/step_check.py: head output, RegionLoss, BatchNorm
running statistics <= 1e-4 against an independent oracle forward; every conv launch's raw output <= 1e-4 against the
oracle's convolution of the same inputs; whole-network parameter gradients against the DECISION-FROZEN oracle
backward (strict bar: no fp32-vs-fp64 envelope).  Reference: /root/reference/train.py:76-106, region_loss.py:95-175,
multi_obj_pose_estimation/train_multi.py:60-100, region_loss_multi.py:94-189.
"""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import GOLD, ROOT, load_state_into, make_targets

pytestmark = pytest.mark.gpu
TOL = 1e-4
# Whole-network parameter gradients against the decision-frozen oracle (raw conv outputs, the leaky branch of every
# element and the winner of every pooling window are the product's own - the fused first block included, re-evaluated for
# the checker by ssp_first_conv_raw + the product's BN / leaky kernel: oracle/darknet_ref.py forward_ref(raw_override,
# act_override, pool_override)).
# Bars (round 5; north_star: 1e-4):
#   every parameter gradient <= 7e-5           measured worst 2.2e-5 over the 20 multi-scale shapes and the headline batch
#                                              (profiles/r05_multiscale_parity.txt; round 4: 7.8e-5 / 9.8e-5 at 832 x 832)
#   the first layer's filter gradient <= 7e-5  against the float64 re-evaluation of the oracle's own operands (sum(dx * image)
#                                              with sum(dx) = 0 exactly over an all-positive image cancels ~1e3 : 1 - the fp32
#                                              oracle itself sits ~1e-4 from that sum; the product 0.8e-5 since its workgroup
#                                              partials are summed in float64, ssp_first_bwd_wgrad ABI 4; round 4: bar 5e-4,
#                                              measured 6e-5 ... 3e-4)
#   head <= 5e-5 on the headline batch, <= 7e-5 at the other shapes   (measured 3.9e-5 / <= 5.0e-5: the forward plans are
#                                              admitted under a network-level rounding budget, engine.Plan._apply_head_budget;
#                                              ~2.4e-5 of it is the fp32 ORACLE's own distance from float64)
# History: before the leaky branches were frozen a plan-set-dependent 6.8e-4 showed up on layer 24
# (tools/plansets/r02i_b64_setC.json replays it): ONE element of that layer sits within fp32 rounding of y = 0, takes the
# other leaky branch in the oracle's BatchNorm arithmetic than in the product's (scale * raw + shift), and happens to
# carry most of its channel's gradient.  Both branches are valid fp32 results; the float64 yardstick (below) exposed it.
GRAD_TOL = 7e-5
GRAD_TOL_FIRST_FILTER = 7e-5
HEAD_TOL_HEADLINE = 5e-5
HEAD_TOL = 7e-5


def _report(tag, res):
    from oracle.step_check import summarize
    worst = sorted(res['grad_by_param'].items(), key=lambda kv: -kv[1])[:3]
    print('%s: %s | worst grads %s | fp64-oracle fallbacks %s | plans %s' % (
        tag, summarize(res), worst, res.get('grad_fp64_oracle', {}), [(i, f, d) for i, f, d in res['plans'] if f or d]))


def _assert_step(res, head_tol=HEAD_TOL):
    assert res['head'] < head_tol, res['head']
    assert res['loss'] < TOL, (res['loss_gpu'], res['loss_ref'])
    assert res['running'] < TOL, res['running']
    assert res['conv'] < TOL, res['conv_by_layer']
    assert res['grad_out'] < TOL, res['grad_out']
    for name, err in res['grad_by_param'].items():
        assert err < (GRAD_TOL_FIRST_FILTER if name == '0.weight' else GRAD_TOL), \
            sorted(res['grad_by_param'].items(), key=lambda kv: -kv[1])[:5]


def _assert_exact(res):
    """Against the FLOAT64 evaluation of the raw-output-frozen network (oracle/step_check.py: float64 re-decides the few
    leaky / pool elements that sit within fp32 rounding of a boundary, so it is a yardstick, not an exact value): every
    parameter gradient of the product is either within 1e-4 of it or no further from it than 3x the fp32 oracle
    (PyTorch-CPU / oneDNN) is.  Measured at B = 64 with the leaky branches frozen: every parameter <= 3e-5 on both sides,
    except the first layer's filter gradient - product 3e-4, fp32 oracle 1.5e-3."""
    worst = sorted(res['grad64_by_param'].items(), key=lambda kv: -kv[1][0])[:4]
    print('vs float64 frozen backward: product %.2e, fp32 oracle %.2e | worst (product, oracle) %s' % (
        res['grad64'], res['grad64_ref'], worst))
    for name, (mine, ref) in res['grad64_by_param'].items():
        assert mine <= max(TOL, 3.0 * ref), (name, mine, ref)


def test_headline_config_b64_train_step_with_tuned_plans():
    """Exactly what bench.py times: torch.manual_seed(0) default-initialised cfg/yolo-pose.cfg, batch 64 of synthetic
    416 x 416 images with one label each (bench.synthetic_batch, seed 1000), epoch 20, autotuner on."""
    sys.path.insert(0, ROOT)
    from bench import synthetic_batch
    from oracle.step_check import check_train_step
    from singleshotpose_amd import engine
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.region_loss import RegionLoss
    assert os.environ.get('SSP_AUTOTUNE', '1') != '0'
    torch.manual_seed(0)
    model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg')).cuda()
    x, tgt = synthetic_batch(64, 416, 416, 1000, 'cpu')
    n_rej = len(engine.TUNE_REJECTED)
    res = check_train_step(model, RegionLoss(), x, tgt, 20, exact=True)
    _report('yolo-pose B=64 416', res)
    assert any(f or d for _, f, d in res['plans']), "the autotuner picked no plan: nothing tuned was exercised"
    assert len(engine.TUNE_REJECTED) == n_rej, engine.TUNE_REJECTED[n_rej:]
    _assert_step(res, HEAD_TOL_HEADLINE)
    _assert_exact(res)
    # the float64 yardstick of the forward pass: the product's head is no further from float64 than 2.5x the fp32 oracle's
    # own head is (measured 3.3e-5 against 2.4e-5), and the error budget of the forward plans did its work
    print('head vs float64: product %.2e, fp32 oracle %.2e' % (res['head64'], res['head64_ref']))
    assert res['head64'] <= max(HEAD_TOL_HEADLINE, 2.5 * res['head64_ref']), (res['head64'], res['head64_ref'])
    hb = next(iter(model._plans.values())).head_budget
    assert hb is not None and hb['head_deviation'] <= hb['budget'], hb


def test_headline_config_b8_seeded_weights_pretrain_epoch():
    """The cfg's own batch (batch=8, yolo-pose.cfg:3) with seeded non-trivial BN parameters / running statistics, epoch 0
    (<= pretrain: no confidence term, region_loss.py:157-161) and 1-3 labels per image."""
    from oracle.darknet_ref import seeded_state
    from oracle.step_check import check_train_step
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.region_loss import RegionLoss
    model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
    load_state_into(model, model.blocks, seeded_state(model.blocks, 11))
    model = model.cuda()
    rs = np.random.RandomState(8)
    x = torch.from_numpy(rs.uniform(0, 1, (8, 3, 416, 416)).astype(np.float32))
    tgt = torch.from_numpy(make_targets(rs, 8, [1, 2, 1, 3, 1, 1, 2, 1]))
    res = check_train_step(model, RegionLoss(), x, tgt, 0)
    _report('yolo-pose B=8 416 epoch 0', res)
    _assert_step(res)


def test_multi_object_full_trunk_train_step():
    """BASELINE config 5: cfg/yolo-pose-multi.cfg (full Darknet-19 trunk, head 5 anchors x 32 = 160 channels), B = 8,
    1-8 labels per image with OCCLUSION-style class ids, region_loss_multi with the cfg's anchors."""
    from oracle.darknet_ref import seeded_state
    from oracle.step_check import check_train_step
    from singleshotpose_amd.darknet import DarknetMulti
    from singleshotpose_amd.region_loss import RegionLossMulti
    model = DarknetMulti(os.path.join(ROOT, 'cfg', 'yolo-pose-multi.cfg'))
    load_state_into(model, model.blocks, seeded_state(model.blocks, 19))
    model = model.cuda()
    rs = np.random.RandomState(9)
    B = 8
    x = torch.from_numpy(rs.uniform(0, 1, (B, 3, 416, 416)).astype(np.float32))
    tgt = torch.from_numpy(make_targets(rs, B, [1, 2, 3, 4, 5, 6, 7, 8], multi=True))
    crit = RegionLossMulti(num_keypoints=9, num_classes=13, anchors=model.anchors, num_anchors=5, pretrain_num_epochs=0)
    kw = dict(num_classes=13, num_anchors=5, anchors=model.anchors, pretrain_num_epochs=0, multi=True)
    res = check_train_step(model, crit, x, tgt, 1, loss_kwargs=kw, exact=True)
    _report('yolo-pose-multi B=8 416', res)
    assert tuple(model._plans.keys())[0][:3] == (B, 416, 416)
    _assert_step(res)
    _assert_exact(res)


# Multi-scale training: dataset.py:66-90 draws H = W from {224, 256, ..., 832} (7..26 cells of 32
# pixels) every 10 batches once `seen` passes 10 epochs, and cfg/yolo-pose.cfg:23-24 tests at 672 x 672.  The FULL network
# at the smallest, a mid and the largest training resolution at the cfg's own batch (8), plus one non-416 shape at the
# metric's batch (64): the same decision-frozen step check and the same bars as the 416 x 416 headline test - tuned plans
# on (these shapes pick other tiles / splits / hybrid launches / XCD orders than 416 does: 7 x 7 ... 26 x 26 head grids).
# (288, 7): 9 x 9 head grid - its deepest 3x3 layers tile as 2 x 2 image mosaics, the second one with a phantom image
MULTISCALE = [(224, 8), (608, 8), (832, 8), (352, 64), (288, 7)]


@pytest.mark.parametrize("size,B", MULTISCALE)
def test_multiscale_full_network_train_step(size, B):
    from oracle.darknet_ref import seeded_state
    from oracle.step_check import check_train_step
    from singleshotpose_amd import engine
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.region_loss import RegionLoss
    assert os.environ.get('SSP_AUTOTUNE', '1') != '0'
    model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
    load_state_into(model, model.blocks, seeded_state(model.blocks, 100 + size))
    model = model.cuda()
    rs = np.random.RandomState(size)
    x = torch.from_numpy(rs.uniform(0, 1, (B, 3, size, size)).astype(np.float32))
    tgt = torch.from_numpy(make_targets(rs, B, [1 + (i % 3 == 2) for i in range(B)]))
    n_rej = len(engine.TUNE_REJECTED)
    res = check_train_step(model, RegionLoss(), x, tgt, 20)
    _report('yolo-pose B=%d %dx%d' % (B, size, size), res)
    assert tuple(model._plans.keys())[0][:3] == (B, size, size)
    assert any(f or d for _, f, d in res['plans']), "the autotuner picked no plan: nothing tuned was exercised"
    assert len(engine.TUNE_REJECTED) == n_rej, engine.TUNE_REJECTED[n_rej:]
    _assert_step(res)


def test_checker_reads_the_fused_first_block_in_batch_slices():
    """oracle/step_check.py re-evaluates the fused first block's raw map with ssp_first_conv_raw, whose output offsets are 32-bit
    (2 GiB per call): batch 64 at 608 x 608 needs two calls.  The slicing, forced here to one image per call on a small
    batch: the step check (every conv launch against the oracle, layer 0 included) passes as with one call."""
    from oracle.darknet_ref import seeded_state
    from oracle.step_check import check_train_step
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.region_loss import RegionLoss
    model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
    load_state_into(model, model.blocks, seeded_state(model.blocks, 5))
    model = model.cuda()
    rs = np.random.RandomState(5)
    B, size = 3, 160
    x = torch.from_numpy(rs.uniform(0, 1, (B, 3, size, size)).astype(np.float32))
    tgt = torch.from_numpy(make_targets(rs, B, [1] * B))
    res = check_train_step(model, RegionLoss(), x, tgt, 20, first_raw_slice_bytes=size * size * 32 * 4)
    assert next(iter(model._plans.values())).convs[0].first_live
    _report('sliced first block, B=3 160x160', res)
    _assert_step(res)
    assert res['conv_by_layer'][0] < TOL


def test_multiscale_schedule_one_model_many_shapes():
    """What train.py does after epoch 10: ONE model, a new resolution every few batches, shapes revisited.  Every visit
    of every shape (first visit = plan build + autotune + verify-after-tune, second visit = cached plan) against the
    oracle with the same bars; BatchNorm running statistics carry over from visit to visit as in the reference."""
    from oracle.darknet_ref import seeded_state
    from oracle.step_check import check_train_step
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.region_loss import RegionLoss
    model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
    load_state_into(model, model.blocks, seeded_state(model.blocks, 77))
    model = model.cuda()
    crit = RegionLoss()
    rs = np.random.RandomState(77)
    B = 4
    for visit, size in enumerate((480, 288, 480, 736, 288)):
        x = torch.from_numpy(rs.uniform(0, 1, (B, 3, size, size)).astype(np.float32))
        tgt = torch.from_numpy(make_targets(rs, B, [1] * B))
        res = check_train_step(model, crit, x, tgt, 20)
        _report('visit %d: yolo-pose B=%d %dx%d' % (visit, B, size, size), res)
        _assert_step(res)
    assert sorted(k[1] for k in model._plans.keys()) == [288, 480, 736]


def test_tuned_plans_every_candidate_matches_default_on_bench_shapes():
    """Every plan code the autotuner may hand to a launch (engine.Plan._autotune candidates), on two of the benchmark's
    launch shapes (64 x 13 x 13, 1024 -> 1024 and 64 x 26 x 26, 256 -> 512): identical results to the default plan
    up to fp32 summation order."""
    from gpu_util import dev, stream
    from singleshotpose_amd import _lib
    cands = (12813, 12814, 6414, 6413, 12824, 12834, 306413, 306414, 312813, 312814, 206413, 212814, 6418, 12818, 6438, 306418)
    g = torch.Generator(device='cuda').manual_seed(3)
    for (B, H, W, Cin, Cout, R) in ((64, 13, 13, 1024, 1024, 3), (64, 26, 26, 256, 512, 3)):
        M = B * H * W
        x = torch.empty(M * Cin, device=dev()).uniform_(-1, 1, generator=g)
        w = torch.empty(Cout * R * R * Cin, device=dev()).uniform_(-0.05, 0.05, generator=g)
        outs = {}
        for code in (0,) + cands:
            wsn = max(1, _lib.query('ssp_conv_workspace_floats', B, H, W, Cin, Cout, R, code))
            ws = torch.empty(wsn, device=dev())
            out = torch.zeros(M * Cout, device=dev())
            _lib.call('ssp_conv_fwd', x.data_ptr(), w.data_ptr(), out.data_ptr(), None, None, B, H, W, Cin, Cout, Cin,
                      Cout, R, 0, code, ws.data_ptr(), wsn, stream())
            outs[code] = out
        torch.cuda.synchronize()
        den = float(outs[0].abs().max())
        for code in cands:
            assert float((outs[code] - outs[0]).abs().max()) <= 1e-5 * den, (code, B, H, W, Cin, Cout)


def test_head_budget_holds_over_training_and_is_remeasured():
    """Round-5 review, weak #1: the rounding budget of the forward plans is measured on a plan's FIRST training batch, but the
    amplification it models is a product of BatchNorm gains that training moves.  Full network, cfg batch 8, seeded non-trivial
    BatchNorm parameters: eight optimizer steps (train.py:89-106's zero_grad / forward / loss / backward / step, product SGD),
    then the ninth step is checked against the independent oracle like every full-size step: head <= 7e-5, every gradient
    <= 7e-5.  Then the two re-measurement triggers: a BatchNorm gain pushed 4x (engine.Plan.head_budget_drifted) and
    load_weights (Darknet._load_blocks -> head_budget_stale) both make the next training forward measure again."""
    from oracle.darknet_ref import seeded_state
    from oracle.step_check import check_train_step
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.optim import SGD
    from singleshotpose_amd.region_loss import RegionLoss
    model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
    load_state_into(model, model.blocks, seeded_state(model.blocks, 23))
    model = model.cuda().train()
    crit = RegionLoss()
    crit.verbose = False
    opt = SGD(model.parameters(), lr=1e-3 / 8, momentum=0.9, weight_decay=5e-4 * 8)
    rs = np.random.RandomState(5)
    for it in range(8):
        x = torch.from_numpy(rs.uniform(0, 1, (8, 3, 416, 416)).astype(np.float32)).cuda()
        tgt = torch.from_numpy(make_targets(rs, 8, [1] * 8))
        opt.zero_grad()
        loss = crit(model(x), tgt, 20)
        loss.backward()
        opt.step()
    plan = next(iter(model._plans.values()))
    hb0 = plan.head_budget
    assert hb0 is not None and plan._head_budget_done
    x = torch.from_numpy(rs.uniform(0, 1, (8, 3, 416, 416)).astype(np.float32))
    tgt = torch.from_numpy(make_targets(rs, 8, [1, 2, 1, 3, 1, 1, 2, 1]))
    res = check_train_step(model, RegionLoss(), x, tgt, 20)
    _report('yolo-pose B=8 416 after 8 SGD steps', res)
    _assert_step(res)
    # trigger 1: a BatchNorm block's gain moves by more than 2x
    assert not plan.head_budget_drifted()
    with torch.no_grad():
        model.models[4][1].weight.mul_(4.0)
    model(x.cuda()).sum().backward()                  # (the plan's scale vectors follow the parameters at the next forward)
    assert plan.head_budget_drifted()
    os.environ['SSP_HEAD_BUDGET_EVERY'] = '1'
    try:
        model.zero_grad()
        model(x.cuda()).sum().backward()
    finally:
        del os.environ['SSP_HEAD_BUDGET_EVERY']
    hb1 = plan.head_budget
    assert hb1 is not hb0 and plan._head_budget_done and not plan.head_budget_drifted()
    print('re-measured after the gain moved: layer 4 candidates %s -> %s' % (hb0['table'].get(4), hb1['table'].get(4)))
    # trigger 2: load_weights
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        wf = os.path.join(d, 'w.weights')
        model.save_weights(wf)
        model.load_weights(wf)
    assert not plan._head_budget_done
    model.zero_grad()
    model(x.cuda()).sum().backward()
    assert plan._head_budget_done and plan.head_budget is not hb1


def test_backward_winograd_choices_stay_under_the_gradient_bar_when_forced():
    """Round-5 review, weak #1 (second half): the data- and filter-gradient Winograd choices are under no budget of their own.
    Worst case made explicit: F(4x4) FORCED on the data gradient and the filter gradient of the two 104 x 104 layers (4, 6) -
    the noisiest admissible choice on the layers whose rounding is amplified most - and every parameter gradient of the
    headline network (B = 8) must still meet the 7e-5 bar against the decision-frozen oracle."""
    from oracle.darknet_ref import seeded_state
    from oracle.step_check import check_train_step
    from singleshotpose_amd import _lib, engine
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.region_loss import RegionLoss
    model = Darknet(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
    load_state_into(model, model.blocks, seeded_state(model.blocks, 29))
    model = model.cuda()
    rs = np.random.RandomState(9)
    x = torch.from_numpy(rs.uniform(0, 1, (8, 3, 416, 416)).astype(np.float32))
    tgt = torch.from_numpy(make_targets(rs, 8, [1, 2, 1, 3, 1, 1, 2, 1]))
    # first step: the plan tunes itself; then force the choices and check a second step
    model.train()
    RegionLoss.verbose = False
    crit = RegionLoss()
    crit.verbose = False
    crit(model(x.cuda()), tgt, 20).backward()
    plan = next(iter(model._plans.values()))
    for i in (4, 6):
        cs = plan.convs[i]
        cs.plan_dgrad = engine.WINO4 + 6413
        cs.wgrad_wino = 4
        cs.wino_ws_floats = _lib.query('ssp_conv_wgrad_wino_workspace_floats_t', plan.B, cs.H, cs.W, cs.cinp, cs.cout, 4)
        cs.wino_ws = None
        cs.ws_dgrad = _lib.query('ssp_conv_workspace_floats', plan.B, cs.H, cs.W, cs.coutp, cs.cin, cs.k, cs.plan_dgrad)
    plan._plan_bn_fusion()
    plan._fit_workspace()
    res = check_train_step(model, crit, x, tgt, 20)
    _report('yolo-pose B=8 416, F(4x4) forced on dgrad / wgrad of layers 4 and 6', res)
    assert [(i, d) for i, f, d in res['plans'] if i in (4, 6)] == [(4, engine.WINO4 + 6413), (6, engine.WINO4 + 6413)]
    _assert_step(res)
