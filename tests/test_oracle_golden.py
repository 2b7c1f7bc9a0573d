"""The oracle (oracle/*.py, CPU) pinned against golden vectors produced by the real reference code
(oracle/gen_golden.py).  CPU only."""
import numpy as np
import torch

from helpers import GOLD, clone_state, gold, golden_input, rel_err
from oracle.darknet_ref import forward_ref, reorg_ref, seeded_state
from oracle.region_loss_ref import get_multi_region_boxes_ref, get_region_boxes_ref, region_loss_ref
from singleshotpose_amd.cfg import parse_cfg

import os

ROOT = os.path.dirname(GOLD.rstrip('/')).rsplit('/tests', 1)[0]


def _status(line):
    # "0: nGT 4, recall 4, proposals 667, loss: x .., y .., conf .., [cls ..,] total .."
    head, tail = line.split('loss:')
    nums = head.replace(':', ',').split(',')
    ngt = int(nums[1].split()[1]); rec = int(nums[2].split()[1]); prop = int(nums[3].split()[1])
    vals = {kv.split()[0]: float(kv.split()[1]) for kv in tail.split(',')}
    return ngt, rec, prop, vals


def test_region_single_matches_reference():
    g = gold('region_single.npz')
    out, tgt = torch.from_numpy(g['output']), torch.from_numpy(g['target'])
    for epoch in (20, 0):
        r = region_loss_ref(out, tgt, epoch)
        assert abs(r['loss'] - float(g['loss_e%d' % epoch])) <= 1e-5 * abs(float(g['loss_e%d' % epoch]))
        assert rel_err(r['grad'].numpy(), g['grad_e%d' % epoch]) < 1e-5
        ngt, rec, prop, vals = _status(str(g['line_e%d' % epoch]))
        assert (r['nGT'], r['nCorrect'], r['nProposals']) == (ngt, rec, prop)
        assert abs(r['loss_x'] - vals['x']) < 1e-5 and abs(r['loss_conf'] - vals['conf']) < 1e-3
    r = region_loss_ref(out, torch.from_numpy(g['target'].astype(np.float32)), 20, noobject_scale=0.1, coord_scale=2.0)
    assert abs(r['loss'] - float(g['loss_f32'])) <= 1e-5 * abs(float(g['loss_f32']))
    assert rel_err(r['grad'].numpy(), g['grad_f32']) < 1e-5


def test_region_multi_matches_reference():
    g = gold('region_multi.npz')
    out, tgt = torch.from_numpy(g['output']), torch.from_numpy(g['target'])
    anchors = [float(a) for a in g['anchors']]
    for epoch in (20, 0):
        r = region_loss_ref(out, tgt, epoch, num_classes=13, num_anchors=5, anchors=anchors, multi=True)
        ref = float(g['loss_e%d' % epoch])
        assert abs(r['loss'] - ref) <= 2e-5 * abs(ref)
        assert rel_err(r['grad'].numpy(), g['grad_e%d' % epoch]) < 1e-5
        ngt, rec, prop, vals = _status(str(g['line_e%d' % epoch]))
        assert (r['nGT'], r['nCorrect'], r['nProposals']) == (ngt, rec, prop)
        assert abs(r['loss_cls'] - vals['cls']) < 1e-3


def test_decode_matches_reference():
    g = gold('decode.npz')
    for name in ('a', 'b'):
        box = get_region_boxes_ref(torch.from_numpy(g['out_' + name]), 1, 9)
        np.testing.assert_allclose(np.array(box, dtype=np.float64), g['box_' + name], rtol=1e-6, atol=1e-7)


def test_decode_multi_matches_reference():
    g = gold('decode_multi.npz')
    out = torch.from_numpy(g['output'])
    for corr in (4, 7):
        boxes = get_multi_region_boxes_ref(out, 0.05, 13, 9, 5, corr, only_objectness=0)
        for b, bl in enumerate(boxes):
            ref = g['boxes_c%d_b%d' % (corr, b)]
            assert len(bl) == ref.shape[0]
            np.testing.assert_allclose(np.array(bl, dtype=np.float64), ref, rtol=1e-5, atol=1e-6)
    assert g['boxes_c7_b0'][-1, 20] == 7      # the fallback path is exercised


def test_reorg_bit_exact():
    g = gold('reorg.npz')
    y = reorg_ref(torch.from_numpy(g['x']), 2).numpy()
    assert np.array_equal(y, g['y'])


def _check_net(cfgfile, tag, B, H, W, seed, with_grad):
    g = gold('darknet_%s.npz' % tag)
    blocks = parse_cfg(cfgfile)
    state = seeded_state(blocks, seed)
    x = torch.from_numpy(golden_input(g, B, H, W))
    with torch.no_grad():
        y = forward_ref(blocks, clone_state(state), x, training=False)
    assert rel_err(y.numpy(), g['y_eval']) < 1e-5
    if with_grad:
        st = clone_state(state, requires_grad=True)
        y = forward_ref(blocks, st, x, training=True)
        assert rel_err(y.detach().numpy(), g['y_train']) < 1e-5
        (y * torch.from_numpy(g['probe'])).sum().backward()
        # reference parameter names: models.<ind>.conv<k>.weight / bn<k>.weight|bias
        conv_id = 0
        for ind, b in enumerate(blocks[1:]):
            if b['type'] != 'convolutional':
                continue
            conv_id += 1
            e = st[ind]
            names = {'weight': 'models.%d.conv%d.weight' % (ind, conv_id)}
            if 'bn_weight' in e:
                names['bn_weight'] = 'models.%d.bn%d.weight' % (ind, conv_id)
                names['bn_bias'] = 'models.%d.bn%d.bias' % (ind, conv_id)
                np.testing.assert_allclose(e['running_mean'].numpy(), g['buf/models.%d.bn%d.running_mean' % (ind, conv_id)], rtol=1e-4, atol=1e-6)
                np.testing.assert_allclose(e['running_var'].numpy(), g['buf/models.%d.bn%d.running_var' % (ind, conv_id)], rtol=1e-4, atol=1e-6)
            else:
                names['bias'] = 'models.%d.conv%d.bias' % (ind, conv_id)
            for k, n in names.items():
                gr = e[k].grad.numpy()
                ref_norm = float(g['gnorm/' + n][0])
                assert abs(np.sqrt((gr.astype(np.float64) ** 2).sum()) - ref_norm) <= 2e-4 * ref_norm + 1e-7, n
                if 'grad/' + n in g.files:
                    assert rel_err(gr, g['grad/' + n]) < 2e-4, n


def test_darknet_tiny_matches_reference():
    _check_net(os.path.join(GOLD, 'tiny-pose.cfg'), 'tiny', 2, 96, 96, 5, True)


def test_darknet_full_eval_matches_reference(root_dir):
    _check_net(os.path.join(root_dir, 'cfg', 'yolo-pose.cfg'), 'full_eval', 1, 416, 416, 6, False)


def test_darknet_full_train_matches_reference(root_dir):
    _check_net(os.path.join(root_dir, 'cfg', 'yolo-pose.cfg'), 'full_train', 2, 416, 416, 7, True)


def test_eval_metrics_oracle_vs_reference_golden():
    """oracle.eval_ref against the reference's own utils.py / valid.py:146-172 results (tests/golden/eval_metrics.npz)."""
    from oracle.eval_ref import pose_errors_ref, pts_diameter_ref, synthetic_eval_case
    g = np.load(os.path.join(GOLD, 'eval_metrics.npz'))
    for seed, nv in ((0, 700), (1, 257)):
        pts, K, R_gt, t_gt, R_pr, t_pr = synthetic_eval_case(seed, n_pose=6, n_vert=nv)
        vertices = np.concatenate((pts.T, np.ones((1, nv))), axis=0)
        got = np.array([pose_errors_ref(vertices, R_gt[i], t_gt[i], R_pr[i], t_pr[i], K) for i in range(6)])
        assert np.array_equal(got, g['errors_%d' % seed])
        assert pts_diameter_ref(pts) == float(g['diameter_%d' % seed][0])


def test_decision_overrides_are_identities_on_own_decisions_and_local_when_flipped():
    """forward_ref(act_override / pool_override) - the decision freezing oracle/step_check.py uses: fed the oracle's OWN
    activations and pool winners it changes nothing (values and gradients bit-identical); with one element's leaky branch
    flipped only that layer's channel moves."""
    import torch.nn.functional as F
    from oracle.darknet_ref import forward_ref, seeded_state
    from singleshotpose_amd.cfg import parse_cfg
    blocks = parse_cfg(os.path.join(GOLD, 'tiny-pose.cfg'))
    base = seeded_state(blocks, 4)

    def fresh():
        return [None if e is None else {k: v.clone().requires_grad_(not k.startswith('running')) for k, v in e.items()}
                for e in base]
    x = torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(1))
    st0 = fresh()
    y0, outs = forward_ref(blocks, st0, x, training=True, keep=True)
    y0.sum().backward()
    acts, pools = {}, {}
    for ind, b in enumerate(blocks[1:]):
        if b['type'] == 'convolutional' and b['activation'] == 'leaky':
            acts[ind] = outs[ind].detach()
        if b['type'] == 'maxpool':
            pools[ind] = F.max_pool2d(outs[ind - 1].detach(), 2, 2, return_indices=True)[1]
    assert acts and pools
    st1 = fresh()
    y1 = forward_ref(blocks, st1, x, training=True, act_override=acts, pool_override=pools)
    y1.sum().backward()
    assert torch.equal(y0, y1)
    for a, b in zip(st0, st1):
        if a is not None:
            for k in a:
                if not k.startswith('running'):
                    assert torch.equal(a[k].grad, b[k].grad), k
    # flip one element of the last leaky block: its filter gradient moves in that output channel only
    last = max(acts)
    flipped = dict(acts)
    t = acts[last].clone()
    t[0, 3, 1, 1] = -t[0, 3, 1, 1] if float(t[0, 3, 1, 1]) != 0 else 1.0
    flipped[last] = t
    st2 = fresh()
    forward_ref(blocks, st2, x, training=True, act_override=flipped, pool_override=pools).sum().backward()
    d = (st2[last]['weight'].grad - st0[last]['weight'].grad).abs().flatten(1).max(1)[0]
    assert float(d[3]) > 0 and float(d.sum() - d[3]) == 0
