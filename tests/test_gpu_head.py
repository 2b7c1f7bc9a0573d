"""RegionLoss / get_region_boxes / PnP on the GPU against the oracle and the reference-generated golden vectors."""
import numpy as np
import pytest
import torch

from helpers import gold, make_targets, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _gpu_loss(mod, out, tgt, epoch):
    o = torch.from_numpy(out).cuda().requires_grad_(True)
    loss = mod(o, torch.from_numpy(tgt), epoch)
    loss.backward()
    s = mod.last_stats().cpu().numpy()
    return float(loss), o.grad.cpu().numpy(), s


def test_region_loss_single_golden(capsys):
    from singleshotpose_amd.region_loss import RegionLoss
    g = gold('region_single.npz')
    for epoch in (20, 0):
        mod = RegionLoss()
        loss, grad, s = _gpu_loss(mod, g['output'], g['target'], epoch)
        ref = float(g['loss_e%d' % epoch])
        assert abs(loss - ref) <= TOL * abs(ref)
        assert rel_err(grad, g['grad_e%d' % epoch]) < TOL
        line = capsys.readouterr().out.strip().splitlines()[-1]
        ref_line = str(g['line_e%d' % epoch])
        # same integers in the status line, floats within print precision
        assert line.split('loss:')[0] == ref_line.split('loss:')[0]
    mod = RegionLoss()
    mod.noobject_scale, mod.coord_scale = 0.1, 2.0
    loss, grad, s = _gpu_loss(mod, g['output'], g['target'].astype(np.float32), 20)
    assert abs(loss - float(g['loss_f32'])) <= TOL * abs(float(g['loss_f32']))
    assert rel_err(grad, g['grad_f32']) < TOL


def test_region_loss_multi_golden(capsys):
    from singleshotpose_amd.region_loss import RegionLossMulti
    g = gold('region_multi.npz')
    anchors = [float(a) for a in g['anchors']]
    for epoch in (20, 0):
        mod = RegionLossMulti(num_keypoints=9, num_classes=13, anchors=anchors, num_anchors=5, pretrain_num_epochs=15)
        loss, grad, s = _gpu_loss(mod, g['output'], g['target'], epoch)
        ref = float(g['loss_e%d' % epoch])
        assert abs(loss - ref) <= TOL * abs(ref)
        assert rel_err(grad, g['grad_e%d' % epoch]) < TOL
        line = capsys.readouterr().out.strip().splitlines()[-1]
        assert line.split('loss:')[0] == str(g['line_e%d' % epoch]).split('loss:')[0]


@pytest.mark.parametrize("nB,grid,ngt", [(8, 13, [1] * 8), (3, 21, [0, 2, 50]), (64, 13, None), (2, 7, [5, 1])])
def test_region_loss_single_vs_oracle(nB, grid, ngt):
    """Edge cases the reference cannot run itself (0 or many labels per image) checked against the oracle."""
    from oracle.region_loss_ref import region_loss_ref
    from singleshotpose_amd.region_loss import RegionLoss
    rs = np.random.RandomState(nB * 100 + grid)
    if ngt is None:
        ngt = [int(v) for v in rs.randint(0, 4, nB)]
    out = (rs.standard_normal((nB, 20, grid, grid)) * 0.7).astype(np.float32)
    tgt = make_targets(rs, nB, ngt)
    # two labels in the same cell (later one wins) and a centroid at 0.999 (last cell)
    t3 = tgt.reshape(nB, 50, 21)
    if ngt[1] >= 2:
        t3[1, 1, 1:3] = t3[1, 0, 1:3] + 1e-3
    if ngt[-1] >= 1:
        t3[-1, 0, 1:3] = 0.999
    mod = RegionLoss()
    mod.verbose = False
    for epoch in (16, 15):
        loss, grad, s = _gpu_loss(mod, out, tgt, epoch)
        r = region_loss_ref(torch.from_numpy(out), torch.from_numpy(tgt), epoch)
        assert abs(loss - r['loss']) <= TOL * abs(r['loss'])
        assert rel_err(grad, r['grad'].numpy()) < TOL
        assert (int(s[5]), int(s[6]), int(s[7])) == (r['nGT'], r['nCorrect'], r['nProposals'])


def test_region_loss_multi_vs_oracle():
    from oracle.region_loss_ref import region_loss_ref
    from singleshotpose_amd.region_loss import RegionLossMulti
    anchors = [1.4820, 2.2412, 2.0501, 3.1265, 2.3946, 4.6891, 3.1018, 3.9910, 3.4879, 5.8851]
    rs = np.random.RandomState(11)
    nB = 6
    ngt = [1, 8, 0, 3, 2, 5]
    out = (rs.standard_normal((nB, 160, 13, 13)) * 0.7).astype(np.float32)
    tgt = make_targets(rs, nB, ngt, multi=True)
    mod = RegionLossMulti(anchors=anchors)
    mod.verbose = False
    loss, grad, s = _gpu_loss(mod, out, tgt, 20)
    r = region_loss_ref(torch.from_numpy(out), torch.from_numpy(tgt), 20, num_classes=13, num_anchors=5, anchors=anchors, multi=True)
    assert abs(loss - r['loss']) <= TOL * abs(r['loss'])
    assert rel_err(grad, r['grad'].numpy()) < TOL
    assert (int(s[5]), int(s[6]), int(s[7])) == (r['nGT'], r['nCorrect'], r['nProposals'])


def test_get_region_boxes_golden():
    from singleshotpose_amd.utils import get_region_boxes, region_boxes_batched
    g = gold('decode.npz')
    for name in ('a', 'b'):
        box = get_region_boxes(torch.from_numpy(g['out_' + name]).cuda(), 1, 9)
        assert len(box) == 21
        np.testing.assert_allclose(np.array([float(v) for v in box]), g['box_' + name], rtol=1e-5, atol=1e-6)
        assert np.reshape(np.array([float(v) for v in box[:18]]), [-1, 2]).shape == (9, 2)
    # ties: first maximum in (b, cy, cx) order
    o = torch.zeros(2, 20, 5, 5)
    o[:, 18] = -3.0
    o[1, 18, 2, 3] = 4.0
    o[0, 18, 4, 1] = 4.0
    o[0, 18, 1, 2] = 4.0
    box = get_region_boxes(o.cuda(), 1, 9)
    assert abs(float(box[0]) - (0.5 + 2) / 5) < 1e-6 and abs(float(box[1]) - (0.5 + 1) / 5) < 1e-6
    per = region_boxes_batched(o.cuda(), 1, 9).cpu()
    assert abs(float(per[1, 0]) - (0.5 + 3) / 5) < 1e-6


def test_get_multi_region_boxes_golden():
    from singleshotpose_amd.utils_multi import bbox_iou, get_multi_region_boxes
    g = gold('decode_multi.npz')
    anchors = [float(a) for a in g['anchors']]
    out = torch.from_numpy(g['output']).cuda()
    for corr in (4, 7):
        boxes = get_multi_region_boxes(out, 0.05, 13, 9, anchors, 5, corr, only_objectness=0)
        assert len(boxes) == 2
        for b, bl in enumerate(boxes):
            ref = g['boxes_c%d_b%d' % (corr, b)]
            assert len(bl) == ref.shape[0]
            np.testing.assert_allclose(np.array(bl, dtype=np.float64), ref, rtol=1e-4, atol=1e-6)
    assert abs(bbox_iou([0, 0, 2, 2], [0, 0, 1, 3]) - 0.4) < 1e-12


def test_pnp_round_trip_and_oracle():
    from oracle.pnp_ref import project, rodrigues, solve_pnp_ref
    from singleshotpose_amd.utils import pnp, pnp_batched
    K = np.array([[572.4114, 0, 325.2611], [0, 573.5704, 242.0489], [0, 0, 1.0]])
    hx, hy, hz = 0.038, 0.039, 0.046
    corners = np.array([[sx * hx, sy * hy, sz * hz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
    X = np.concatenate([np.zeros((1, 3)), corners], 0)
    rs = np.random.RandomState(0)
    n = 64
    Rs, ts, uvs, uvn = [], [], [], []
    for i in range(n):
        axis = rs.standard_normal(3)
        axis /= np.linalg.norm(axis)
        R = rodrigues(axis * rs.uniform(0, np.pi / 3))
        t = np.array([rs.uniform(-.1, .1), rs.uniform(-.1, .1), rs.uniform(0.6, 1.2)])
        uv = project(X, R, t, K)
        Rs.append(R); ts.append(t); uvs.append(uv); uvn.append(uv + rs.uniform(-1, 1, uv.shape))
    Xb = np.broadcast_to(X, (n, 9, 3))
    # noise-free: recover the pose to ~1e-6 px
    R_g, t_g = pnp_batched(Xb, np.stack(uvs), K)
    for i in range(n):
        assert np.abs(project(X, R_g[i], t_g[i], K) - uvs[i]).max() < 1e-6
        assert np.abs(R_g[i] - Rs[i]).max() < 1e-6 and np.abs(t_g[i].ravel() - ts[i]).max() < 1e-6
    # 1 px noise: same least-squares pose as the numpy restatement (reprojection within 1e-3 px)
    R_g, t_g = pnp_batched(Xb, np.stack(uvn), K)
    for i in range(n):
        R_o, t_o = solve_pnp_ref(X, uvn[i], K)
        assert np.abs(project(X, R_g[i], t_g[i], K) - project(X, R_o, t_o, K)).max() < 1e-3
        assert abs(np.linalg.det(R_g[i]) - 1) < 1e-9 and np.abs(R_g[i].dot(R_g[i].T) - np.eye(3)).max() < 1e-9
    # reference signature: float32 inputs, (3,3)/(3,1) float64 outputs
    R1, t1 = pnp(X.astype(np.float32), uvs[0].astype(np.float32), K.astype(np.float32))
    assert R1.shape == (3, 3) and t1.shape == (3, 1) and R1.dtype == np.float64
    assert np.abs(project(X, R1, t1, K) - uvs[0]).max() < 1e-2
