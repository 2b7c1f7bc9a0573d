"""CPU-only checks: C-ABI surface, cfg / weights formats, host-side mirror of the reference interface, helpers."""
import io
import os
import re
import contextlib
import ctypes

import numpy as np
import pytest
import torch

from helpers import GOLD, ROOT


def test_library_exports_every_declared_symbol():
    from singleshotpose_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'ssp_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(ssp_[a-z0-9_]+)\s*\(', header)))
    assert len(declared) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)          # dlopen only: no GPU call is made on the CPU box
    for name in declared:
        assert hasattr(lib, name), "libssp_hip.so does not export %s" % name
    # the ctypes table covers the whole header and nothing else
    assert sorted(_lib.exported_symbols()) == declared
    assert _lib.query('ssp_abi_version') == 5
    with pytest.raises(_lib.SspError, match="unknown option"):
        _lib.call('ssp_set_option', b'no_such_knob', 1)


def test_parse_cfg_and_shapes():
    from singleshotpose_amd.cfg import layer_shapes, parse_cfg, print_cfg
    blocks = parse_cfg(os.path.join(ROOT, 'cfg', 'yolo-pose.cfg'))
    assert len(blocks) == 33 and blocks[0]['type'] == 'net' and blocks[-1]['type'] == 'region'
    assert blocks[0]['steps'] == '-1,80,160' and blocks[0]['test_width'] == '672'    # values stay strings
    assert blocks[1]['batch_normalize'] == '1' and blocks[31]['batch_normalize'] == 0  # default for the head conv
    assert blocks[-1]['anchors'] == '' and blocks[-1]['classes'] == '1'
    shapes = layer_shapes(blocks)
    assert shapes[0] == (416, 416, 32) and shapes[16] == (26, 26, 512)
    assert shapes[27] == (13, 13, 256) and shapes[28] == (13, 13, 1280) and shapes[30] == (13, 13, 20)
    assert layer_shapes(blocks, 672, 672)[30] == (21, 21, 20)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        print_cfg(blocks)
    lines = buf.getvalue().splitlines()
    assert lines[1] == '    0 conv     32  3 x 3 / 1   416 x 416 x   3   ->   416 x 416 x  32'
    assert lines[2] == '    1 max          2 x 2 / 2   416 x 416 x  32   ->   208 x 208 x  32'
    assert lines[26] == '   25 route  16'
    assert lines[28] == '   27 reorg              / 2    26 x  26 x  64   ->    13 x  13 x 256'
    assert lines[29] == '   28 route  27 24'
    assert lines[31] == '   30 conv     20  1 x 1 / 1    13 x  13 x1024   ->    13 x  13 x  20'
    assert lines[32] == '   31 detection'
    multi = parse_cfg(os.path.join(ROOT, 'cfg', 'yolo-pose-multi.cfg'))
    assert multi[31]['filters'] == '160' and multi[-1]['num'] == '5' and multi[-1]['classes'] == '13'
    assert len(multi[-1]['anchors'].split(',')) == 10


def test_module_tree_and_weight_round_trip(tmp_path):
    from oracle.darknet_ref import seeded_state, write_weights
    from singleshotpose_amd.darknet import Darknet
    cfg = os.path.join(GOLD, 'tiny-pose.cfg')
    m = Darknet(cfg)
    names = [n for n, _ in m.named_parameters()]
    assert names[:3] == ['models.0.conv1.weight', 'models.0.bn1.weight', 'models.0.bn1.bias']
    assert names[-2:] == ['models.19.conv12.weight', 'models.19.conv12.bias']
    assert all(('.bn' in n) or ('.bias' in n) or ('.conv' in n) for n in names)     # train.py:384 filters on these
    assert m.models[0][0].weight.shape == (8, 3, 3, 3) and m.models[0][1].eps == 1e-4
    assert (m.width, m.height, m.test_width, m.num_keypoints, m.num_anchors, m.num_classes) == (96, 96, 160, 9, 1, 1)
    assert m.anchors == [] and m.loss.noobject_scale == 0.1 and m.loss.object_scale == 5.0   # cfg-configured region loss

    state = seeded_state(m.blocks, 31)
    ref_file = str(tmp_path / 'ref.weights')
    write_weights(ref_file, m.blocks, state, seen=77)        # oracle writer = the reference's stream order
    m.load_weights(ref_file)
    assert int(m.seen) == 77
    assert torch.equal(m.models[4][0].weight, state[4]['weight'])
    assert torch.equal(m.models[4][1].running_var, state[4]['running_var'])
    assert torch.equal(m.models[19][0].bias, state[19]['bias'])
    out_file = str(tmp_path / 'out.weights')
    m.save_weights(out_file)
    assert open(out_file, 'rb').read() == open(ref_file, 'rb').read()    # byte-exact round trip

    # load_weights_until_last leaves the head conv (and everything after it) untouched (darknet.py:310)
    m2 = Darknet(cfg)
    head_before = m2.models[19][0].weight.clone()
    m2.load_weights_until_last(ref_file)
    assert torch.equal(m2.models[18][0].weight, state[18]['weight'])
    assert torch.equal(m2.models[19][0].weight, head_before)
    # cutoff: only the first blocks are written
    m.save_weights(str(tmp_path / 'cut.weights'), cutoff=3)
    assert os.path.getsize(str(tmp_path / 'cut.weights')) == 16 + 4 * (4 * 8 + 8 * 27 + 4 * 16 + 16 * 8 * 9)


def test_no_cpu_fallback():
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.region_loss import RegionLoss
    from singleshotpose_amd.utils import get_region_boxes
    m = Darknet(os.path.join(GOLD, 'tiny-pose.cfg'))
    with pytest.raises(RuntimeError, match="HIP"):
        m(torch.zeros(1, 3, 96, 96))
    with pytest.raises(RuntimeError, match="HIP"):
        RegionLoss()(torch.zeros(1, 20, 3, 3), torch.zeros(1, 50 * 21), 0)
    with pytest.raises(RuntimeError, match="HIP"):
        get_region_boxes(torch.zeros(1, 20, 3, 3), 1, 9)
    with pytest.raises(RuntimeError):
        m.models[16](torch.zeros(1, 8, 6, 6))        # Reorg has no standalone eager path either
    # nothing in the product package imports the oracle
    pkg = os.path.join(ROOT, 'singleshotpose_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            assert 'oracle' not in open(os.path.join(pkg, fn)).read().replace('the oracle', '').replace('oracle/', ''), fn


def test_unsupported_blocks_fail_loudly(tmp_path):
    from singleshotpose_amd.darknet import Darknet
    from singleshotpose_amd.engine import Plan
    cfg = tmp_path / 'bad.cfg'
    cfg.write_text('[net]\nheight=32\nwidth=32\nchannels=3\n\n[convolutional]\nfilters=8\nsize=3\nstride=2\npad=1\nactivation=leaky\n')
    m = Darknet(str(cfg))
    with pytest.raises(NotImplementedError):
        Plan(m, 1, 32, 32, torch.device('cpu'))


def test_utils_helpers(tmp_path):
    from singleshotpose_amd import utils as U
    rs = np.random.RandomState(0)
    verts = np.concatenate([rs.uniform(-0.05, 0.05, (3, 200)), np.ones((1, 200))], 0)
    c = U.get_3D_corners(verts)
    assert c.shape == (4, 8)
    mn, mx = verts[:3].min(1), verts[:3].max(1)
    np.testing.assert_allclose(c[:3, 0], mn)
    np.testing.assert_allclose(c[:3, 1], [mn[0], mn[1], mx[2]])
    np.testing.assert_allclose(c[:3, 4], [mx[0], mn[1], mn[2]])
    np.testing.assert_allclose(c[:3, 7], mx)
    K = U.get_camera_intrinsic(325.2611, 242.0489, 572.4114, 573.5704)
    Rt = np.concatenate([np.eye(3), np.array([[0.01], [0.02], [0.9]])], 1)
    proj = U.compute_projection(verts, Rt, K)
    assert proj.shape == (2, 200) and proj.dtype == np.float32
    p = K.dot(Rt.dot(verts))
    np.testing.assert_allclose(proj, (p[:2] / p[2]).astype(np.float32), rtol=1e-6)
    pts = verts[:3].T
    brute = max(np.linalg.norm(pts[i] - pts[j]) for i in range(len(pts)) for j in range(i, len(pts)))
    assert abs(U.calc_pts_diameter(pts) - brute) < 1e-9
    assert abs(U.calcAngularDistance(np.eye(3), np.eye(3))) < 1e-6
    d = tmp_path / 'x.data'
    d.write_text('train  = a/train.txt\nvalid = a/test.txt\n\nfx = 572.4114 \n')
    o = U.read_data_cfg(str(d))
    assert o['gpus'] == '0' and o['num_workers'] == '10' and o['train'] == 'a/train.txt' and o['fx'] == '572.4114'
    lab = tmp_path / 'l.txt'
    lab.write_text(' '.join(str(i / 100.0) for i in range(21)) + '\n' + ' '.join(str(i / 50.0) for i in range(21)) + '\n')
    t = U.read_truths_args(str(lab))
    assert t.shape == (38,) and abs(t[19] - 0.0) < 1e-12 and abs(t[18] - 0.18) < 1e-12   # 19 numbers per row, packed
    g = np.arange(18, dtype=np.float32).reshape(9, 2)
    f = U.fix_corner_order(g)
    assert f[2, 0] == g[3, 0] and f[5, 0] == g[2, 0] and f[8, 0] == g[8, 0]
    pr = torch.rand(18, 7)
    gt = torch.rand(18, 1).repeat(1, 7)
    from oracle.region_loss_ref import corner_confidence_ref, corner_confidences_ref
    np.testing.assert_allclose(U.corner_confidences(gt, pr).numpy(), corner_confidences_ref(gt, pr).numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(float(U.corner_confidence(list(gt[:, 0]), pr[:, 0])), float(corner_confidence_ref(gt[:, 0], pr[:, 0])), rtol=1e-5, atol=1e-7)


def test_pnp_oracle_round_trip():
    """The PnP restatement (oracle/pnp_ref.py, parity unpinned - no cv2 here) recovers synthetic poses."""
    from oracle.pnp_ref import project, rodrigues, solve_pnp_ref
    K = np.array([[572.4114, 0, 325.2611], [0, 573.5704, 242.0489], [0, 0, 1.0]])
    X = np.concatenate([np.zeros((1, 3)), np.array([[sx * .038, sy * .039, sz * .046] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])], 0)
    rs = np.random.RandomState(1)
    for _ in range(20):
        axis = rs.standard_normal(3)
        R = rodrigues(axis / np.linalg.norm(axis) * rs.uniform(0, np.pi / 3))
        t = np.array([rs.uniform(-.1, .1), rs.uniform(-.1, .1), rs.uniform(.6, 1.2)])
        uv = project(X, R, t, K)
        R2, t2 = solve_pnp_ref(X, uv, K)
        assert np.abs(project(X, R2, t2, K) - uv).max() < 1e-6
        assert np.abs(R2 - R).max() < 1e-6 and np.abs(t2.ravel() - t).max() < 1e-6
        # 1 px noise: the LM result is a stationary point of the reprojection error
        uvn = uv + rs.uniform(-1, 1, uv.shape)
        R3, t3 = solve_pnp_ref(X, uvn, K)
        e0 = ((project(X, R3, t3, K) - uvn) ** 2).sum()
        for d in range(3):
            for s in (-1e-4, 1e-4):
                tt = t3.ravel().copy()
                tt[d] += s
                assert ((project(X, R3, tt, K) - uvn) ** 2).sum() >= e0 - 1e-9


def test_fused_sgd_constructor_mirrors_torch():
    """singleshotpose_amd.optim.SGD keeps torch.optim.SGD's constructor contract (train.py:388) - host logic only."""
    import torch
    from singleshotpose_amd.optim import SGD
    p = torch.nn.Parameter(torch.zeros(4))
    opt = SGD([p], lr=0.001 / 8, momentum=0.9, dampening=0, weight_decay=0.0005 * 8)
    assert opt.param_groups[0]['lr'] == 0.001 / 8 and opt.param_groups[0]['weight_decay'] == 0.004
    for grp in opt.param_groups:      # train.py:44-45
        grp['lr'] = 0.5
    assert opt.state_dict()['param_groups'][0]['lr'] == 0.5
    opt.zero_grad()
    assert opt.step() is None          # no gradients: nothing to launch, no GPU needed
    for bad in (dict(lr=-1.0), dict(lr=0.1, momentum=-0.1), dict(lr=0.1, weight_decay=-1.0),
                dict(lr=0.1, nesterov=True), dict(lr=0.1, momentum=0.9, dampening=0.1, nesterov=True)):
        with pytest.raises(ValueError):
            SGD([p], **bad)


def test_dropin_modules_resolve_like_the_reference_scripts(tmp_path):
    """`PYTHONPATH=repo:repo/dropin` makes the reference's own import lines (train.py:18-23, valid.py:8-13,
    train_multi.py / valid_multi.py) resolve to this package: run them in a fresh interpreter."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import cv2                                   # train.py:15 / valid.py:8: the stand-in (never called on the hot path)
from darknet import Darknet                  # train.py:18
from cfg import parse_cfg                    # train.py:19
from region_loss import RegionLoss           # train.py:21
from utils import *                          # train.py:22
import utils
for n in """makedirs get_all_files calcAngularDistance get_camera_intrinsic compute_projection compute_transformation
calc_pts_diameter adi get_3D_corners pnp get_2d_bb compute_2d_bb compute_2d_bb_from_orig_pix corner_confidences
corner_confidence sigmoid softmax fix_corner_order convert2cpu convert2cpu_long get_region_boxes read_truths
read_truths_args read_pose load_class_names image2torch read_data_cfg scale_bboxes file_lines get_image_size
logging""".split():
    assert callable(getattr(utils, n)), n
import singleshotpose_amd
assert Darknet is singleshotpose_amd.darknet.Darknet and RegionLoss is singleshotpose_amd.region_loss.RegionLoss
m = Darknet(r"%s")
assert m.loss.__class__ is RegionLoss or m.loss.__class__.__name__ == "RegionLoss"
assert len(parse_cfg(r"%s")) == 33
import sys, os
sys.path.insert(0, os.path.join(r"%s", "dropin", "multi_obj_pose_estimation"))
from darknet_multi import Darknet as DM      # train_multi.py
from region_loss_multi import RegionLoss as RLM
import utils_multi
for n in "bbox_iou nms get_multi_region_boxes corner_confidences pnp get_3D_corners".split():
    assert callable(getattr(utils_multi, n)), n
print("ok")
''' % (os.path.join(root, 'cfg', 'yolo-pose.cfg'), os.path.join(root, 'cfg', 'yolo-pose.cfg'), root)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([root, os.path.join(root, 'dropin')]))
    out = subprocess.run([sys.executable, '-c', code], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith('ok'), out.stderr[-2000:]


def test_bbox_iou_and_nms_match_the_reference():
    """utils_multi.bbox_iou / nms (multi_obj_pose_estimation/utils_multi.py:125-156,223-241) against outputs of the
    reference's own functions (oracle/gen_golden.py --nms): kept boxes in order, and the in-place det_conf zeroing."""
    import numpy as np
    from helpers import gold
    from singleshotpose_amd.utils_multi import bbox_iou, nms
    g = gold('nms.npz')
    for case in range(3):
        b = g['in_%d' % case]
        iou = np.array([[bbox_iou(x, y, x1y1x2y2=False) for y in b] for x in b])
        np.testing.assert_allclose(iou, g['iou_%d' % case], rtol=1e-12, atol=1e-15)
        iou2 = np.array([[bbox_iou(np.sort(x[:4]), np.sort(y[:4]), x1y1x2y2=True) for y in b] for x in b])
        np.testing.assert_allclose(iou2, g['iou_xyxy_%d' % case], rtol=1e-12, atol=1e-15)
        boxes = [list(map(float, r)) for r in b]
        kept = nms(boxes, 0.4)
        assert np.array_equal(np.array(kept, dtype=np.float64).reshape(len(kept), -1), g['kept_%d' % case])
        assert np.array_equal(np.array(boxes, dtype=np.float64), g['after_%d' % case])
    assert nms([], 0.4) == []


def test_tune_cache_round_trip(tmp_path, monkeypatch):
    """SSP_TUNE_CACHE=<file>: the autotuner's per-shape plan choices survive the process (host logic, no GPU)."""
    from singleshotpose_amd import engine
    path = str(tmp_path / 'tune.json')
    monkeypatch.setenv('SSP_TUNE_CACHE', path)
    saved = dict(engine._TUNE_CACHE)
    old_file = engine._TUNE_CACHE_FILE[0]
    try:
        engine._TUNE_CACHE.clear()
        engine._TUNE_CACHE_FILE[0] = None
        engine._tune_cache_load()                      # no file yet: nothing loaded, path remembered
        key = ('fwd', 64, 13, 13, 512, 1024, 3, 512, 1024, True)
        engine._TUNE_CACHE[key] = 306413
        engine._TUNE_CACHE[('dgrad', 64, 26, 26, 512, 256, 3, 512, 256)] = 6413
        engine._tune_cache_save()
        assert os.path.isfile(path)
        engine._TUNE_CACHE.clear()
        engine._TUNE_CACHE_FILE[0] = None
        engine._tune_cache_load()
        assert engine._TUNE_CACHE[key] == 306413 and len(engine._TUNE_CACHE) == 2
        engine._TUNE_CACHE[key] = 12813                # an in-process choice is not overwritten by the file
        engine._TUNE_CACHE_FILE[0] = None
        engine._tune_cache_load()
        assert engine._TUNE_CACHE[key] == 12813
    finally:
        engine._TUNE_CACHE.clear()
        engine._TUNE_CACHE.update(saved)
        engine._TUNE_CACHE_FILE[0] = old_file


def _scipy_pnp(X, uv, K, R0, t0):
    """Independent minimiser of the pixel reprojection error (the objective cv2.solvePnP's ITERATIVE flag minimises):
    scipy's MINPACK Levenberg-Marquardt over (rotation vector, t), started from (R0, t0)."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation
    x0 = np.concatenate([Rotation.from_matrix(R0).as_rotvec(), np.asarray(t0).ravel()])

    def resid(p):
        R = Rotation.from_rotvec(p[:3]).as_matrix()
        cam = X.dot(R.T) + p[3:]
        proj = cam[:, :2] / cam[:, 2:3] * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]])
        return (proj - uv).ravel()
    sol = least_squares(resid, x0, method='lm', xtol=1e-15, ftol=1e-15, gtol=1e-15)
    return Rotation.from_rotvec(sol.x[:3]).as_matrix(), sol.x[3:], float((sol.fun ** 2).sum())


def test_pnp_oracle_matches_independent_least_squares_solver():
    """PnP parity is unpinned against OpenCV (not installable here).  What CAN be pinned: the ITERATIVE flag returns the
    Levenberg-Marquardt minimiser of the pixel reprojection error; an independent solver (scipy / MINPACK) must land on
    the same pose from noisy corners, whatever the starting point."""
    from oracle.pnp_ref import project, rodrigues, solve_pnp_ref
    K = np.array([[572.4114, 0, 325.2611], [0, 573.5704, 242.0489], [0, 0, 1.0]])
    X = np.concatenate([np.zeros((1, 3)), np.array([[sx * .038, sy * .039, sz * .046] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])], 0)
    rs = np.random.RandomState(5)
    for _ in range(10):
        axis = rs.standard_normal(3)
        R = rodrigues(axis / np.linalg.norm(axis) * rs.uniform(0, np.pi / 3))
        t = np.array([rs.uniform(-.1, .1), rs.uniform(-.1, .1), rs.uniform(.6, 1.2)])
        uvn = project(X, R, t, K) + rs.uniform(-1, 1, (9, 2))
        R1, t1 = solve_pnp_ref(X, uvn, K)
        R2, t2, cost2 = _scipy_pnp(X, uvn, K, R, t)              # started from the TRUE pose, not from ours
        cost1 = float(((project(X, R1, t1, K) - uvn) ** 2).sum())
        assert abs(cost1 - cost2) <= 1e-7 * max(cost2, 1.0)
        assert np.abs(project(X, R1, t1, K) - project(X, R2, t2, K)).max() < 1e-3
        assert np.abs(R1 - R2).max() < 1e-4 and np.abs(np.asarray(t1).ravel() - t2).max() < 1e-4


def test_flat_gradient_buffer_reuse_guard():
    """engine._storage_shared: the flat gradient buffer may be reused only while nothing but the buffer itself views its
    storage (a .grad view kept by a parameter or by the caller blocks the reuse)."""
    import torch
    from singleshotpose_amd import engine
    flat = torch.zeros(64)
    alone = engine._storage_refs(flat)
    p = torch.nn.Parameter(torch.zeros(4, 4))
    assert not engine._storage_shared(flat, alone, [p])
    view = flat[16:32].view(4, 4)
    assert engine._storage_shared(flat, alone, [p])            # a caller-held view
    del view
    assert not engine._storage_shared(flat, alone, [p])
    p.grad = torch.as_strided(flat, (4, 4), (4, 1), 0)
    assert engine._storage_shared(flat, alone, [p])            # a parameter's .grad
    assert engine._storage_shared(flat, None, [p])             # ... also seen without the storage counter
    p.grad = None
    assert not engine._storage_shared(flat, None, [p])


def test_reference_cpu_baseline_harness_runs_from_the_staged_archive():
    """bench.py's cpu_baseline leg, kind "reference": oracle/time_reference_cpu.py in its own process, reading the
    reference's modules from oracle/_ref/modules.zip as on the GPU box (staged by oracle/stage_reference.py)."""
    import json
    import subprocess
    import sys
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isfile(os.path.join(root, 'oracle', '_ref', 'modules.zip')):
        pytest.skip("oracle/_ref/modules.zip is not staged (oracle/stage_reference.py needs /root/reference)")
    prov = open(os.path.join(root, 'oracle', '_ref', 'PROVENANCE.txt')).read()
    assert 'darknet.py' in prov and 'sha1' in prov
    out = subprocess.run([sys.executable, os.path.join(root, 'oracle', 'time_reference_cpu.py'),
                          os.path.join(root, 'cfg', 'yolo-pose.cfg'), '1', '96', '4', '1'], capture_output=True, text=True,
                         timeout=600, env=dict(os.environ, SSP_REF_FROM_ZIP='1', CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES=''))
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec['params'] == 50547764 and rec['seconds_per_step']['4'] > 0 and 'modules.zip' in rec['modules']


def test_winograd_plan_queries_without_a_gpu():
    """The host-side sizing queries of the Winograd plans (pure functions of the launch shape: nothing touches the GPU):
    tile size of a plan code, workspace = (tile+2)^2 * tiles * (Cin + Cout), statistics in the counted format (groups of 16
    tiles, the groups' pixel counts behind the pairs), and the engine's own view of the codes."""
    from singleshotpose_amd import _lib
    from singleshotpose_amd.engine import WINO, WINO4, wino_tile
    q = _lib.query
    assert [q('ssp_conv_plan_wino_tile', c) for c in (0, 6413, 312813, 9006413, 9012818, 8006413, 8012814, 10000000)] == \
        [0, 0, 0, 2, 2, 4, 4, 0]
    assert [wino_tile(c) for c in (0, 6413, WINO + 6413, WINO4 + 12814)] == [0, 0, 2, 4]
    B, H, W, Cin, Cout = 64, 13, 13, 1024, 1024
    # tiles: plain B * ceil(H/n) * ceil(W/n), or the 2 x 2 image mosaic when it needs fewer (13 x 13 at n = 4: 49 per 4 images)
    assert q('ssp_conv_wino_tiles', 64, 13, 13, 4) == 16 * 49 and q('ssp_conv_wino_tiles', 64, 13, 13, 2) == 64 * 49
    assert q('ssp_conv_wino_tiles', 64, 26, 26, 4) == 64 * 49 and q('ssp_conv_wino_tiles', 64, 52, 52, 4) == 64 * 169
    assert q('ssp_conv_wino_tiles', 1, 21, 21, 4) == 36 and q('ssp_conv_wino_tiles', 4, 21, 21, 4) == 121
    assert q('ssp_conv_wino_tiles', 7, 13, 13, 4) == 2 * 49 and q('ssp_conv_wino_tiles', 5, 13, 13, 4) == 5 * 16
    assert q('ssp_conv_wino_tiles', 4, 13, 9, 4) == 7 * 5 and q('ssp_conv_wino_tiles', 8, 13, 13, 3) == 0
    for code, tile in ((9006413, 2), (8006413, 4)):
        T = q('ssp_conv_wino_tiles', B, H, W, tile)
        P = (tile + 2) ** 2
        assert q('ssp_conv_workspace_floats', B, H, W, Cin, Cout, 3, code) == P * T * (Cin + Cout)
        groups = (T + 15) // 16
        assert q('ssp_conv_stats_tile_m', B, H, W, Cin, Cout, 3, code) == 0
        assert q('ssp_conv_stats_tiles', B, H, W, Cin, Cout, 3, code) == groups
        assert q('ssp_conv_stats_floats', B, H, W, Cin, Cout, 3, code) == groups * Cout * 2 + groups
        assert q('ssp_conv_wgrad_wino_workspace_floats_t', B, H, W, Cin, Cout, tile) == P * (T * (Cin + Cout) + Cin * Cout)
    # (tile 2 is what the un-suffixed entry points mean)
    assert q('ssp_conv_wgrad_wino_workspace_floats', B, H, W, Cin, Cout) == q('ssp_conv_wgrad_wino_workspace_floats_t', B, H, W, Cin, Cout, 2)
    # a direct plan: rows per statistics tile, pairs only
    tm = q('ssp_conv_stats_tile_m', B, H, W, Cin, Cout, 3, 0)
    nt = q('ssp_conv_stats_tiles', B, H, W, Cin, Cout, 3, 0)
    assert tm in (64, 128) and nt == (B * H * W + tm - 1) // tm
    assert q('ssp_conv_stats_floats', B, H, W, Cin, Cout, 3, 0) == nt * Cout * 2


def test_tune_cache_keys_carry_the_candidate_set(monkeypatch):
    """A tune-cache entry written with a family of candidates switched off (or before it existed) must not be found by a
    process that has it on, and the other way round (engine._tune_tag)."""
    from singleshotpose_amd import engine
    monkeypatch.delenv('SSP_WINOGRAD', raising=False)
    monkeypatch.delenv('SSP_WINO_TILES', raising=False)
    monkeypatch.delenv('SSP_WINO_MIN_CHANNELS', raising=False)
    both = engine._tune_tag()
    monkeypatch.setenv('SSP_WINO_TILES', '2')
    only2 = engine._tune_tag()
    monkeypatch.setenv('SSP_WINOGRAD', '0')
    off = engine._tune_tag()
    assert len({both, only2, off}) == 3


def test_error_budget_choice_on_the_measured_table():
    """engine.choose_under_budget on the table the MI355X measured for yolo-pose.cfg at 416 x 416, batch 64
    (profiles/r05_bench.json head_budget): head deviation of F(4x4) / F(2x2) against the direct code, launch times in ms."""
    from singleshotpose_amd.engine import choose_under_budget
    F4 = {4: (0.6308, 3.05e-5), 6: (0.6308, 2.25e-5), 8: (0.4111, 1.96e-5), 10: (0.4111, 1.54e-5), 12: (0.328, 9.79e-6),
          14: (0.328, 6.25e-6), 16: (0.328, 4.98e-6), 18: (0.3027, 3.92e-6), 20: (0.3027, 2.82e-6), 22: (0.3027, 2.01e-6),
          23: (0.52, 1.80e-6), 24: (0.52, 1.68e-6), 29: (0.6339, 2.16e-6)}
    F2 = {8: (0.6752, 9.19e-6), 10: (0.6752, 6.43e-6), 12: (0.477, 4.60e-6), 14: (0.477, 2.84e-6), 16: (0.477, 2.69e-6),
          18: (0.4512, 1.80e-6), 20: (0.4512, 1.32e-6), 22: (0.4512, 1.04e-6), 23: (0.8425, 7.51e-7), 24: (0.8425, 7.06e-7),
          29: (1.0249, 5.56e-7)}
    D = {4: 0.7839, 6: 0.7839, 8: 0.7627, 10: 0.7627, 12: 0.725, 14: 0.725, 16: 0.725, 18: 0.7205, 20: 0.7205, 22: 0.7205,
         23: 1.392, 24: 1.392, 29: 1.7214}
    table = {}
    for i in F4:
        rows = [(4, 8000000 + i, F4[i][0], F4[i][1])]
        if i in F2:
            rows.append((2, 9000000 + i, F2[i][0], F2[i][1]))
        rows.append((0, 12813, D[i], 0.0))
        table[i] = rows
    dev = lambda ch: sum(table[i][ch[i]][3] ** 2 for i in table) ** 0.5
    ch, moved = choose_under_budget(table, 1e-3)                 # a budget nothing reaches: the fastest code everywhere
    assert moved == [] and all(k == 0 for k in ch.values()) and 4.5e-5 < dev(ch) < 5.0e-5
    ch, moved = choose_under_budget(table, 3.5e-5)               # the default: the two 104 x 104 layers go back to the direct code
    assert moved == [(4, 4, 0), (6, 4, 0)] and dev(ch) <= 3.5e-5
    ch, moved = choose_under_budget(table, 2.5e-5)               # ... then layer 8 to F(2x2): the cheapest error per millisecond
    assert moved == [(4, 4, 0), (6, 4, 0), (8, 4, 2)] and dev(ch) <= 2.5e-5
    ch, moved = choose_under_budget(table, 0.0)                  # nothing but the direct codes meets a zero budget
    assert all(table[i][ch[i]][0] == 0 for i in table) and dev(ch) == 0.0
    # a layer without a more accurate candidate cannot be moved: the loop ends instead of spinning
    ch, moved = choose_under_budget({3: [(4, 1, 0.1, 1e-5)]}, 1e-6)
    assert ch == {3: 0} and moved == []


def test_tune_cache_file_keeps_family_tables_and_budget_decisions(tmp_path, monkeypatch):
    from singleshotpose_amd import engine
    path = str(tmp_path / 'tune.json')
    monkeypatch.setenv('SSP_TUNE_CACHE', path)
    saved = (dict(engine._TUNE_CACHE), dict(engine._TUNE_FAMILY), dict(engine._HEAD_BUDGET_PINNED), engine._TUNE_CACHE_FILE[0])
    try:
        for d in (engine._TUNE_CACHE, engine._TUNE_FAMILY, engine._HEAD_BUDGET_PINNED):
            d.clear()
        engine._TUNE_CACHE_FILE[0] = None
        engine._tune_cache_load()
        key = ('fwd', 64, 104, 104, 64, 128, 3, 64, 128, True, engine._tune_tag())
        engine._TUNE_CACHE[key] = 8012813
        engine._TUNE_FAMILY[key] = {4: (8012813, 0.63), 0: (12813, 0.78)}
        engine._HEAD_BUDGET_PINNED[(64, 416, 416, engine._tune_tag(), 3.5e-5)] = {4: 12813, 8: 8006413}
        engine._tune_cache_save()
        for d in (engine._TUNE_CACHE, engine._TUNE_FAMILY, engine._HEAD_BUDGET_PINNED):
            d.clear()
        engine._TUNE_CACHE_FILE[0] = None
        engine._tune_cache_load()
        assert engine._TUNE_CACHE[key] == 8012813
        assert engine._TUNE_FAMILY[key] == {4: (8012813, 0.63), 0: (12813, 0.78)}
        assert engine._HEAD_BUDGET_PINNED[(64, 416, 416, engine._tune_tag(), 3.5e-5)] == {4: 12813, 8: 8006413}
    finally:
        for d, v in zip((engine._TUNE_CACHE, engine._TUNE_FAMILY, engine._HEAD_BUDGET_PINNED), saved[:3]):
            d.clear()
            d.update(v)
        engine._TUNE_CACHE_FILE[0] = saved[3]


def test_plan_code_families():
    """Plan-code helpers of the engine: the on-chip F(2x2) code is a Winograd plan of tile 2 (same filter transform, same
    error family in the head budget) and is told apart by wino_fused."""
    from singleshotpose_amd import engine
    assert engine.wino_tile(0) == 0 and engine.wino_tile(12813) == 0 and engine.wino_tile(306413) == 0
    assert engine.wino_tile(9006413) == 2 and engine.wino_tile(8012814) == 4
    assert engine.wino_tile(engine.WINOF) == 2 and engine.wino_fused(engine.WINOF)
    assert not engine.wino_fused(9006413) and not engine.wino_fused(8006413) and not engine.wino_fused(0)
    assert engine.WGRAD_FUSED == 12
