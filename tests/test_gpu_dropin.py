"""The drop-in, executed: the reference's UNMODIFIED valid.py and train.py run against this package on the MI355X.

north_star: "keep the reference's plugin surface ... so train.py / valid.py drop in unchanged".  The driver scripts are
the reference's own files, byte for byte (staged by __graft_entry__.build() into the git-ignored oracle/_ref/callers.zip
because the GPU box has no /root/reference; or taken from /root/reference when it is there).  They run as subprocesses
with PYTHONPATH=<repo>:<repo>/dropin - the module-name shims (cfg, darknet, region_loss, utils, cv2 stand-in,
torchvision's Compose / ToTensor) - over the synthetic LINEMOD-shaped fixture of tests/fixture_linemod.py, and what they
print is compared with tests/golden/dropin_*.json: the same scripts run on the CPU reference in the build container
(oracle/gen_dropin_golden.py; darknet.py / region_loss.py / utils.py = the reference's, PnP = oracle/pnp_ref.py).

valid.py (valid.py:15-233): Darknet(cfg).load_weights, eval forward at 672 x 672, get_region_boxes, pnp x 2 per image,
the evaluation maths, calc_pts_diameter.   train.py (train.py:330-410, 48-127): load_weights_until_last, the
DataLoader over dataset.listDataset(train=True) with the PIL augmentations, adjust_learning_rate, zero_grad / forward /
RegionLoss / backward / torch.optim.SGD.step for two epochs of two batches.
"""
import json
import os
import subprocess
import sys
import zipfile

import pytest

from helpers import GOLD, ROOT
import fixture_linemod as fx

pytestmark = pytest.mark.gpu
CALLERS = ('valid.py', 'train.py', 'dataset.py', 'image.py', 'MeshPly.py')


def _callers_dir(tmp_path, names=CALLERS, sub='reference_callers'):
    """A directory holding ONLY the driver scripts (and dataset.py / image.py / MeshPly.py): `python <dir>/valid.py`
    puts <dir> first on sys.path, and darknet / region_loss / utils / cfg must resolve to dropin/, not to the reference.
    `names` without dataset.py / image.py: those two resolve to dropin/ as well (the GPU augmentation)."""
    import shutil
    dst = str(tmp_path / sub)
    os.makedirs(dst, exist_ok=True)
    ref = '/root/reference'
    if all(os.path.isfile(os.path.join(ref, n)) for n in names):
        for n in names:
            shutil.copy(os.path.join(ref, n), os.path.join(dst, n))
        return dst
    z = os.path.join(ROOT, 'oracle', '_ref', 'callers.zip')
    if not os.path.isfile(z):
        pytest.skip("the reference's driver scripts are not staged (oracle/_ref/callers.zip: run __graft_entry__.build() "
                    "in the build container, where /root/reference exists)")
    with zipfile.ZipFile(z) as f:
        for n in names:
            f.extract(n, dst)
    return dst


def _run(cmd, cwd, timeout=900, first=()):
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join(list(first) + [ROOT, os.path.join(ROOT, 'dropin')])
    env['PYTHONUNBUFFERED'] = '1'
    p = subprocess.run(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-4000:]
    return p.stdout


def _close(a, b, rel, abs_=0.0):
    return abs(a - b) <= abs_ + rel * abs(b)


def test_unmodified_valid_py_runs_and_matches_the_cpu_reference(tmp_path):
    gold = json.load(open(os.path.join(GOLD, 'dropin_valid.json')))
    ref = _callers_dir(tmp_path)
    root = str(tmp_path / 'fixture')
    fx.make(root, max_epochs=2)
    out = _run([sys.executable, os.path.join(ref, 'valid.py'), '--datacfg', 'cfg/ape.data', '--modelcfg',
                'cfg/yolo-pose.cfg', '--weightfile', 'init.weights'], root)
    got = fx.parse_valid_output(out)
    print({k: (got[k], gold[k]) for k in got})
    assert got['nsamples'] == gold['nsamples'] == 8
    # network + decode only (no PnP): mean 2D corner error of the predicted box, pixels of the 640 x 480 image
    assert _close(got['mean_corner_err'], gold['mean_corner_err'], 1e-4, 1e-3)
    # ADD threshold = 0.1 * calc_pts_diameter(mesh): exact arithmetic on both sides
    assert got['adi_threshold'] == gold['adi_threshold']
    # through PnP (HIP kernel here, oracle/pnp_ref.py in the golden): the printed means have 6 decimals
    assert _close(got['mean_pixel_err'], gold['mean_pixel_err'], 1e-4, 1e-2)
    assert _close(got['pixel_err'], gold['pixel_err'], 1e-4, 1e-2)
    assert _close(got['mean_vertex_err'], gold['mean_vertex_err'], 1e-4, 1e-5)
    assert _close(got['trans_err'], gold['trans_err'], 1e-4, 1e-5)
    assert _close(got['angle_err'], gold['angle_err'], 1e-4, 1e-3)
    for k in ('acc_2d_5px', 'acc_3d_10pct', 'acc_5cm5deg'):
        assert got[k] == gold[k], k


@pytest.mark.parametrize("data_pipeline", ["reference_pillow", "dropin_gpu"])
def test_unmodified_train_py_runs_and_matches_the_cpu_reference(tmp_path, data_pipeline):
    """reference_pillow: `import dataset` in train.py finds the reference's dataset.py / image.py (host Pillow pipeline).
    dropin_gpu: those two files are NOT beside train.py, so `dataset` is dropin/dataset.py - every batch is augmented by
    DeviceAugmenter on the GPU inside train.py's own `data.cuda()`; same draws, same bytes, hence the same golden numbers."""
    gold = json.load(open(os.path.join(GOLD, 'dropin_train.json')))
    ref = _callers_dir(tmp_path) if data_pipeline == 'reference_pillow' else \
        _callers_dir(tmp_path, ('valid.py', 'train.py', 'MeshPly.py'), 'callers_without_dataset')
    assert os.path.isfile(os.path.join(ref, 'dataset.py')) == (data_pipeline == 'reference_pillow')
    root = str(tmp_path / 'fixture')
    fx.make(root, max_epochs=2)
    out = _run([sys.executable, os.path.join(ROOT, 'tools', 'run_pinned.py'), os.path.join(ref, 'train.py'), '--datacfg',
                'cfg/ape.data', '--modelcfg', 'cfg/yolo-pose.cfg', '--initweightfile', 'init.weights'], root)
    got = fx.parse_train_output(out)
    print(json.dumps(got['steps']))
    assert got['epochs'] == gold['epochs']                       # lr schedule (train.py:34-46) and sample counters
    assert len(got['steps']) == len(gold['steps']) == 4
    alts = gold['steps_other_thread_counts']
    for i, (a, b) in enumerate(zip(got['steps'], gold['steps'])):
        assert (a['seen'], a['nGT']) == (b['seen'], b['nGT'])
        for k in ('loss_x', 'loss_y', 'loss_conf', 'total'):
            if i == 0:
                # identical weights: the fp32 parity bar (north_star: conv / loss within 1e-4 relative)
                assert _close(a[k], b[k], 1e-4, 1e-5), (i, k, a[k], b[k])
            else:
                # After an optimizer step the reference does not reproduce itself across summation orders (the golden
                # holds its runs with 1 and 3 OpenMP threads next to the default): the first layer's filter gradient is
                # a heavily cancelling sum (tools/train_parity.py) and training amplifies its
                # rounding (measured here: 1.2e-3 / 2.7 % / 6.7 % on loss_x at batches 2 / 3 / 4 against the reference's own
                # 5e-4 / 0.4 % / 4.3 %).  Allowed: 3x the reference's own spread at this batch + 5 %.  What this cannot pin
                # - the optimizer step itself - is pinned where the arithmetic is well conditioned:
                # tests/test_gpu_optim.py and test_gpu_darknet.py::test_training_trajectory_tracks_oracle (1e-5 over 4 steps).
                spread = max(abs(alt[i][k] - b[k]) for alt in alts)
                assert abs(a[k] - b[k]) <= 3.0 * spread + 0.05 * abs(b[k]), (i, k, a[k], b[k], spread)
        if i == 0:
            assert (a['proposals'], a['recall']) == (b['proposals'], b['recall'])


def _digest(z):
    import hashlib
    import numpy as np
    out, i = [], 0
    while 'u8_%d' % i in z:
        u8 = np.ascontiguousarray(z['u8_%d' % i])
        out.append(([int(u8.shape[2]), int(u8.shape[1])], hashlib.sha1(u8.tobytes()).hexdigest()))
        i += 1
    return out


@pytest.mark.parametrize("name", ["fixed_416", "multiscale_stage1", "multiscale_last_stage"])
def test_dropin_dataset_epoch_is_the_reference_epoch_byte_for_byte(tmp_path, name):
    """SURVEY.md section 8(f) row 3 as a drop-in: dropin/dataset.py's listDataset inside a DataLoader, `data.cuda()` as in
    train.py:82-83, over seeded epochs of the fixture (three background sizes; fixed 416 x 416, the first and the last
    stage of the multi-scale schedule: 256 x 256 ... 736 x 736) - the pixels of every batch are the bytes the reference's
    dataset.py + image.py produce (SHA-1 per batch from oracle/gen_dataset_golden.py, the reference run in the build
    container), the labels are equal as float64.  Where the reference's two files are staged the reference pipeline is
    also run HERE, on this machine's Pillow, and compared array against array."""
    import numpy as np
    gold = json.load(open(os.path.join(GOLD, 'dataset_epochs.json')))[name]
    root = str(tmp_path / 'fixture')
    fx.make(root)
    fx.add_backgrounds(root)
    tool = os.path.join(ROOT, 'tools', 'dump_dataset_epoch.py')
    args = ['--seed', str(gold['seed']), '--seen', str(gold['seen']), '--epochs', str(gold['epochs'])]
    out = str(tmp_path / 'dropin.npz')
    _run([sys.executable, tool, root, out] + args, str(tmp_path))
    got = np.load(out)
    assert str(got['module']) == os.path.join(ROOT, 'dropin', 'dataset.py')
    dig = _digest(got)
    assert [d[0] for d in dig] == [b['shape'] for b in gold['batches']]
    assert [d[1] for d in dig] == [b['sha1'] for b in gold['batches']]
    for i, b in enumerate(gold['batches']):
        rows = got['lab_%d' % i].reshape(b['batch'], 50, 21)
        for s in range(b['batch']):
            want = np.array([[float.fromhex(v) for v in r] for r in b['labels'][s]]).reshape(-1, 21)
            assert np.array_equal(rows[s, :len(want)], want) and not rows[s, len(want):].any()
    z = os.path.join(ROOT, 'oracle', '_ref', 'callers.zip')
    if os.path.isfile(z) or os.path.isdir('/root/reference'):
        ref = _callers_dir(tmp_path, ('dataset.py', 'image.py'), 'reference_dataset')
        out_ref = str(tmp_path / 'reference.npz')
        _run([sys.executable, tool, root, out_ref, '--cpu'] + args, str(tmp_path), first=[ref])
        want = np.load(out_ref)
        assert str(want['module']).startswith(ref)
        for i in range(len(dig)):
            a, b = got['u8_%d' % i], want['u8_%d' % i]
            assert a.shape == b.shape and a.dtype == b.dtype
            assert np.array_equal(a, b), "batch %d: %d of %d bytes differ" % (i, int((a != b).sum()), a.size)
            assert np.array_equal(got['lab_%d' % i], want['lab_%d' % i])


_IMAGE_PROBE = r'''
import hashlib, random, sys
import numpy as np
from torchvision import transforms      # (dropin's shim also restores ImageMath.eval, which Pillow 12 dropped: image.py:116)
import image
random.seed(11)
img, label = image.load_data_detection('LINEMOD/ape/JPEGImages/000003.png', (352, 288), 0.2, 0.1, 1.5, 1.5,
                                       'VOCdevkit/VOC2012/JPEGImages/bg2.png', 9, 50)
d = image.distort_image(img, 0.07, 1.3, 0.8)
print('PROBE', image.__file__, img.size, hashlib.sha1(np.asarray(img.convert('RGB')).tobytes()).hexdigest(),
      hashlib.sha1(np.asarray(label, dtype=np.float64).tobytes()).hexdigest(),
      hashlib.sha1(np.asarray(d.convert('RGB')).tobytes()).hexdigest())
'''


def test_dropin_image_module_per_sample_functions_equal_the_reference(tmp_path):
    """dropin/image.py keeps image.py's per-sample names (PIL in / out) on the GPU kernels: load_data_detection and
    distort_image give the reference's bytes and labels for the same seed (non-square network shape, a background of
    another aspect ratio)."""
    z = os.path.join(ROOT, 'oracle', '_ref', 'callers.zip')
    if not (os.path.isfile(z) or os.path.isdir('/root/reference')):
        pytest.skip("the reference's image.py is not staged (oracle/_ref/callers.zip)")
    root = str(tmp_path / 'fixture')
    fx.make(root, n_train=4, n_test=1)
    fx.add_backgrounds(root)
    ref = _callers_dir(tmp_path, ('image.py',), 'reference_image')
    a = [l for l in _run([sys.executable, '-c', _IMAGE_PROBE], root).splitlines() if l.startswith('PROBE')][0].split()
    b = [l for l in _run([sys.executable, '-c', _IMAGE_PROBE], root, first=[ref]).splitlines() if l.startswith('PROBE')][0].split()
    assert a[1] == os.path.join(ROOT, 'dropin', 'image.py') and b[1].startswith(ref)
    assert a[2:] == b[2:], (a, b)


_BATCH_PROBE = r'''
import os, random, sys
import numpy as np, torch
from torchvision import transforms
import dataset, image
random.seed(5); torch.manual_seed(5)
bgdir = 'VOCdevkit/VOC2012/JPEGImages'
ds = dataset.listDataset('LINEMOD/ape/train.txt', shape=(416, 416), shuffle=False, transform=transforms.Compose([transforms.ToTensor()]),
                         train=True, seen=0, batch_size=4, num_workers=0, bg_file_names=[os.path.join(bgdir, f) for f in sorted(os.listdir(bgdir))])
data, target = next(iter(torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, num_workers=0, pin_memory=True)))
a = data.cuda()
b = data.to('cuda')
c = data.to(device=torch.device('cuda', 0))
assert a.dtype == torch.uint8 and tuple(a.shape) == (4, 416, 416, 3) == data.size() and torch.equal(a, b) and torch.equal(a, c)
os.environ['SSP_DATASET_FLOAT'] = '1'
f = data.cuda()
assert f.dtype == torch.float32 and tuple(f.shape) == (4, 3, 416, 416)
assert torch.equal(f, a.permute(0, 3, 1, 2).float().div(255))          # ToTensor's layout and arithmetic
try:
    data.to('cpu'); raise SystemExit('RawBatch.to(cpu) must refuse')
except RuntimeError as e:
    assert 'no CPU fallback' in str(e)
# image.random_distort_image: the three draws of image.py:39-44, then the GPU distort of the same values
from PIL import Image
im = Image.fromarray(a[0].cpu().numpy())
random.seed(9); got = image.random_distort_image(im, 0.1, 1.5, 1.5)
random.seed(9); dh = random.uniform(-0.1, 0.1); ds_ = image.rand_scale(1.5); de = image.rand_scale(1.5)
want = image.distort_image(im, dh, ds_, de)
assert np.array_equal(np.asarray(got), np.asarray(want))
from singleshotpose_amd.darknet import Darknet
print('PROBE_OK', float(a.float().mean()))
'''


def test_rawbatch_device_entry_points_and_float_mode(tmp_path):
    """RawBatch.cuda() / .to('cuda') / .to(device=...) give the same uint8 batch; SSP_DATASET_FLOAT=1 returns ToTensor's own
    (B, 3, H, W) float batch of the same bytes; .to('cpu') refuses; dropin image.random_distort_image draws in image.py's
    order."""
    root = str(tmp_path / 'fixture')
    fx.make(root, n_train=4, n_test=1, batch=4)
    fx.add_backgrounds(root)
    out = _run([sys.executable, '-c', _BATCH_PROBE], root)
    assert 'PROBE_OK' in out, out[-2000:]


# ---- BASELINE config 5: the multi-object drivers ("train_multi.py, valid_multi.py drop in unchanged") ----
MULTI_CALLERS = ('train_multi.py', 'valid_multi.py', 'dataset_multi.py', 'image_multi.py')


def _multi_callers_dir(tmp_path):
    """multi_obj_pose_estimation/{train_multi,valid_multi,dataset_multi,image_multi}.py + MeshPly.py, the reference's own files,
    in a directory of their own: `python <dir>/train_multi.py` puts <dir> first on sys.path, so dataset_multi / image_multi
    (the PIL pipeline, out of this repo's scope) are the reference's, while darknet_multi / region_loss_multi / utils_multi /
    cfg resolve to dropin/."""
    import shutil
    dst = str(tmp_path / 'multi_callers')
    os.makedirs(dst, exist_ok=True)
    ref = '/root/reference'
    if all(os.path.isfile(os.path.join(ref, 'multi_obj_pose_estimation', n)) for n in MULTI_CALLERS):
        for n in MULTI_CALLERS:
            shutil.copy(os.path.join(ref, 'multi_obj_pose_estimation', n), os.path.join(dst, n))
        shutil.copy(os.path.join(ref, 'MeshPly.py'), os.path.join(dst, 'MeshPly.py'))
        return dst
    z = os.path.join(ROOT, 'oracle', '_ref', 'callers.zip')
    if not os.path.isfile(z) or 'multi_obj_pose_estimation/train_multi.py' not in zipfile.ZipFile(z).namelist():
        pytest.skip("the reference's multi-object driver scripts are not staged (oracle/_ref/callers.zip: run "
                    "__graft_entry__.build() in the build container, where /root/reference exists)")
    with zipfile.ZipFile(z) as f:
        for n in MULTI_CALLERS:
            with open(os.path.join(dst, n), 'wb') as o:
                o.write(f.read('multi_obj_pose_estimation/' + n))
        f.extract('MeshPly.py', dst)
    return dst


MULTI_SHIMS = os.path.join(ROOT, 'dropin', 'multi_obj_pose_estimation')


def test_unmodified_train_multi_py_runs_and_matches_the_cpu_reference(tmp_path):
    """train_multi.py:76-100,330-410 against the drop-in: Darknet(cfg) from darknet_multi, load_weights_until_last,
    `torch.nn.DataParallel(model, device_ids=[0]).cuda()` around the product module (train_multi.py:387; `model.module.seen`,
    `.width`, `.save_weights`), the reference's own dataset_multi / image_multi pipeline (8 labels per image: the benchvise
    scene + 7 pasted objects), RegionLoss from region_loss_multi with the cfg's anchors, torch.optim.SGD - one epoch of two
    batches of four over the OCCLUSION-shaped fixture.  The first batch (identical weights) must give the reference's
    loss terms to the fp32 bar; the second, after one optimizer step, within 3x the reference's own run-to-run spread."""
    import fixture_occlusion as fo
    gold = json.load(open(os.path.join(GOLD, 'dropin_multi.json')))['train']
    ref = _multi_callers_dir(tmp_path)
    info = fo.make(str(tmp_path / 'fixture'))
    out = _run([sys.executable, os.path.join(ROOT, 'tools', 'run_pinned.py'), os.path.join(ref, 'train_multi.py'), '--datacfg',
                'cfg/occlusion.data', '--modelcfg', 'cfg/yolo-pose-multi.cfg', '--initweightfile', 'init.weights'], info['cwd'],
               first=[MULTI_SHIMS])
    got = fo.parse_train_output(out)
    print(json.dumps(got['steps']))
    assert got['epochs'] == gold['epochs']
    assert len(got['steps']) == len(gold['steps']) == 2
    for i, (a, b, b1) in enumerate(zip(got['steps'], gold['steps'], gold['steps_one_thread'])):
        assert (a['seen'], a['nGT']) == (b['seen'], b['nGT']) == (4 * (i + 1), 32)
        for k in ('loss_x', 'loss_y', 'loss_conf', 'loss_cls', 'total'):
            if i == 0:
                assert _close(a[k], b[k], 1e-4, 1e-5), (i, k, a[k], b[k])
            else:
                # The second batch sits behind one optimizer step AND the loss's hard decisions (which anchor / cell owns a
                # label, which predictions pass the confidence threshold: `proposals` moves by 1-2): it is a chaotic function of
                # the first step's rounding.  Measured on one box over four plan sets of the SAME kernels (round 6: library
                # heuristics / tuned direct plans only / + Winograd / + on-chip Winograd): loss_x 141.933 / 142.235 / 142.196 /
                # 142.297 against the reference's 141.932 (all-threads) and 141.939 (one thread) - fp32 summation order alone
                # (the tuned DIRECT plans) moves it by 0.21 %.  Bar: 3x the reference's own spread + 0.5 %.
                spread = abs(b1[k] - b[k])
                assert abs(a[k] - b[k]) <= 3.0 * spread + 5e-3 * abs(b[k]), (i, k, a[k], b[k], spread)
        if i == 0:
            assert a['recall'] == b['recall'] and abs(a['proposals'] - b['proposals']) <= 2, (a, b)


def test_unmodified_valid_multi_py_runs_and_matches_the_cpu_reference(tmp_path):
    """valid_multi.py:18-174 against the drop-in, run as `python valid_multi.py` (its main block also tests __package__):
    Darknet(cfg).load_weights, eval forward, get_multi_region_boxes(output, conf_thresh, ..., only_objectness=0) on the real
    head (~740 boxes pass the threshold per image), the highest-confidence box of the truth's class, pnp x 2, the mesh
    reprojection - for six objects x four scenes.  The fixture's labels put the reference's own pixel error of every sample
    at a known 3 ... 57 px, each at least 1.2 px from a threshold (tests/fixture_occlusion.py), so its ten accuracy lines per
    object read 25 / 50 / 75 / 100 %; the drop-in must print the same sixty numbers."""
    import fixture_occlusion as fo
    gold = json.load(open(os.path.join(GOLD, 'dropin_multi.json')))['valid']
    ref = _multi_callers_dir(tmp_path)
    info = fo.make(str(tmp_path / 'fixture'))
    out = _run([sys.executable, os.path.join(ref, 'valid_multi.py'), '--modelcfg', 'cfg/yolo-pose-multi.cfg', '--initweightfile',
                'init.weights'], info['cwd'], first=[MULTI_SHIMS])
    got = fo.parse_valid_output(out)
    print(json.dumps(got))
    assert list(got) == list(fo.VALID)
    assert all(len(v) == 10 for v in got.values())
    assert any(0.0 < a < 100.0 for v in gold.values() for a in v)
    assert got == gold
