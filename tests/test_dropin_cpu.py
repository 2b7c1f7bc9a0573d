"""CPU-side checks of the drop-in plumbing: the torchvision shim, the fixture generator, the output parsers."""
import json
import os
import subprocess
import sys

import numpy as np
import torch

from helpers import GOLD, ROOT
import fixture_linemod as fx


def test_torchvision_shim_matches_totensor_semantics():
    sys.path.insert(0, os.path.join(ROOT, 'dropin'))
    try:
        sys.modules.pop('torchvision', None)
        from torchvision import datasets, transforms  # noqa: F401
        from PIL import Image, ImageMath
        rs = np.random.RandomState(0)
        arr = rs.randint(0, 256, (5, 7, 3)).astype(np.uint8)
        t = transforms.Compose([transforms.ToTensor(), ])(Image.fromarray(arr))
        want = torch.from_numpy(arr).permute(2, 0, 1).float().div(255)
        assert t.dtype == torch.float32 and torch.equal(t, want)
        assert hasattr(ImageMath, 'eval')          # image.py:116 (Pillow 12 dropped the name)
    finally:
        sys.path.remove(os.path.join(ROOT, 'dropin'))
        for k in [k for k in sys.modules if k == 'torchvision' or k.startswith('torchvision.')]:
            del sys.modules[k]


def test_fixture_is_deterministic_and_linemod_shaped(tmp_path):
    from singleshotpose_amd.utils import read_data_cfg, read_truths_args
    a, b = str(tmp_path / 'a'), str(tmp_path / 'b')
    fx.make(a, n_train=2, n_test=1)
    fx.make(b, n_train=2, n_test=1)
    for rel in ('LINEMOD/ape/JPEGImages/000001.png', 'LINEMOD/ape/mask/0001.png', 'LINEMOD/ape/labels/000002.txt',
                'LINEMOD/ape/ape.ply', 'init.weights', 'cfg/ape.data'):
        assert open(os.path.join(a, rel), 'rb').read() == open(os.path.join(b, rel), 'rb').read(), rel
    opt = read_data_cfg(os.path.join(a, 'cfg', 'ape.data'))
    assert opt['mesh'] == 'LINEMOD/ape/ape.ply' and float(opt['fx']) == fx.FX
    lab = read_truths_args(os.path.join(a, 'LINEMOD/ape/labels/000000.txt'))
    assert lab.shape == (19,) and lab[0] == 0 and 0 < lab[1] < 1
    assert os.path.getsize(os.path.join(a, 'init.weights')) == 16 + 4 * (50547764 + 2 * 0 + sum(
        2 * c for c in (32, 64, 128, 64, 128, 256, 128, 256, 512, 256, 512, 256, 512, 1024, 512, 1024, 512, 1024, 1024, 1024, 64, 1024)))


def test_parsers_on_the_golden_records():
    v = json.load(open(os.path.join(GOLD, 'dropin_valid.json')))
    t = json.load(open(os.path.join(GOLD, 'dropin_train.json')))
    text = ("x Number of test samples: 8\nx    Acc using 5 px 2D Projection = %.2f%%\n"
            "x    Acc using 10%% threshold - %s vx 3D Transformation = %.2f%%\nx    Acc using 5 cm 5 degree metric = %.2f%%\n"
            "x    Mean 2D pixel error is %f, Mean vertex error is %f, mean corner error is %f\n"
            "x    Translation error: %f m, angle error: %f degree, pixel error: % f pix\n" % (
                v['acc_2d_5px'], v['adi_threshold'], v['acc_3d_10pct'], v['acc_5cm5deg'], v['mean_pixel_err'],
                v['mean_vertex_err'], v['mean_corner_err'], v['trans_err'], v['angle_err'], v['pixel_err']))
    got = fx.parse_valid_output(text)
    for k in got:
        assert abs(got[k] - v[k]) <= 1e-6 * max(1.0, abs(v[k])), k
    s = t['steps'][0]
    line = '%d: nGT %d, recall %d, proposals %d, loss: x %f, y %f, conf %f, total %f' % (
        s['seen'], s['nGT'], s['recall'], s['proposals'], s['loss_x'], s['loss_y'], s['loss_conf'], s['total'])
    assert fx.parse_train_output('epoch 0, processed 0 samples, lr 0.000100\n' + line + '\n')['steps'][0] == s


def test_run_pinned_pins_the_wall_clock_seed(tmp_path):
    script = tmp_path / 'probe.py'
    script.write_text("import time, random, torch\nseed = int(time.time())\ntorch.manual_seed(seed)\n"
                      "print('PROBE', random.random(), float(torch.rand(1)))\n")
    outs = [subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'run_pinned.py'), str(script)], cwd=str(tmp_path),
                           stdout=subprocess.PIPE, text=True, check=True).stdout for _ in range(2)]
    assert outs[0] == outs[1] and 'PROBE' in outs[0]


def test_array_shim_honours_numpy2_copy_semantics():
    """dropin/_compat.py: np.array(cpu_tensor) must not alias the tensor (NumPy 2 passes copy=True to __array__ and does
    not copy on top of what it gets back); np.asarray may; copy=False on a tensor that has to be moved raises."""
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "import _compat\n"
        "t = torch.arange(4, dtype=torch.float32)\n"
        "a = np.array(t)\n"
        "a[0] = 99\n"
        "assert t[0].item() == 0.0, 'np.array(t) aliases the tensor'\n"
        "assert not np.shares_memory(np.array(t), t.numpy())\n"
        "b = np.asarray(t)\n"
        "assert np.shares_memory(b, t.numpy())\n"
        "g = torch.ones(3, requires_grad=True)\n"
        "assert np.array(g).tolist() == [1.0, 1.0, 1.0]\n"
        "try:\n"
        "    g.__array__(copy=False)\n"
        "    raise SystemExit('copy=False on a grad tensor did not raise')\n"
        "except ValueError:\n"
        "    pass\n"
        "assert np.array(t, dtype=np.float64).dtype == np.float64\n"
        "print('ok')\n") % os.path.join(ROOT, 'dropin')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stdout + r.stderr


# ---- dropin/dataset.py: the host half (draw order, multi-scale schedule, labels, collate), no GPU --------------------------
def _dropin_dataset():
    """(module, cleanup): dropin's `dataset` (and the `utils` / torchvision shims it resolves) imported from dropin/."""
    d = os.path.join(ROOT, 'dropin')
    sys.path.insert(0, d)
    for k in ('dataset', 'utils', 'image'):
        sys.modules.pop(k, None)
    import dataset

    def cleanup():
        sys.path.remove(d)
        for k in [k for k in sys.modules if k in ('dataset', 'utils', 'image', '_compat', 'torchvision') or k.startswith('torchvision.')]:
            del sys.modules[k]
    return dataset, cleanup


def _epoch_batches(dataset, root, seed, seen, epochs, workers=0, batch=8):
    import random
    from torchvision import transforms
    cwd = os.getcwd()
    os.chdir(root)
    try:
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        bgdir = os.path.join('VOCdevkit', 'VOC2012', 'JPEGImages')
        bgs = [os.path.join(bgdir, f) for f in sorted(os.listdir(bgdir))]
        out = []
        for _ in range(epochs):
            loader = torch.utils.data.DataLoader(
                dataset.listDataset(os.path.join('LINEMOD', 'ape', 'train.txt'), shape=(416, 416), shuffle=True,
                                    transform=transforms.Compose([transforms.ToTensor(), ]), train=True, seen=seen,
                                    batch_size=batch, num_workers=workers, bg_file_names=bgs),
                batch_size=batch, shuffle=False, num_workers=workers)
            for data, target in loader:
                out.append((data, target))
                seen += len(data)
        return out
    finally:
        os.chdir(cwd)


def test_dropin_dataset_draws_shapes_and_labels_of_the_reference(tmp_path):
    """The reference's own dataset.listDataset over the same seeded epochs (oracle/gen_dataset_golden.py): the network shape
    of every batch (multi-scale schedule, dataset.py:66-90) and every label value, bit for bit - they depend on the shuffle,
    the shape / background draws and the four jitter draws of every sample, i.e. on the whole draw order."""
    gold = json.load(open(os.path.join(GOLD, 'dataset_epochs.json')))
    root = str(tmp_path / 'fixture')
    fx.make(root)
    fx.add_backgrounds(root)
    dataset, cleanup = _dropin_dataset()
    try:
        for name in ('fixed_416', 'multiscale_stage1', 'multiscale_last_stage'):
            g = gold[name]
            got = _epoch_batches(dataset, root, g['seed'], g['seen'], g['epochs'])
            assert len(got) == len(g['batches'])
            for (data, target), gb in zip(got, g['batches']):
                assert type(data).__name__ == 'RawBatch' and list(data.shape) == gb['shape'] and len(data) == gb['batch']
                assert data.size() == (gb['batch'], gb['shape'][1], gb['shape'][0], 3) and data.size(0) == gb['batch']
                assert target.dtype == torch.float64 and tuple(target.shape) == (gb['batch'], 50 * 21)
                rows = target.numpy().reshape(gb['batch'], 50, 21)
                for s in range(gb['batch']):
                    want = np.array([[float.fromhex(v) for v in r] for r in gb['labels'][s]]).reshape(-1, 21)
                    assert np.array_equal(rows[s, :len(want)], want), (name, s)
                    assert not rows[s, len(want):].any()
                # what the GPU pass will be handed: decoded bytes and one draw set per sample
                smp = data.samples[0]
                assert smp.img.dtype == torch.uint8 and tuple(smp.img.shape) == (480, 640, 3) == tuple(smp.mask.shape)
                assert set(smp.draws) == {'pleft', 'pright', 'ptop', 'pbot', 'flip', 'dhue', 'dsat', 'dexp'}
                try:
                    data.mean()
                    assert False, "RawBatch must not behave like a tensor before .cuda()"
                except AttributeError as e:
                    assert 'cuda()' in str(e)
    finally:
        cleanup()


def test_dropin_dataset_through_worker_processes_and_eval_branch(tmp_path):
    """num_workers > 0: default_collate runs in the worker (the RawSample registration is inherited) and the RawBatch
    crosses back with its byte tensors; workers are seeded by the DataLoader, so two runs agree.  train=False: the
    reference's host branch (dataset.py:108-131) - resized PIL image through the caller's transform, float32 labels."""
    root = str(tmp_path / 'fixture')
    fx.make(root, n_train=8, n_test=2, batch=4)
    fx.add_backgrounds(root)
    dataset, cleanup = _dropin_dataset()
    try:
        a = _epoch_batches(dataset, root, 3, 80, 1, workers=2, batch=4)
        b = _epoch_batches(dataset, root, 3, 80, 1, workers=2, batch=4)
        assert len(a) == len(b) == 2
        for (da, ta), (db, tb) in zip(a, b):
            assert type(da).__name__ == 'RawBatch' and da.shape == db.shape and torch.equal(ta, tb)
            assert [s.draws for s in da.samples] == [s.draws for s in db.samples]
            assert all(torch.equal(x.img, y.img) and torch.equal(x.bg, y.bg) for x, y in zip(da.samples, db.samples))
            # ONE tensor crosses the worker boundary per batch (one shared-memory segment / file descriptor, one pin):
            # image, mask and background of every sample are views of it
            assert da.blob.dtype == torch.uint8 and da.blob.dim() == 1 and len(da.layout) == 4
            base = da.blob.untyped_storage().data_ptr()
            assert all(t.untyped_storage().data_ptr() == base for x in da.samples for t in (x.img, x.mask, x.bg))
        assert dataset.multiscale_width(0, 2, 8) == 13 and dataset.multiscale_width(159, 2, 8) == 13

        class Fixed(object):
            def __init__(self):
                self.calls = []

            def randint(self, lo, hi):
                self.calls.append((lo, hi))
                return hi
        for stage, (hi, base) in enumerate([(7, 13), (9, 12), (11, 11), (13, 10), (15, 9), (17, 8), (19, 7), (19, 7), (19, 7)], 1):
            r = Fixed()
            assert dataset.multiscale_width(160 * stage, 2, 8, r) == hi + base and r.calls == [(0, hi)]
        assert dataset.multiscale_width(5, 0, 8, Fixed()) == 26          # fewer samples than one batch: the last branch

        from torchvision import transforms
        cwd = os.getcwd()
        os.chdir(root)
        try:
            ds = dataset.listDataset('LINEMOD/ape/test.txt', shape=(96, 64), shuffle=False,
                                     transform=transforms.Compose([transforms.ToTensor(), ]), train=False)
            img, label = ds[1]
            from PIL import Image
            want = transforms.ToTensor()(Image.open('LINEMOD/ape/JPEGImages/000009.png').convert('RGB').resize((96, 64)))
            assert torch.equal(img, want) and tuple(img.shape) == (3, 64, 96)
            assert label.dtype == torch.float32 and label.numel() == 50 * 21 and label[1] > 0 and not label[21:].any()
            try:
                dataset.listDataset('LINEMOD/ape/train.txt', train=True, transform=lambda x: x, bg_file_names=['x'])
                assert False
            except TypeError as e:
                assert 'ToTensor' in str(e)
        finally:
            os.chdir(cwd)
    finally:
        cleanup()


def test_occlusion_fixture_is_what_the_multi_drivers_open(tmp_path):
    """tests/fixture_occlusion.py: the files train_multi.py / valid_multi.py open, where they look for them (cwd = <root>/multi,
    '../LINEMOD/...', '../VOCdevkit/...'), the labels_occlusion rows from the golden record, deterministic; and the parsers on
    the golden's own numbers."""
    import hashlib
    import fixture_occlusion as fo
    gold = json.load(open(os.path.join(GOLD, 'dropin_multi.json')))
    info = fo.make(str(tmp_path / 'a'))
    root, cwd = info['root'], info['cwd']
    assert info['test_images'] == sorted(gold['labels_occlusion']['ape'])
    for rel in ('cfg/occlusion.data', 'cfg/train_occlusion.txt', 'cfg/yolo-pose-multi.cfg', 'init.weights'):
        assert os.path.isfile(os.path.join(cwd, rel)), rel
    for o in fo.VALID:
        assert os.path.isfile(os.path.join(cwd, 'cfg', '%s_occlusion.data' % o))
        lines = open(os.path.join(root, 'LINEMOD', o, 'test_occlusion.txt')).read().split()
        assert len(lines) == 4 and all(os.path.isfile(os.path.join(cwd, l)) for l in lines)
        for nm in info['test_images']:       # dataset_multi.py:76: benchvise image path -> this object's labels_occlusion row
            row = np.loadtxt(os.path.join(root, 'LINEMOD', o, 'labels_occlusion', nm + '.txt'))
            assert row.shape == (21,) and int(row[0]) == fo.CLASS_ID[o]
            assert np.allclose(row, gold['labels_occlusion'][o][nm], atol=1e-8)
    for o in fo.PASTED:                     # image_multi.py:320-326: '../LINEMOD/<obj>/train.txt' lines, prefixed with '../'
        for l in open(os.path.join(root, 'LINEMOD', o, 'train.txt')).read().split():
            img = os.path.join(cwd, '..', l)
            assert os.path.isfile(img)
            assert os.path.isfile(img.replace('JPEGImages', 'mask').replace('/00', '/'))
            assert os.path.isfile(img.replace('JPEGImages', 'labels').replace('.png', '.txt'))
    train = open(os.path.join(cwd, 'cfg', 'train_occlusion.txt')).read().split()
    assert len(train) == 8 and all(os.path.isfile(os.path.join(cwd, l)) for l in train)
    # deterministic: a second build gives the same scene bytes and the same weights
    b = fo.make(str(tmp_path / 'b'))
    for rel in ('LINEMOD/benchvise/JPEGImages/000009.png', 'LINEMOD/cat/mask/0002.png', 'multi/init.weights'):
        ha, hb = (hashlib.sha1(open(os.path.join(r, rel), 'rb').read()).hexdigest() for r in (root, b['root']))
        assert ha == hb, rel
    # the golden record itself: two training batches of four with 8 labels per image, sixty accuracy numbers with a spread
    assert [s['nGT'] for s in gold['train']['steps']] == [32, 32] and len(gold['train']['steps_one_thread']) == 2
    assert list(gold['valid']) == list(fo.VALID) and all(len(v) == 10 for v in gold['valid'].values())
    assert {a for v in gold['valid'].values() for a in v} >= {0.0, 25.0, 50.0, 75.0, 100.0}
    text = '4: nGT 32, recall 0, proposals 3309, loss: x 168.752808, y 210.650909, conf 349.375793, cls 89.837532, total 469.241241\n' \
           '2026-09-25 08:05:47 epoch 0, processed 0 samples, lr 0.000100\n2026 Testing ape...\n   Acc using 5 px 2D Projection = 25.00%\n'
    assert fo.parse_train_output(text)['steps'][0]['loss_cls'] == 89.837532
    assert fo.parse_valid_output(text) == {'ape': [25.0]}
