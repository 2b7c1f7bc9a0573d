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
