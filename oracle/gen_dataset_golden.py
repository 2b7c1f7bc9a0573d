#!/usr/bin/env python
"""Golden epochs for the drop-in dataset (tests/test_dropin_cpu.py, tests/test_gpu_dropin.py): the reference's own
dataset.listDataset(train=True) + image.py (Pillow), iterated by a DataLoader as train.py:56-83 does, over the synthetic
LINEMOD-shaped fixture with three backgrounds of different sizes.  TEST INFRASTRUCTURE, build container only (needs
/root/reference):

    python oracle/gen_dataset_golden.py      -> tests/golden/dataset_epochs.json

Per configuration (seed, samples already seen -> which stage of the multi-scale schedule, epochs) and batch: the network
shape, the non-zero label rows exactly as float64 hex, and the SHA-1 of the batch's augmented pixels (B, H, W, 3 uint8).
The shapes and labels pin the host half of dropin/dataset.py (draw order, schedule, label arithmetic) on the CPU; the
pixel digests pin the GPU half.  Pillow computes the resampling in integer arithmetic, so the bytes do not depend on the
machine, only on the Pillow version (recorded).
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

# (name, seed, seen, epochs): `seen` 0 = the fixed 416 x 416 stage; 160 = 10 epochs of 16 samples -> widths 13..20 cells;
# 1200 = past epoch 70 -> widths 7..26 cells
CONFIGS = (('fixed_416', 0, 0, 1), ('multiscale_stage1', 1, 160, 2), ('multiscale_last_stage', 2, 1200, 2))


def digest_batches(npz):
    out = []
    i = 0
    while 'u8_%d' % i in npz:
        u8, lab = npz['u8_%d' % i], npz['lab_%d' % i]
        rows = lab.reshape(lab.shape[0], -1, 21)
        out.append(dict(shape=[int(u8.shape[2]), int(u8.shape[1])], batch=int(u8.shape[0]),
                        sha1=hashlib.sha1(np.ascontiguousarray(u8).tobytes()).hexdigest(),
                        labels=[[[float(v).hex() for v in r] for r in s if np.any(r != 0)] for s in rows]))
        i += 1
    return out


def main():
    import PIL
    import fixture_linemod as fx
    tmp = tempfile.mkdtemp(prefix='ssp_fixture_')
    gold = {}
    try:
        root = os.path.join(tmp, 'fixture')
        fx.make(root)
        fx.add_backgrounds(root)
        callers = os.path.join(tmp, 'callers')
        os.makedirs(callers)
        for n in ('dataset.py', 'image.py'):
            shutil.copy(os.path.join(REF, n), os.path.join(callers, n))
        env = dict(os.environ)
        env['PYTHONPATH'] = os.pathsep.join([callers, ROOT, os.path.join(ROOT, 'dropin')])
        for name, seed, seen, epochs in CONFIGS:
            out = os.path.join(tmp, name + '.npz')
            p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'dump_dataset_epoch.py'), root, out, '--seed',
                                str(seed), '--seen', str(seen), '--epochs', str(epochs), '--cpu'], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            assert p.returncode == 0, p.stdout[-3000:]
            z = np.load(out)
            assert str(z['module']).startswith(callers), z['module']
            gold[name] = dict(seed=seed, seen=seen, epochs=epochs, batches=digest_batches(z))
            print(name, [(b['shape'], b['sha1'][:10]) for b in gold[name]['batches']])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    gold['_meta'] = dict(generator='oracle/gen_dataset_golden.py', pillow=PIL.__version__,
                         source='/root/reference/dataset.py + image.py (unmodified) under tools/dump_dataset_epoch.py --cpu',
                         fixture='tests/fixture_linemod.py make() + add_backgrounds()')
    json.dump(gold, open(os.path.join(ROOT, 'tests', 'golden', 'dataset_epochs.json'), 'w'), indent=0, sort_keys=True)


if __name__ == '__main__':
    main()
