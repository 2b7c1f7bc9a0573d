#!/usr/bin/env python
"""Times the REFERENCE's own modules on the host CPU - bench.py's `cpu_baseline` leg, kind "reference".

BASELINE INFRASTRUCTURE (last row), always run as its OWN PROCESS: it redirects torch.cuda.* to
the CPU, which must never happen inside a process that drives the GPU.  The modules are the reference's unmodified
darknet.py / utils.py / cfg.py and region_loss.py with the three mechanical torch >= 0.5 patches of SURVEY.md section
8(c) applied in memory (oracle/gen_golden.load_patched), taken from /root/reference when it is mounted, else from
oracle/_ref/modules.zip (oracle/stage_reference.py; the GPU box).  One step = what train.py:83-103 does per batch:
Darknet.forward (train mode), RegionLoss.forward (epoch 20), loss.backward().

    python oracle/time_reference_cpu.py CFG BATCH SIZE THREADS[,THREADS...] [REPEATS]   -> one JSON line
"""
import contextlib
import io
import json
import os
import sys
import tempfile
import time
import types
import zipfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_dir():
    if os.path.isfile('/root/reference/darknet.py') and os.environ.get('SSP_REF_FROM_ZIP') != '1':
        return '/root/reference', 'in place'
    z = os.path.join(ROOT, 'oracle', '_ref', 'modules.zip')
    if not os.path.isfile(z):
        return None, None
    d = tempfile.mkdtemp(prefix='ssp_refmod_')
    zipfile.ZipFile(z).extractall(d)
    return d, 'oracle/_ref/modules.zip'


def main(argv):
    cfgfile, B, size = argv[0], int(argv[1]), int(argv[2])
    threads = [int(t) for t in argv[3].split(',')]
    repeats = int(argv[4]) if len(argv) > 4 else 1
    ref, how = reference_dir()
    if ref is None:
        print(json.dumps({"error": "no reference modules (neither /root/reference nor oracle/_ref/modules.zip)"}))
        return 2
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from oracle.gen_golden import load_patched
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))      # utils.py:10 imports cv2; never called on this path
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.LongTensor = torch.LongTensor
    sys.path.insert(0, ref)
    sys.modules['region_loss'] = load_patched(os.path.join(ref, 'region_loss.py'), 'region_loss')
    import darknet as ref_darknet
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref_darknet.Darknet(cfgfile)
    model.train()
    crit = model.loss
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, 3, size, size, generator=g)
    t = torch.zeros(B, 50, 21, dtype=torch.float64)
    t[:, 0, 1:19] = torch.rand(B, 18, generator=g, dtype=torch.float64) * 0.5 + 0.25
    t[:, 0, 19:21] = 0.2
    tgt = t.view(B, -1)

    def one_step():
        t0 = time.time()
        model.zero_grad()
        with contextlib.redirect_stdout(io.StringIO()):       # RegionLoss prints a status line per call
            loss = crit(model(x), tgt, 20)
        loss.backward()
        return time.time() - t0

    res = {}
    for nt in threads:
        torch.set_num_threads(nt)
        one_step()                                  # warm-up at this thread count (oneDNN primitive caches)
        res[nt] = float(np.median([one_step() for _ in range(repeats)]))
    print(json.dumps({"seconds_per_step": {str(k): v for k, v in res.items()}, "batch": B, "size": size,
                      "modules": "reference darknet.Darknet + region_loss.RegionLoss (%s)" % how,
                      "params": sum(p.numel() for p in model.parameters())}))
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
