#!/usr/bin/env python
"""Runs one of the reference's own driver scripts (valid.py / train.py) ON THE CPU REFERENCE - golden generation only.

TEST INFRASTRUCTURE, build container only (needs /root/reference):
    python oracle/run_reference_cpu.py [--seed N] /root/reference/valid.py --datacfg ... (cwd = the fixture directory)
    python oracle/run_reference_cpu.py /root/reference/multi_obj_pose_estimation/train_multi.py ...   (cwd = fixture/multi)

The script and every module it imports (darknet.py, utils.py, dataset.py, image.py, MeshPly.py, cfg.py) are the
reference's own files, imported from where they lie.  What this harness supplies around them, because the reference
targets CUDA + torch 0.4 + OpenCV and this container has none of the three:
  * torch.cuda.* / .cuda() -> CPU (as oracle/gen_golden.py);
  * region_loss.py with the three mechanical torch >= 0.5 patches of SURVEY.md section 8(c), loaded in memory;
  * a `cv2` module whose solvePnP / Rodrigues are oracle/pnp_ref.py (OpenCV's ITERATIVE algorithm restated: parity
    unpinned against real OpenCV);
  * torchvision's Compose / ToTensor from dropin/torchvision (pure PIL / numpy, also restores PIL.ImageMath.eval);
  * the same randomness pinning as tools/run_pinned.py.
"""
import os
import runpy
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'


def install(multi=False):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from oracle.gen_golden import load_patched
    from oracle.pnp_ref import solve_pnp_ref
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.LongTensor = torch.LongTensor
    cv2 = types.ModuleType('cv2')

    class _Pose(object):      # what solvePnP hands to Rodrigues
        def __init__(self, R):
            self.R = R

    def solvePnP(points_3D, points_2D, cameraMatrix, distCoeffs, *a, **k):
        R, t = solve_pnp_ref(np.asarray(points_3D, dtype=np.float64), np.asarray(points_2D, dtype=np.float64).reshape(-1, 2),
                             np.asarray(cameraMatrix, dtype=np.float64))
        return True, _Pose(R), t

    cv2.solvePnP = solvePnP
    cv2.Rodrigues = lambda pose: (pose.R, None)
    sys.modules['cv2'] = cv2
    # module search order: the reference first (darknet, utils, cfg, dataset, image, MeshPly), then dropin/ for the one
    # name the reference does not have (torchvision)
    sys.path.insert(0, os.path.join(ROOT, 'dropin'))
    sys.path.insert(0, REF)
    sys.modules['region_loss'] = load_patched(os.path.join(REF, 'region_loss.py'), 'region_loss')
    if multi:
        # multi_obj_pose_estimation/: darknet_multi, utils_multi, dataset_multi, image_multi are the reference's own files;
        # region_loss_multi.py gets the same three mechanical patches as region_loss.py
        mdir = os.path.join(REF, 'multi_obj_pose_estimation')
        sys.path.insert(0, mdir)
        sys.modules['region_loss_multi'] = load_patched(os.path.join(mdir, 'region_loss_multi.py'), 'region_loss_multi')


def main(argv):
    seed = 0
    if argv and argv[0] == '--seed':
        seed, argv = int(argv[1]), argv[2:]
    install(multi='multi' in os.path.basename(argv[0]))
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from run_pinned import pin
    pin(seed)
    sys.argv = argv
    # executed as `python script.py` would: __name__ == '__main__' and __package__ None (valid_multi.py:161 tests both;
    # runpy.run_path would set __package__ to '')
    path = os.path.abspath(argv[0])
    glb = {'__name__': '__main__', '__package__': None, '__file__': path, '__builtins__': __builtins__}
    exec(compile(open(path).read(), path, 'exec'), glb)


if __name__ == '__main__':
    main(sys.argv[1:])
