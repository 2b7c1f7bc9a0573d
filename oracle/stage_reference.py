#!/usr/bin/env python
"""Stages reference files for the GPU box - TEST / BASELINE INFRASTRUCTURE, build container only (needs /root/reference).

The GPU box has no /root/reference.  Two things there need the reference's own files, byte for byte:
  * tests/test_gpu_dropin.py EXECUTES the reference's unmodified driver scripts against dropin/   -> callers.zip
  * bench.py's `cpu_baseline` leg (kind "reference") times the reference's own Darknet + RegionLoss
    modules on the host CPU (oracle/time_reference_cpu.py, SURVEY.md section 8(d) last row)        -> modules.zip
Both archives go to oracle/_ref/, which is listed in .gitignore (reference source never enters this repo's history) and
not in .gpurunignore (it travels with the snapshot, like a compiled oracle/_ref artefact would).  Nothing on the product
path reads them; the consumers skip / fall back ("kind": "port") with a clear message when an archive is absent.
PROVENANCE.txt records where the files came from and under which licence (the reference's LICENSE.txt: MIT).

    python oracle/stage_reference.py          (also run by __graft_entry__.build() when /root/reference is mounted)
"""
import hashlib
import os
import zipfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
DST = os.path.join(ROOT, 'oracle', '_ref')

# driver scripts + the helper modules they import that are OUT of this repo's scope (PIL data pipeline, mesh reader);
# darknet.py / region_loss.py / utils.py / cfg.py are deliberately NOT in this archive: for the drivers those names
# resolve to dropin/
CALLERS = ('valid.py', 'train.py', 'dataset.py', 'image.py', 'MeshPly.py',
           # BASELINE config 5: the multi-object drivers and THEIR data pipeline (darknet_multi / region_loss_multi / utils_multi
           # resolve to dropin/multi_obj_pose_estimation/)
           'multi_obj_pose_estimation/train_multi.py', 'multi_obj_pose_estimation/valid_multi.py',
           'multi_obj_pose_estimation/dataset_multi.py', 'multi_obj_pose_estimation/image_multi.py')
# the hot path's own modules, for the CPU baseline only (never on sys.path of a test or of the product)
MODULES = ('darknet.py', 'region_loss.py', 'utils.py', 'cfg.py')


def _zip(name, files):
    path = os.path.join(DST, name)
    with zipfile.ZipFile(path, 'w', zipfile.ZIP_DEFLATED) as z:
        for f in files:
            z.write(os.path.join(REF, f), f)
    return path


def stage():
    if not os.path.isdir(REF):
        return None
    os.makedirs(DST, exist_ok=True)
    out = [_zip('callers.zip', CALLERS), _zip('modules.zip', MODULES)]
    lic = [f for f in ('LICENSE.txt', 'LICENSE', 'LICENSE.md') if os.path.isfile(os.path.join(REF, f))]
    with open(os.path.join(DST, 'PROVENANCE.txt'), 'w') as fp:
        fp.write("Unmodified files of microsoft/singleshotpose, copied from %s by oracle/stage_reference.py.\n" % REF)
        fp.write("Licence: see the reference's %s (MIT).  Git-ignored: test / baseline staging only.\n" % (lic[0] if lic else 'LICENSE'))
        for f in CALLERS + MODULES:
            fp.write("%s  sha1 %s\n" % (f, hashlib.sha1(open(os.path.join(REF, f), 'rb').read()).hexdigest()))
    return out


if __name__ == '__main__':
    print(stage())
