"""Environment shim: the two names the reference's drivers take from torchvision.

`from torchvision import datasets, transforms` (train.py:15, valid.py:8) is only ever used as
`transforms.Compose([transforms.ToTensor(),])` (train.py:59, valid.py:100); this image has no torchvision wheel and
no network.  First-party equivalents of exactly those two transforms live in .transforms; `datasets` is an empty
namespace (imported, never touched).  If a real torchvision is installed, do not put this directory on PYTHONPATH.

Also restored here, because this is the first module the drivers import that needs Pillow: `PIL.ImageMath.eval`, which
the reference's image.py:116 calls and Pillow 12 removed (renamed `unsafe_eval` in 10.3; same evaluator, same
semantics).  An environment shim, not a CUDA shim - nothing on the device path depends on either.
"""
from . import datasets, transforms  # noqa: F401

try:
    from PIL import ImageMath as _ImageMath
    if not hasattr(_ImageMath, 'eval') and hasattr(_ImageMath, 'unsafe_eval'):
        _ImageMath.eval = _ImageMath.unsafe_eval
except ImportError:      # no Pillow: dataset.py will say so itself
    pass

__version__ = '0+singleshotpose_amd.shim'
