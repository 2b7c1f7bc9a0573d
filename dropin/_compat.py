"""torch-0.4 behaviour the reference's 2018 drivers rely on, restored for the processes that import the drop-in shims.

valid.py:124-138 / train.py:182-192 build Python lists of 0-dim CUDA tensors (`truths[k][j]` after `target.cuda()`) and
hand them to numpy (`np.reshape(box_gt[:18], [-1, 2])`).  In the torch 0.4.1 the reference was written for,
`Tensor.__array__` was `self.cpu().numpy()`; current torch raises for non-CPU tensors.  An environment shim like
dropin/torchvision and dropin/cv2.py: nothing on the device path depends on it, and it is installed only when one of the
module-name shims (utils / darknet) is imported, i.e. inside the reference's own driver processes.
"""
import torch

_orig_array = torch.Tensor.__array__


def _array_04(self, dtype=None):
    t = self.detach().cpu() if (self.is_cuda or self.requires_grad) else self
    return _orig_array(t) if dtype is None else _orig_array(t, dtype)


if getattr(torch.Tensor.__array__, '__name__', '') != '_array_04':
    torch.Tensor.__array__ = _array_04
