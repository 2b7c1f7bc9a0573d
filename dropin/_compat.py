"""torch-0.4 behaviour the reference's 2018 drivers rely on, restored for the processes that import the drop-in shims.

valid.py:124-138 / train.py:182-192 build Python lists of 0-dim CUDA tensors (`truths[k][j]` after `target.cuda()`) and
hand them to numpy (`np.reshape(box_gt[:18], [-1, 2])`).  In the torch 0.4.1 the reference was written for,
`Tensor.__array__` was `self.cpu().numpy()`; current torch raises for non-CPU tensors.  An environment shim like
dropin/torchvision and dropin/cv2.py: nothing on the device path depends on it, and it is installed only when one of the
module-name shims (utils / darknet) is imported, i.e. inside the reference's own driver processes.
"""
import torch

_orig_array = torch.Tensor.__array__


def _array_04(self, dtype=None, copy=None):
    """NumPy 2 passes `copy=`: a device tensor always comes back as a fresh host array, so only copy=False on a tensor
    that had to be moved cannot be honoured (numpy's own rule: raise)."""
    moved = self.is_cuda or self.requires_grad
    if moved and copy is False:
        raise ValueError("a %s tensor cannot be viewed as a numpy array without a copy" % self.device)
    t = self.detach().cpu() if moved else self
    a = _orig_array(t) if dtype is None else _orig_array(t, dtype)
    # copy=True (what NumPy 2 passes for np.array(t)) asks THIS method for a fresh array and does not copy on top of it: a
    # CPU tensor that was not moved would otherwise come back as torch's zero-copy view and np.array(t) would alias t
    if copy and not moved:
        a = a.copy()
    return a


# SSP_DROPIN_ARRAY_COMPAT=0 leaves torch.Tensor.__array__ alone (the reference's drivers then need torch 0.4 semantics from
# elsewhere); default on: the shims exist to run those drivers unchanged.
import os
if os.environ.get('SSP_DROPIN_ARRAY_COMPAT', '1') != '0' and \
        getattr(torch.Tensor.__array__, '__name__', '') != '_array_04':
    torch.Tensor.__array__ = _array_04
