"""GPU-side helpers for the -m gpu tests: thin wrappers that call the C ABI on torch device tensors."""
import numpy as np
import torch

from singleshotpose_amd import _lib


def dev():
    return torch.device('cuda', 0)


def stream():
    return torch.cuda.current_stream().cuda_stream


def to_nhwc(x_nchw, ld=None, off=0):
    """CPU NCHW tensor -> device [B*H*W][ld] buffer with the channels at [off, off+C); other columns NaN-poisoned."""
    B, C, H, W = x_nchw.shape
    ld = ld or C
    buf = torch.full((B * H * W, ld), float('nan'), dtype=torch.float32)
    buf[:, off:off + C] = x_nchw.permute(0, 2, 3, 1).reshape(B * H * W, C)
    return buf.to(dev())


def from_nhwc(buf, B, C, H, W, off=0):
    return buf.cpu()[:, off:off + C].reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()


def pack_fwd(w, cinp=None):
    cout, cin, k, _ = w.shape
    cinp = cinp or cin
    wd = w.contiguous().to(dev())
    out = torch.empty(cout * k * k * cinp, dtype=torch.float32, device=dev())
    _lib.call('ssp_repack_fwd', wd.data_ptr(), out.data_ptr(), cout, cin, cinp, k, stream())
    return out


def pack_dgrad(w, coutp=None):
    cout, cin, k, _ = w.shape
    coutp = coutp or cout
    wd = w.contiguous().to(dev())
    out = torch.empty(cin * k * k * coutp, dtype=torch.float32, device=dev())
    _lib.call('ssp_repack_dgrad', wd.data_ptr(), out.data_ptr(), cout, cin, coutp, k, stream())
    return out


def p(t, off=0):
    return t.data_ptr() + 4 * off
