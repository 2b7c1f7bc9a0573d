"""oracle/wino_ref.py (numpy restatement of the transforms the Winograd kernels use) against PyTorch's convolution and
its autograd (what the reference runs, darknet.py:154-160): forward, data gradient (the same routine on flipped /
transposed filters, as the kernels do it) and filter gradient, on odd, even and single-row maps, in float64 (formula
check) and float32 (the size of the rounding the algorithm adds)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import wino_ref as W


@pytest.mark.parametrize("B,C,K,H,Wd", [(2, 3, 4, 5, 7), (1, 5, 2, 6, 4), (3, 2, 3, 1, 5), (1, 4, 4, 13, 13)])
def test_winograd_forward_dgrad_wgrad_formulas(B, C, K, H, Wd):
    rs = np.random.RandomState(B * 100 + H)
    x = rs.standard_normal((B, C, H, Wd))
    g = rs.standard_normal((K, C, 3, 3))
    dy = rs.standard_normal((B, K, H, Wd))
    xt = torch.from_numpy(x).requires_grad_(True)
    gt = torch.from_numpy(g).requires_grad_(True)
    y = F.conv2d(xt, gt, padding=1)
    y.backward(torch.from_numpy(dy))
    assert np.abs(W.conv3x3(x, g) - y.detach().numpy()).max() < 1e-12
    # data gradient = the same correlation of dy with the flipped, in/out-transposed filters (ssp_repack_dgrad's layout)
    gflip = g[:, :, ::-1, ::-1].transpose(1, 0, 2, 3)
    assert np.abs(W.conv3x3(dy, gflip) - xt.grad.numpy()).max() < 1e-12
    assert np.abs(W.conv3x3_wgrad(x, dy) - gt.grad.numpy()).max() < 1e-11


def test_winograd_float32_rounding_is_far_inside_the_parity_bar():
    """In float32 the transforms (constants 1, -1, 1/2 only) add ~1e-6 of the output's range - two orders of magnitude
    inside the 1e-4 bar, and what tests/test_gpu_wino.py measures for the kernels."""
    rs = np.random.RandomState(0)
    x = rs.standard_normal((2, 64, 13, 13)).astype(np.float32)
    g = (rs.standard_normal((32, 64, 3, 3)) / 24.0).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(g).double(), padding=1).numpy()
    got = W.conv3x3(x, g)
    assert got.dtype == np.float32
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-5
