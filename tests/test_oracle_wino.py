"""oracle/wino_ref.py (numpy restatement of the transforms the Winograd kernels use, F(2x2,3x3) and F(4x4,3x3)) against
PyTorch's convolution and its autograd (what the reference runs, darknet.py:154-160): forward, data gradient (the same
routine on flipped / transposed filters, as the kernels do it) and filter gradient, on odd, even and single-row maps, in
float64 (formula check) and float32 (the size of the rounding the algorithm adds); and the kernels' own coefficient
tables, read out of csrc/conv_wino.hip, against the matrices the oracle derives from the interpolation points."""
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import wino_ref as W


@pytest.mark.parametrize("tile", [2, 4])
@pytest.mark.parametrize("B,C,K,H,Wd", [(2, 3, 4, 5, 7), (1, 5, 2, 6, 4), (3, 2, 3, 1, 5), (1, 4, 4, 13, 13), (2, 2, 2, 8, 12)])
def test_winograd_forward_dgrad_wgrad_formulas(B, C, K, H, Wd, tile):
    rs = np.random.RandomState(B * 100 + H)
    x = rs.standard_normal((B, C, H, Wd))
    g = rs.standard_normal((K, C, 3, 3))
    dy = rs.standard_normal((B, K, H, Wd))
    xt = torch.from_numpy(x).requires_grad_(True)
    gt = torch.from_numpy(g).requires_grad_(True)
    y = F.conv2d(xt, gt, padding=1)
    y.backward(torch.from_numpy(dy))
    assert np.abs(W.conv3x3(x, g, tile) - y.detach().numpy()).max() < 1e-11
    # data gradient = the same correlation of dy with the flipped, in/out-transposed filters (ssp_repack_dgrad's layout)
    gflip = g[:, :, ::-1, ::-1].transpose(1, 0, 2, 3)
    assert np.abs(W.conv3x3(dy, gflip, tile) - xt.grad.numpy()).max() < 1e-11
    assert np.abs(W.conv3x3_wgrad(x, dy, tile) - gt.grad.numpy()).max() < 1e-10


@pytest.mark.parametrize("tile,bar", [(2, 1e-5), (4, 2e-5)])
def test_winograd_float32_rounding_is_far_inside_the_parity_bar(tile, bar):
    """In float32 the transforms add ~1e-6 (tile 2: constants 1, -1, 1/2 only) / a few 1e-6 (tile 4) of the output's range -
    more than an order of magnitude inside the 1e-4 bar, and what tests/test_gpu_wino.py measures for the kernels."""
    rs = np.random.RandomState(0)
    x = rs.standard_normal((2, 64, 13, 13)).astype(np.float32)
    g = (rs.standard_normal((32, 64, 3, 3)) / 24.0).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(g).double(), padding=1).numpy()
    got = W.conv3x3(x, g, tile)
    assert got.dtype == np.float32
    assert np.abs(got - ref).max() / np.abs(ref).max() < bar


def _kernel_tables(tile):
    """The BT / G / AT initialisers of `template <> struct WinoMat<tile>` in conv_wino.hip, evaluated."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'singleshotpose_amd', 'csrc', 'conv_wino.hip')
    src = open(path).read()
    body = src[src.index('template <> struct WinoMat<%d>' % tile):]
    end = body.index('\n};')
    body = body[:end]
    out = {}
    for name in ('BT', 'G', 'AT'):
        m = re.search(r'static constexpr float %s\[(\d+)\]\[(\d+)\] = (\{.*?\});' % name, body, re.S)
        rows, cols = int(m.group(1)), int(m.group(2))
        text = re.sub(r'(\d+\.?\d*)f', r'\1', m.group(3)).replace('{', '[').replace('}', ']')
        arr = np.array(eval(text), dtype=np.float64)      # literals and fractions of literals only
        assert arr.shape == (rows, cols)
        out[name] = arr
    return out['BT'], out['G'], out['AT']


@pytest.mark.parametrize("tile", [2, 4])
def test_kernel_coefficient_tables_are_the_oracle_matrices(tile):
    BT, G, AT = _kernel_tables(tile)
    oBT, oG, oAT = W.matrices(tile)
    np.testing.assert_allclose(BT, oBT, rtol=0, atol=0)
    np.testing.assert_allclose(AT, oAT, rtol=0, atol=0)
    np.testing.assert_allclose(G, oG, rtol=1e-15, atol=0)      # thirds / fifteenths: the same quotient of the same literals
