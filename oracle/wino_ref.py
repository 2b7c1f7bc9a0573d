"""Winograd F(2x2, 3x3) in numpy - TEST INFRASTRUCTURE ONLY (never imported by the product path).

The product evaluates the deep 3x3 layers in the Winograd domain (singleshotpose_amd/csrc/conv_wino.hip).  The reference
has no such code - it calls nn.Conv2d (darknet.py:154-160) - so the parity target of those kernels is F.conv2d itself
(tests/test_gpu_wino.py, and every full-network check).  This file restates the transform arithmetic the kernels use, line
for line (same row / column combinations, same constants 1, -1, 1/2), so that the formulas are pinned on the CPU against
PyTorch's convolution and its autograd without a GPU (tests/test_oracle_wino.py):

  forward / data gradient   Y  = A^T [ sum_c (G g G^T) . (B^T d B) ] A      per 2x2 output tile
  filter gradient           dg = G^T [ sum_t (A dY A^T) . (B^T d B) ] G     summed over the tiles

Algorithm: Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks" (2016), F(2x2, 3x3), correlation form."""
import numpy as np


def input_transform(d):
    """B^T d B of 4x4 patches d[..., 4, 4] - wino_input_kernel's two passes."""
    r = np.stack([d[..., 0, :] - d[..., 2, :], d[..., 1, :] + d[..., 2, :], d[..., 2, :] - d[..., 1, :],
                  d[..., 1, :] - d[..., 3, :]], axis=-2)
    return np.stack([r[..., 0] - r[..., 2], r[..., 1] + r[..., 2], r[..., 2] - r[..., 1], r[..., 1] - r[..., 3]], axis=-1)


def filter_transform(g):
    """G g G^T of 3x3 filters g[..., 3, 3] - wino_filter_kernel."""
    h = np.stack([g[..., 0, :], (g[..., 0, :] + g[..., 1, :] + g[..., 2, :]) * 0.5,
                  (g[..., 0, :] - g[..., 1, :] + g[..., 2, :]) * 0.5, g[..., 2, :]], axis=-2)
    return np.stack([h[..., 0], (h[..., 0] + h[..., 1] + h[..., 2]) * 0.5, (h[..., 0] - h[..., 1] + h[..., 2]) * 0.5,
                     h[..., 2]], axis=-1)


def output_transform(m):
    """A^T m A of 4x4 tiles m[..., 4, 4] -> 2x2 - the gather of reduce_kernel<true> (rows {0,1,2} with +,+,+ for an even
    position, {1,2,3} with +,-,- for an odd one; the same over the columns)."""
    r = np.stack([m[..., 0, :] + m[..., 1, :] + m[..., 2, :], m[..., 1, :] - m[..., 2, :] - m[..., 3, :]], axis=-2)
    return np.stack([r[..., 0] + r[..., 1] + r[..., 2], r[..., 1] - r[..., 2] - r[..., 3]], axis=-1)


def outgrad_transform(dy):
    """A dY A^T of 2x2 output-gradient tiles dy[..., 2, 2] -> 4x4 - wino_outgrad_kernel."""
    r = np.stack([dy[..., 0, :], dy[..., 0, :] + dy[..., 1, :], dy[..., 0, :] - dy[..., 1, :], -dy[..., 1, :]], axis=-2)
    return np.stack([r[..., 0], r[..., 0] + r[..., 1], r[..., 0] - r[..., 1], -r[..., 1]], axis=-1)


def filtergrad_transform(du):
    """G^T dU G of 4x4 transform-domain filter gradients du[..., 4, 4] -> 3x3 - wino_wgrad_finish_kernel."""
    h = np.stack([du[..., 0, :] + (du[..., 1, :] + du[..., 2, :]) * 0.5, (du[..., 1, :] - du[..., 2, :]) * 0.5,
                  (du[..., 1, :] + du[..., 2, :]) * 0.5 + du[..., 3, :]], axis=-2)
    return np.stack([h[..., 0] + (h[..., 1] + h[..., 2]) * 0.5, (h[..., 1] - h[..., 2]) * 0.5,
                     (h[..., 1] + h[..., 2]) * 0.5 + h[..., 3]], axis=-1)


def _patches(x):
    """x (B, C, H, W) -> zero-padded 4x4 patches (B, th, tw, C, 4, 4) at origins (2ty-1, 2tx-1), th = ceil(H/2)."""
    B, C, H, W = x.shape
    th, tw = (H + 1) // 2, (W + 1) // 2
    xp = np.zeros((B, C, 2 * th + 2, 2 * tw + 2), dtype=x.dtype)
    xp[:, :, 1:H + 1, 1:W + 1] = x
    out = np.empty((B, th, tw, C, 4, 4), dtype=x.dtype)
    for ty in range(th):
        for tx in range(tw):
            out[:, ty, tx] = xp[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
    return out


def conv3x3(x, w):
    """'same' 3x3 cross-correlation of x (B, Cin, H, W) with w (Cout, Cin, 3, 3) through the Winograd domain."""
    B, _, H, W = x.shape
    V = input_transform(_patches(x))                            # (B, th, tw, Cin, 4, 4)
    U = filter_transform(w)                                     # (Cout, Cin, 4, 4)
    M = np.einsum('btscij,kcij->btskij', V, U)                  # 16 GEMMs over the channels
    Y = output_transform(M)                                     # (B, th, tw, Cout, 2, 2)
    th, tw = Y.shape[1], Y.shape[2]
    out = Y.transpose(0, 3, 1, 4, 2, 5).reshape(B, w.shape[0], 2 * th, 2 * tw)
    return out[:, :, :H, :W]


def conv3x3_wgrad(x, dy):
    """Filter gradient (Cout, Cin, 3, 3) of that convolution for the output gradient dy (B, Cout, H, W)."""
    B, Cout, H, W = dy.shape
    th, tw = (H + 1) // 2, (W + 1) // 2
    V = input_transform(_patches(x))
    dyp = np.zeros((B, Cout, 2 * th, 2 * tw), dtype=dy.dtype)
    dyp[:, :, :H, :W] = dy
    tiles = dyp.reshape(B, Cout, th, 2, tw, 2).transpose(0, 2, 4, 1, 3, 5)      # (B, th, tw, Cout, 2, 2)
    dM = outgrad_transform(tiles)
    dU = np.einsum('btskij,btscij->kcij', dM, V)                # 16 GEMMs over the tiles
    return filtergrad_transform(dU)
