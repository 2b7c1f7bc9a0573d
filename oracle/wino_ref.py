"""Winograd F(n x n, 3x3), n = 2 and 4, in numpy - TEST INFRASTRUCTURE ONLY (never imported by the product path).

The product evaluates the deep 3x3 layers in the Winograd domain (singleshotpose_amd/csrc/conv_wino.hip).  The reference
has no such code - it calls nn.Conv2d (darknet.py:154-160) - so the parity target of those kernels is F.conv2d itself
(tests/test_gpu_wino.py, and every full-network check).  This file restates the transform arithmetic the kernels use -
the same matrices B^T, G, A^T, applied rows first, then columns - so that the formulas are pinned on the CPU against
PyTorch's convolution and its autograd without a GPU (tests/test_oracle_wino.py); the same test reads the kernels' own
coefficient tables out of conv_wino.hip and compares them with the matrices derived here.

  forward / data gradient   Y  = A^T [ sum_c (G g G^T) . (B^T d B) ] A      per n x n output tile
  filter gradient           dg = G^T [ sum_t (A dY A^T) . (B^T d B) ] G     summed over the tiles

Algorithm: Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks" (2016), correlation form; the matrices follow
from the Cook-Toom construction on the interpolation points below (Vandermonde A^T, Lagrange-denominator-scaled G, B^T
from the products of the linear factors).  n = 2: points (0, 1, -1), rows signed / scaled as Lavin & Gray print them.
n = 4: points (0, 1, -1, 1/2, -2) rather than the textbook (0, +-1, +-2): smaller constants in B^T and A^T, and on
this network's operand statistics about half the rounding error (conv_wino.hip's header has the numbers)."""
from fractions import Fraction as Fr

import numpy as np

POINTS = {2: (0, 1, -1), 4: (0, 1, -1, Fr(1, 2), -2)}


def _polymul(p, q):
    res = [Fr(0)] * (len(p) + len(q) - 1)
    for i, x in enumerate(p):
        for j, y in enumerate(q):
            res[i + j] += x * y
    return res


def cook_toom(points, n, r=3):
    """Exact (Fraction) matrices A^T (n x a), G (a x r), B^T (a x a), a = n + r - 1, for the finite points + infinity."""
    a = n + r - 1
    pts = [Fr(p) for p in points]
    assert len(pts) == a - 1
    AT = [[pts[j] ** i for j in range(a - 1)] + [Fr(1 if i == n - 1 else 0)] for i in range(n)]
    G = []
    for j in range(a - 1):
        den = Fr(1)
        for m in range(a - 1):
            if m != j:
                den *= pts[j] - pts[m]
        G.append([pts[j] ** k / den for k in range(r)])
    G.append([Fr(0)] * (r - 1) + [Fr(1)])
    BT = []
    for j in range(a - 1):
        poly = [Fr(1)]
        for m in range(a - 1):
            if m != j:
                poly = _polymul(poly, [-pts[m], Fr(1)])
        BT.append(poly + [Fr(0)] * (a - len(poly)))
    poly = [Fr(1)]
    for m in range(a - 1):
        poly = _polymul(poly, [-pts[m], Fr(1)])
    BT.append(poly)
    return AT, G, BT


def matrices(tile, dtype=np.float64):
    """(B^T, G, A^T) as the kernels hold them.  tile 2: Lavin & Gray's sign / scale convention (row 0 of G and B^T negated,
    row 3 of A^T's column negated, relative to the raw construction - an equivalent factorisation)."""
    AT, G, BT = cook_toom(POINTS[tile], tile)
    f = lambda m: np.array([[float(x) for x in row] for row in m], dtype=np.float64)
    AT, G, BT = f(AT), f(G), f(BT)
    if tile == 2:
        # raw: G row 0 = (-1, 0, 0), B^T row 0 = (-1, 0, 1, 0), B^T row 3 = (0, -1, 0, 1), A^T column 3 = (0, 1)
        G[0] *= -1.0
        BT[0] *= -1.0
        BT[3] *= -1.0
        AT[:, 3] *= -1.0
    return BT.astype(dtype), G.astype(dtype), AT.astype(dtype)


def _apply(mat, x, axis):
    """mat applied along `axis` of x (one pass of a separable transform), in x's dtype."""
    return np.moveaxis(np.tensordot(mat.astype(x.dtype), np.moveaxis(x, axis, 0), axes=(1, 0)), 0, axis)


def input_transform(d, tile=2):
    """B^T d B of (tile+2)^2 patches d[..., a, a] - wino_input_kernel's two passes (rows, then columns)."""
    BT, _, _ = matrices(tile)
    return _apply(BT, _apply(BT, d, -2), -1)


def filter_transform(g, tile=2):
    """G g G^T of 3x3 filters g[..., 3, 3] - wino_filter_kernel."""
    _, G, _ = matrices(tile)
    return _apply(G, _apply(G, g, -2), -1)


def output_transform(m, tile=2):
    """A^T m A of a x a tiles m[..., a, a] -> tile x tile - wino_output_kernel."""
    _, _, AT = matrices(tile)
    return _apply(AT, _apply(AT, m, -1), -2)


def outgrad_transform(dy, tile=2):
    """A dY A^T of tile x tile output-gradient tiles -> a x a - wino_outgrad_kernel."""
    _, _, AT = matrices(tile)
    return _apply(AT.T, _apply(AT.T, dy, -2), -1)


def filtergrad_transform(du, tile=2):
    """G^T dU G of a x a transform-domain filter gradients -> 3x3 - wino_wgrad_finish_kernel."""
    _, G, _ = matrices(tile)
    return _apply(G.T, _apply(G.T, du, -1), -2)


def _patches(x, tile=2):
    """x (B, C, H, W) -> zero-padded a x a patches (B, th, tw, C, a, a) at origins (tile*ty-1, tile*tx-1), th = ceil(H/tile)."""
    B, C, H, W = x.shape
    a = tile + 2
    th, tw = (H + tile - 1) // tile, (W + tile - 1) // tile
    xp = np.zeros((B, C, tile * th + 2, tile * tw + 2), dtype=x.dtype)
    xp[:, :, 1:H + 1, 1:W + 1] = x
    out = np.empty((B, th, tw, C, a, a), dtype=x.dtype)
    for ty in range(th):
        for tx in range(tw):
            out[:, ty, tx] = xp[:, :, tile * ty:tile * ty + a, tile * tx:tile * tx + a]
    return out


def conv3x3(x, w, tile=2):
    """'same' 3x3 cross-correlation of x (B, Cin, H, W) with w (Cout, Cin, 3, 3) through the Winograd domain."""
    B, _, H, W = x.shape
    V = input_transform(_patches(x, tile), tile)                # (B, th, tw, Cin, a, a)
    U = filter_transform(w, tile)                               # (Cout, Cin, a, a)
    M = np.einsum('btscij,kcij->btskij', V, U)                  # (tile+2)^2 GEMMs over the channels
    Y = output_transform(M, tile)                               # (B, th, tw, Cout, tile, tile)
    th, tw = Y.shape[1], Y.shape[2]
    out = Y.transpose(0, 3, 1, 4, 2, 5).reshape(B, w.shape[0], tile * th, tile * tw)
    return out[:, :, :H, :W]


def conv3x3_wgrad(x, dy, tile=2):
    """Filter gradient (Cout, Cin, 3, 3) of that convolution for the output gradient dy (B, Cout, H, W)."""
    B, Cout, H, W = dy.shape
    th, tw = (H + tile - 1) // tile, (W + tile - 1) // tile
    V = input_transform(_patches(x, tile), tile)
    dyp = np.zeros((B, Cout, tile * th, tile * tw), dtype=dy.dtype)
    dyp[:, :, :H, :W] = dy
    tiles = dyp.reshape(B, Cout, th, tile, tw, tile).transpose(0, 2, 4, 1, 3, 5)      # (B, th, tw, Cout, tile, tile)
    dM = outgrad_transform(tiles, tile)
    dU = np.einsum('btskij,btscij->kcij', dM, V)                # (tile+2)^2 GEMMs over the tiles
    return filtergrad_transform(dU, tile)
