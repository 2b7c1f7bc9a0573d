#!/usr/bin/env python
"""Where the rounding error of a Winograd F(4x4, 3x3) layer comes from - a CPU simulation (numpy, no GPU).

  python tools/wino_error_budget.py [K] [chunk ...]        e.g.  python tools/wino_error_budget.py 1024 64 128 144 256

One 6x6 input patch per tile, K input channels, float32 operands with the network's statistics (leaky(N(0,1)) activations,
N(0, 1/sqrt(9K)) filters), everything compared with the same arithmetic in float64.  Three questions:
  1. does the precision of the TRANSFORMS matter (B^T d B, G g G^T, A^T M A in float32 or float64)?          - no (1 %)
  2. does ROUNDING V and U to float32 matter (what the GEMM reads)?                                            - 5e-8, no
  3. the fp32 ACCUMULATION chain of the GEMM (the MFMA adds every product into one running sum)?              - all of it
and what chunked accumulation (conv_igemm_dma.hip, FL > 0: the K loop in groups, group sums added to a second register
set) does to it, for the Winograd planes and for a direct 3x3 convolution.  Numbers this printed (K = 1024, rms / max error
relative to the output range): F(4x4) one chain 8.9e-7 / 6.5e-6, groups of 128: 3.4e-7 / 2.0e-6; direct one chain
4.3e-7 / 3.2e-6, groups of 128: 6.5e-8 / 3.4e-7.
"""
import os
import sys
from fractions import Fraction as Fr

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.wino_ref import POINTS, cook_toom      # noqa: E402  (exact matrices from the interpolation points)


def mats(points, n=4):
    AT, G, BT = cook_toom(points, n)
    f = lambda m: np.array([[float(x) for x in row] for row in m], dtype=np.float64)
    return f(BT), f(G), f(AT)


def fma32(acc, a, c):
    """one fused multiply-add rounded to float32 (the product is exact in float64)"""
    return (acc.astype(np.float64) + a.astype(np.float64) * c).astype(np.float32)


def transform(mat, x, f32):
    """mat applied to the last two axes (rows, then columns) by sequential multiply-adds in float32 or float64"""
    def one(x, axis):
        x = np.moveaxis(x, axis, 0)
        out = np.zeros((mat.shape[0],) + x.shape[1:], dtype=np.float32 if f32 else np.float64)
        for i in range(mat.shape[0]):
            acc = np.zeros(x.shape[1:], dtype=out.dtype)
            for k in range(mat.shape[1]):
                if mat[i, k] != 0:
                    acc = fma32(acc, x[k], mat[i, k]) if f32 else acc + x[k] * mat[i, k]
            out[i] = acc
        return np.moveaxis(out, 0, axis)
    return one(one(x.astype(np.float32 if f32 else np.float64), -2), -1)


def chain(V, U, chunk=None):
    """sum_c V[t, c] * U[k, c] as the MFMA forms it: one fp32 running sum per output, optionally in groups of `chunk`"""
    T, K = V.shape[:2]
    acc = np.zeros((T, U.shape[0]) + V.shape[2:], dtype=np.float32)
    tot = np.zeros_like(acc)
    for c in range(K):
        acc = fma32(acc, V[:, None, c], U[None, :, c].astype(np.float64))
        if chunk and (c + 1) % chunk == 0:
            tot = (tot + acc).astype(np.float32)
            acc = np.zeros_like(acc)
    return (tot + acc).astype(np.float32) if chunk else acc


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    chunks = [int(c) for c in sys.argv[2:]] or [64, 128]
    T, Co = 32, 32
    BT, G, AT = mats(POINTS[4])
    rs = np.random.RandomState(0)
    x = rs.randn(T, K, 6, 6)
    x = np.where(x > 0, x, 0.1 * x).astype(np.float32)
    w = (rs.randn(Co, K, 3, 3) / np.sqrt(9 * K)).astype(np.float32)
    V64, U64 = transform(BT, x, False), transform(G, w, False)
    M64 = np.einsum('tcij,kcij->tkij', V64, U64)
    Y64 = transform(AT, M64, False)
    rng = np.abs(Y64).max()

    def rep(name, Y):
        e = Y.astype(np.float64) - Y64
        print('%-58s rms %.2e  max %.2e' % (name, np.sqrt((e ** 2).mean()) / rng, np.abs(e).max() / rng))

    print('F(4x4, 3x3), points %s, K = %d' % (POINTS[4], K))
    V32, U32 = transform(BT, x, True), transform(G, w, True)
    rep('all float32 (the kernels)', transform(AT, chain(V32, U32), True))
    rep('float64 transforms, float32 GEMM chain', transform(AT, chain(V64.astype(np.float32), U64.astype(np.float32)), False))
    rep('V, U rounded to float32, exact accumulation',
        transform(AT, np.einsum('tcij,kcij->tkij', V64.astype(np.float32).astype(np.float64), U64.astype(np.float32).astype(np.float64)), False))
    for c in chunks:
        rep('all float32, accumulation in groups of K = %d' % c, transform(AT, chain(V32, U32, c), True))
    print('|M| / |Y| (rms) per transform-domain position:')
    print(np.round(np.sqrt((M64 ** 2).mean(axis=(0, 1))) / np.sqrt((Y64 ** 2).mean()), 2))

    # the direct 3x3 convolution of the same patches (4 x 4 outputs): one chain of 9 K products
    Yd64 = np.zeros((T, Co, 4, 4))
    for a in range(3):
        for b in range(3):
            Yd64 += np.einsum('tcpq,kc->tkpq', x[:, :, a:a + 4, b:b + 4].astype(np.float64), w[:, :, a, b].astype(np.float64))
    for chunk in [None] + chunks:
        acc = np.zeros((T, Co, 4, 4), dtype=np.float32)
        tot = np.zeros_like(acc)
        n = 0
        for a in range(3):
            for b in range(3):
                for c in range(K):
                    acc = fma32(acc, x[:, None, c, a:a + 4, b:b + 4], w[None, :, c, a, b, None, None].astype(np.float64))
                    n += 1
                    if chunk and n % chunk == 0:
                        tot = (tot + acc).astype(np.float32)
                        acc = np.zeros_like(acc)
        e = (tot + acc).astype(np.float32) - Yd64
        print('%-58s rms %.2e  max %.2e' % ('direct 3x3, %s' % ('groups of %d' % chunk if chunk else 'one chain of %d' % (9 * K)),
                                            np.sqrt((e ** 2).mean()) / np.abs(Yd64).max(), np.abs(e).max() / np.abs(Yd64).max()))


if __name__ == '__main__':
    main()
