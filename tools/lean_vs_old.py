#!/usr/bin/env python
"""Element-wise comparison of the lean (buffer-store) conv epilogues against the generic predicated ones
(igemm_variant=63) on given launch shapes / plans - debugging tool."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from singleshotpose_amd import _lib  # noqa: E402

dev = torch.device('cuda', 0)
st = torch.cuda.current_stream().cuda_stream


def run(kind, B, H, W, Cin, Cout, R, plan, variant, x, w, accumulate=0, prefill=None):
    _lib.call('ssp_set_option', b'igemm_variant', variant)
    M = B * H * W
    wsn = max(1, _lib.query('ssp_conv_workspace_floats', B, H, W, Cin, Cout, R, plan))
    ws = torch.zeros(wsn, device=dev)
    out = torch.full((M * Cout,), 7.0, device=dev) if prefill is None else prefill.clone()
    if kind == 'fwd':
        _lib.call('ssp_conv_fwd', x.data_ptr(), w.data_ptr(), out.data_ptr(), None, None, B, H, W, Cin, Cout, Cin, Cout, R,
                  accumulate, plan, ws.data_ptr(), wsn, st)
    else:
        _lib.call('ssp_conv_dgrad', x.data_ptr(), w.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, Cin, Cout, R, accumulate,
                  plan, ws.data_ptr(), wsn, st)
    torch.cuda.synchronize()
    _lib.call('ssp_set_option', b'igemm_variant', 0)
    return out


def main():
    g = torch.Generator(device='cuda').manual_seed(0)
    cases = [('dgrad', 64, 104, 104, 64, 128, 1, 6413), ('dgrad', 64, 104, 104, 64, 128, 1, 12813),
             ('dgrad', 64, 13, 13, 1024, 1280, 3, 306413), ('fwd', 64, 13, 13, 1024, 1024, 3, 306413),
             ('dgrad', 64, 52, 52, 256, 128, 3, 306414), ('fwd', 64, 52, 52, 128, 256, 3, 12813),
             ('dgrad', 64, 26, 26, 512, 256, 3, 306413), ('dgrad', 64, 13, 13, 1024, 512, 3, 12834),
             ('dgrad', 64, 13, 13, 512, 1024, 1, 206413), ('dgrad', 64, 26, 26, 64, 512, 1, 6413)]
    for kind, B, H, W, Cin, Cout, R, plan in cases:
        M = B * H * W
        x = torch.empty(M * Cin, device=dev).uniform_(-1, 1, generator=g)
        w = torch.empty(Cout * R * R * Cin, device=dev).uniform_(-0.05, 0.05, generator=g)
        for acc in (0, 1):
            pre = torch.empty(M * Cout, device=dev).uniform_(-1, 1, generator=g) if acc else None
            a = run(kind, B, H, W, Cin, Cout, R, plan, 63, x, w, acc, pre)
            b = run(kind, B, H, W, Cin, Cout, R, plan, 0, x, w, acc, pre)
            d = (a != b).view(M, Cout)
            n = int(d.sum())
            msg = ''
            if n:
                idx = d.nonzero()
                rows, cols = idx[:, 0], idx[:, 1]
                msg = ' rows%%128 %s cols%%128 %s first %s maxdiff %.3g' % (
                    sorted(set((rows % 128).tolist()))[:20], sorted(set((cols % 128).tolist()))[:20], idx[:3].tolist(),
                    float((a - b).abs().max()))
            print('%-5s B%d %dx%d %d->%d R%d plan %d acc %d: %d mismatching of %d%s' % (kind, B, H, W, Cin, Cout, R, plan, acc, n,
                                                                                        M * Cout, msg))


if __name__ == '__main__':
    main()
