#!/usr/bin/env python
"""Winograd transform / finishing passes alone (HIP-event time of the SSP_PROF_WINO_* families inside the conv entry points),
per layer shape and `wino_variant` option (bit 0: 4 channels per thread in the F(4x4) input / output-gradient transforms,
bit 1: non-temporal plane stores).  GPU only.   python tools/wino_xform_bench.py [--variants 0,1,2,3] [--plan 8006413]"""
import argparse, ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from singleshotpose_amd import _lib
from conv_bench import CASES


def collect(nk):
    ms = (ctypes.c_double * nk)(); work = (ctypes.c_double * nk)(); cnt = (ctypes.c_int64 * nk)()
    _lib.call('ssp_prof_collect', ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(work, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p))
    return list(ms), list(work), list(cnt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', default='l4,l8,l12,l18,l23')
    ap.add_argument('--variants', default='0,1,2,3')
    ap.add_argument('--plan', type=int, default=8006413)
    ap.add_argument('--B', type=int, default=64)
    ap.add_argument('--iters', type=int, default=10)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    st = torch.cuda.current_stream().cuda_stream
    nk = _lib.query('ssp_prof_nkinds')
    B = args.B
    tile = _lib.query('ssp_conv_plan_wino_tile', args.plan)
    for name in args.cases.split(','):
        H, Cin, Cout, R = CASES[name]
        W, M = H, B * H * H
        g = torch.Generator(device='cpu').manual_seed(1)
        x = (torch.rand(M * Cin, generator=g) * 2 - 1).to(dev)
        dy = (torch.rand(M * Cout, generator=g) * 2 - 1).to(dev)
        w = ((torch.rand(Cout * 9 * Cin, generator=g) * 2 - 1) * 0.05).to(dev)
        out = torch.empty(M * Cout, device=dev)
        dw = torch.zeros(Cout * 9 * Cin, device=dev)
        U = torch.empty((tile + 2) ** 2 * Cout * Cin, device=dev)
        _lib.call('ssp_wino_filter_transform_t', w.data_ptr(), U.data_ptr(), Cout, Cin, tile, st)
        wsn = _lib.query('ssp_conv_workspace_floats', B, H, W, Cin, Cout, 3, args.plan)
        ws = torch.empty(wsn, device=dev)
        wsw = torch.empty(_lib.query('ssp_conv_wgrad_wino_workspace_floats_t', B, H, W, Cin, Cout, tile), device=dev)
        stats = torch.empty(_lib.query('ssp_conv_stats_floats', B, H, W, Cin, Cout, 3, args.plan), device=dev)
        for v in [int(t) for t in args.variants.split(',')]:
            _lib.call('ssp_set_option', b'wino_variant', v)
            def fwd():
                _lib.call('ssp_conv_fwd', x.data_ptr(), U.data_ptr(), out.data_ptr(), None, stats.data_ptr(), B, H, W, Cin, Cout, Cin, Cout,
                          3, 0, args.plan, ws.data_ptr(), wsn, st)
            def wg():
                _lib.call('ssp_conv_wgrad_wino_t', dy.data_ptr(), x.data_ptr(), dw.data_ptr(), B, H, W, Cin, Cout, Cout, Cin, tile,
                          wsw.data_ptr(), wsw.numel(), st)
            for _ in range(2):
                fwd(); wg()
            torch.cuda.synchronize()
            _lib.call('ssp_prof_enable', (1 << 0) | (1 << 2) | (1 << 9) | (1 << 11))
            for _ in range(args.iters):
                fwd(); wg()
            torch.cuda.synchronize()
            _lib.call('ssp_prof_enable', 0)
            ms, work, cnt = collect(nk)
            n = args.iters
            print('%-4s variant %d | fwd unit %7.1f us, its transform+finish passes %7.1f us (%5.0f GB/s) | wgrad unit %7.1f us, its passes %7.1f us (%5.0f GB/s)'
                  % (name, v, ms[0] / n * 1e3, ms[9] / n * 1e3, work[9] / max(ms[9], 1e-9) / 1e6, ms[2] / n * 1e3, ms[11] / n * 1e3,
                     work[11] / max(ms[11], 1e-9) / 1e6), flush=True)
        _lib.call('ssp_set_option', b'wino_variant', 0)


if __name__ == '__main__':
    main()
