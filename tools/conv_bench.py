#!/usr/bin/env python
"""Single-layer conv kernel timing through the C ABI (optimisation harness; GPU only).

  python tools/conv_bench.py [--cases l2,l8,l18,l23,l29] [--iters 20] [--ops fwd,dgrad,wgrad] [--B 64]
Prints TFLOP/s per (case, op) from torch.cuda events on the launch stream (median of iters).
"""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from singleshotpose_amd import _lib

# name: (H, Cin, Cout, R) of yolo-pose.cfg layers at 416x416
CASES = {
    'l0': (416, 4, 32, 3), 'l2': (208, 32, 64, 3), 'l4': (104, 64, 128, 3), 'l5': (104, 128, 64, 1),
    'l6': (104, 64, 128, 3), 'l8': (52, 128, 256, 3), 'l9': (52, 256, 128, 1), 'l12': (26, 256, 512, 3), 'l13': (26, 512, 256, 1),
    'l18': (13, 512, 1024, 3), 'l19': (13, 1024, 512, 1), 'l23': (13, 1024, 1024, 3), 'l29': (13, 1280, 1024, 3),
    'l30': (13, 1024, 20, 1), 'l26': (26, 512, 64, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', default='l2,l4,l8,l12,l18,l23,l29')
    ap.add_argument('--ops', default='fwd,dgrad,wgrad')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--B', type=int, default=64)
    ap.add_argument('--opt', default='', help='name=value,... passed to ssp_set_option')
    ap.add_argument('--wvariants', default='', help='comma list of wgrad_variant values to loop over')
    ap.add_argument('--variants', default='', help='comma list: run every case once per igemm_variant value')
    ap.add_argument('--plans', default='0', help='comma list of explicit plan codes for fwd / dgrad (0 = heuristic; 9006413 etc. = '
                                                 'Winograd: filters are transformed first, outside the timed launches)')
    ap.add_argument('--err', action='store_true', help='also print the forward / data-gradient error of every row against a float64 '
                                                       'evaluation of 4096 sampled output pixels (rms and max, relative to the output range)')
    ap.add_argument('--stats', default='unit', choices=['unit', 'net'], help="operand statistics: unit = uniform(-1, 1) everywhere; "
                    "net = what the network's layers see (leaky(N(0,1)) activations, N(0, 1/sqrt(fan-in)) filters)")
    ap.add_argument('--sweep', default='', help='option sets "name=value,...;name=value,..." (- = defaults): every case once per set, '
                                                'in one process (options of the previous set are reset to 0)')
    args = ap.parse_args()
    for kv in filter(None, args.opt.split(',')):
        k, v = kv.split('=')
        _lib.call('ssp_set_option', k.encode(), int(v))
    dev = torch.device('cuda', 0)
    st = torch.cuda.current_stream().cuda_stream
    B = args.B
    variants = [int(v) for v in args.variants.split(',')] if args.variants else [None]
    if args.sweep:
        touched = set()
        for cfg in args.sweep.split(';'):
            for name in touched:
                _lib.call('ssp_set_option', name.encode(), 1 if name in ('igemm_xcd', 'acc_chunk') else 0)
            touched = set()
            for kv in filter(None, (cfg if cfg != '-' else '').split(',')):
                k, v = kv.split('=')
                _lib.call('ssp_set_option', k.encode(), int(v))
                touched.add(k)
            print('SWEEP', cfg, flush=True)
            run_cases(args, dev, st, B)
        return
    if args.wvariants:
        for wv in [int(v) for v in args.wvariants.split(',')]:
            _lib.call('ssp_set_option', b'wgrad_variant', wv)
            print('WGRAD VARIANT', wv, flush=True)
            run_cases(args, dev, st, B)
        return
    for variant in variants:
      if variant is not None:
        _lib.call('ssp_set_option', b'igemm_variant', variant)
        print('VARIANT', variant, flush=True)
      run_cases(args, dev, st, B)


def err_refs(x, dy, w, wd, B, H, W, Cin, Cout, coutp, R, dev, n=4096):
    """float64 values of n sampled output pixels of the forward conv (x * w) and of the data gradient (dy * wd): the taps of a
    pixel are gathered with the zero padding and contracted with the [taps * K][N] filter matrix by a float64 matmul."""
    M = B * H * W
    g = torch.Generator(device='cpu').manual_seed(7)
    rows = torch.randperm(M, generator=g)[:min(n, M)].to(dev)
    b, yy, xx = rows // (H * W), (rows // W) % H, rows % W
    pad = R // 2

    def gather(src, K):
        s = src.view(B, H, W, K).double()
        cols = []
        for dy_ in range(R):
            for dx_ in range(R):
                y2, x2 = yy + dy_ - pad, xx + dx_ - pad
                ok = ((y2 >= 0) & (y2 < H) & (x2 >= 0) & (x2 < W)).double().unsqueeze(1)
                cols.append(s[b, y2.clamp(0, H - 1), x2.clamp(0, W - 1)] * ok)
        return torch.cat(cols, 1)
    fw = w.view(Cout, R * R * Cin).double()
    ref_f = gather(x, Cin) @ fw.t()
    fd = wd.view(Cin, R * R * coutp).double()
    ref_d = gather(dy, coutp) @ fd.t()
    return {'fwd': (rows, ref_f), 'dgrad': (rows, ref_d)}


def run_cases(args, dev, st, B):
    for name in args.cases.split(','):
        H, Cin, Cout, R = CASES[name]
        W = H
        M = B * H * W
        g = torch.Generator(device='cpu').manual_seed(1)
        coutp = (Cout + 3) // 4 * 4
        if args.stats == 'net':
            x = torch.randn(M * Cin, generator=g)
            x = torch.where(x > 0, x, 0.1 * x).to(dev)
            dy = torch.randn(M * coutp, generator=g).to(dev)
            w = (torch.randn(Cout * R * R * Cin, generator=g) / (R * R * Cin) ** 0.5).to(dev)
            wd = (torch.randn(Cin * R * R * coutp, generator=g) / (R * R * coutp) ** 0.5).to(dev)
        else:
            x = (torch.rand(M * Cin, generator=g) * 2 - 1).to(dev)
            dy = (torch.rand(M * coutp, generator=g) * 2 - 1).to(dev)
            w = (torch.rand(Cout * R * R * Cin, generator=g) * 2 - 1).to(dev) * 0.05
            wd = (torch.rand(Cin * R * R * coutp, generator=g) * 2 - 1).to(dev) * 0.05
        refs = err_refs(x, dy, w, wd, B, H, W, Cin, Cout, coutp, R, dev) if args.err else None
        out = torch.empty(M * coutp, device=dev)
        dx = torch.empty(M * Cin, device=dev)
        dw = torch.zeros(Cout * R * R * Cin, device=dev)
        stats = torch.empty(((M + 15) // 16 + 16) * (Cout * 2 + 1), device=dev)  # sized for the finest statistics tiling any plan uses
        flop = 2.0 * M * Cout * R * R * Cin
        for plan in [int(v) for v in args.plans.split(',')]:
            tile = _lib.query('ssp_conv_plan_wino_tile', plan)      # 2 / 4: Winograd F(2x2) / F(4x4) plan, 0: direct
            wino = tile > 0
            fused = 7000000 <= plan < 8000000      # on-chip F(2x2) (csrc/conv_wino_fused.hip): 32-channel granularity both ways
            if fused and (R != 3 or Cin % 32 or Cout % 32):
                continue
            if wino and not fused and (R != 3 or Cin % 16 or Cout % 16 or Cout < 64 or Cin < 64):
                continue
            wsn = max(1, _lib.query('ssp_conv_workspace_floats', B, H, W, Cin, Cout, R, plan), _lib.query('ssp_conv_workspace_floats', B, H, W, coutp, Cin, R, plan))
            ws = torch.empty(wsn, device=dev)
            wf, wdd = w, wd
            if wino:
                wf = torch.empty((tile + 2) ** 2 * Cout * Cin, device=dev)
                wdd = torch.empty((tile + 2) ** 2 * Cin * coutp, device=dev)
                _lib.call('ssp_wino_filter_transform_t', w.data_ptr(), wf.data_ptr(), Cout, Cin, tile, st)
                _lib.call('ssp_wino_filter_transform_t', wd.data_ptr(), wdd.data_ptr(), Cin, coutp, tile, st)
            fns = {
                'fwd': lambda: _lib.call('ssp_conv_fwd', x.data_ptr(), wf.data_ptr(), out.data_ptr(), None, stats.data_ptr(),
                                         B, H, W, Cin, Cout, Cin, coutp, R, 0, plan, ws.data_ptr(), wsn, st),
                'dgrad': lambda: _lib.call('ssp_conv_dgrad', dy.data_ptr(), wdd.data_ptr(), dx.data_ptr(), B, H, W, coutp, Cin,
                                           coutp, Cin, R, 0, plan, ws.data_ptr(), wsn, st),
                'wgrad': lambda: _lib.call('ssp_conv_wgrad', dy.data_ptr(), x.data_ptr(), dw.data_ptr(), B, H, W, Cin, Cout,
                                           coutp, Cin, R, st),
                # (the Winograd filter gradient of the tile size of THIS row's plan: direct rows skip it)
                'wgradw': lambda: _lib.call('ssp_conv_wgrad_wino_t', dy.data_ptr(), x.data_ptr(), dw.data_ptr(), B, H, W, Cin, Cout,
                                            coutp, Cin, tile, wsw.data_ptr(), wsw.numel(), st),
                # F(2x2) filter gradient with both transforms on the chip (csrc/conv_wino_wgrad_fused.hip)
                'wgradf': lambda: _lib.call('ssp_conv_wgrad_wino_t', dy.data_ptr(), x.data_ptr(), dw.data_ptr(), B, H, W, Cin, Cout,
                                            coutp, Cin, 12, wsf.data_ptr(), wsf.numel(), st),
                'wfilt': lambda: _lib.call('ssp_wino_filter_transform_t', w.data_ptr(), wf.data_ptr(), Cout, Cin, tile, st),
            }
            wsw = ws
            wsf = torch.empty(max(16, 16 * Cin * Cout), device=dev)
            if 'wgradw' in args.ops.split(',') and wino and R == 3 and Cin % 16 == 0 and Cout % 16 == 0 and Cin >= 64 and Cout >= 64:
                wsw = torch.empty(_lib.query('ssp_conv_wgrad_wino_workspace_floats_t', B, H, W, Cin, Cout, tile), device=dev)
            line = '%-4s H=%3d Cin=%4d Cout=%4d R=%d plan %7d |' % (name, H, Cin, Cout, R, plan)
            for op in args.ops.split(','):
                if op == 'wgradf' and (R != 3 or Cin % 32 or Cout % 32 or plan != int(args.plans.split(',')[0])):
                    continue
                if (op == 'dgrad' and name == 'l0') or (op == 'wfilt' and not wino) or (op == 'wgrad' and plan != int(args.plans.split(',')[0])) or (op == 'wgradw' and wsw is ws):
                    continue
                fn = fns[op]
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                ts = []
                for _ in range(args.iters):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn()
                    e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1))
                med = float(np.median(ts))
                line += ' %s %7.1f us %6.1f TF |' % (op, med * 1e3, flop / (med * 1e-3) / 1e12)
                if refs is not None and op in ('fwd', 'dgrad'):
                    rows, ref = refs[op]
                    got = (out.view(M, coutp)[rows, :Cout] if op == 'fwd' else dx.view(M, Cin)[rows]).double()
                    e = (got - ref) / ref.abs().max()
                    line += ' err rms %.2e max %.2e |' % (float(e.pow(2).mean().sqrt()), float(e.abs().max()))
            print(line, flush=True)


if __name__ == '__main__':
    main()
