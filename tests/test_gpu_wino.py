"""Winograd F(2x2, 3x3) and F(4x4, 3x3) plans of the conv entry points (csrc/conv_wino.hip) through the C ABI against PyTorch-CPU
F.conv2d (what nn.Conv2d runs in the reference, darknet.py:154-160) and against the library's direct plan: forward
(raw output, BatchNorm statistics, bias, accumulate, sliced output), the eval-mode affine form, the data gradient (plain,
accumulating, with the fused BatchNorm-backward reductions), odd and even maps, ragged tiles.  Same 1e-4 bar as every
other fp32 kernel; the distance to the direct plan is printed (the engine's verify-after-tune admits a plan at 1e-5)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
WINO = 9006413      # Winograd F(2x2), 64-row GEMM tiles, 3-slot ring
WINO4 = 8006413     # Winograd F(4x4), same GEMM tiles
FUSED = 7000001     # F(2x2) with the transform domain kept on the chip (csrc/conv_wino_fused.hip): one persistent launch


def _tile(plan):
    from singleshotpose_amd import _lib
    return _lib.query('ssp_conv_plan_wino_tile', plan)


def _imports():
    import gpu_util as G
    from singleshotpose_amd import _lib
    return G, _lib


def _wino_filters(G, _lib, w9, rows, K, tile=2):
    U = torch.empty((tile + 2) ** 2 * rows * K, dtype=torch.float32, device=G.dev())
    _lib.call('ssp_wino_filter_transform_t', w9.data_ptr(), U.data_ptr(), rows, K, tile, G.stream())
    return U


CASES = [
    # B, H, W, Cin, Cout, ldout_extra, bias, plan
    (2, 13, 13, 64, 128, 0, False, WINO),          # odd map: the last tile row / column is half outside
    (3, 14, 10, 128, 96, 32, True, WINO),          # even map, bias, sliced output, Cout not a tile multiple
    (1, 21, 21, 256, 256, 0, False, 9012814),      # valid.py's grid, 128-row tiles, 4-slot ring
    (5, 7, 9, 32, 160, 0, False, WINO),            # tiny odd map, ragged M
    (64, 13, 13, 256, 512, 0, False, WINO),        # benchmark grid: 3136 tiles per transform position
    (2, 1, 5, 64, 128, 0, False, WINO),            # a single image row
    (2, 13, 13, 64, 128, 0, False, WINO4),         # F(4x4): 13 = 3 tiles + 1 pixel: the last tile row / column holds one
    (3, 14, 10, 128, 96, 32, True, WINO4),         # bias, sliced output, ragged tiles both ways
    (1, 21, 21, 256, 256, 0, False, 8012814),      # valid.py's grid
    (5, 7, 9, 32, 160, 0, False, WINO4),
    (64, 13, 13, 256, 512, 0, False, WINO4),       # benchmark grid: 1024 tiles per transform position
    (2, 1, 5, 64, 128, 0, False, WINO4),
    (4, 52, 52, 128, 256, 0, False, WINO4),        # 13 x 13 exact tiles, the K = 128 GEMMs of layers 8 / 10
    # 2 x 2 image mosaics (four images tiled as one map with a zero row / column between them: fewer tiles)
    (7, 13, 13, 64, 128, 0, False, WINO4),         # two mosaics, the second with one phantom image
    (4, 21, 21, 128, 128, 16, True, WINO4),        # valid.py's grid: 121 tiles instead of 144; bias, sliced output
    (3, 9, 9, 32, 128, 0, False, WINO4),           # one mosaic, one phantom image
    (4, 13, 9, 64, 192, 0, False, WINO4),          # non-square
    # on-chip F(2x2): a wave = a patch of 4 x 8 tiles x 32 output channels, workgroups walk (patch block, channel block) items
    (2, 13, 13, 64, 128, 0, False, FUSED),         # odd map: half-empty tiles at the right / bottom edge, masked stores
    (3, 14, 10, 128, 96, 32, True, FUSED),         # bias, sliced output, three channel blocks
    (1, 21, 21, 256, 256, 0, False, FUSED),        # valid.py's grid; K loop of 8 stage pairs
    (5, 7, 9, 32, 160, 0, False, FUSED),           # one stage pair (Cin = 32), tiny ragged patches
    (64, 13, 13, 256, 512, 0, False, FUSED),       # benchmark grid: 4096 items, 16 per workgroup
    (2, 1, 5, 64, 128, 0, False, FUSED),           # a single image row
    (4, 104, 104, 64, 128, 0, False, FUSED),       # layer 4 / 6: 13 x 6.5 patches per image, several items per workgroup
    (2, 208, 208, 32, 64, 0, False, FUSED),        # layer 2
    (3, 24, 40, 64, 64, 0, False, FUSED),          # patches that tile the map exactly, fewer items than workgroups
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,xout,bias,plan", CASES)
def test_wino_conv_fwd(B, H, W, Cin, Cout, xout, bias, plan):
    G, _lib = _imports()
    rs = np.random.RandomState(Cin + 3 * Cout + H)
    x = torch.from_numpy(rs.standard_normal((B, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rs.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32))
    bvec = torch.from_numpy(rs.standard_normal(Cout).astype(np.float32)) if bias else None
    ref = F.conv2d(x, w, bvec, padding=1)
    ldout = Cout + xout
    off = xout // 2
    xd = G.to_nhwc(x)
    wd = G.pack_fwd(w)
    U = _wino_filters(G, _lib, wd, Cout, Cin, _tile(plan))
    bd = bvec.to(G.dev()) if bias else None
    M = B * H * W

    def run(code, wt, accumulate=0, stats=None, out=None):
        wsn = max(1, _lib.query('ssp_conv_workspace_floats', B, H, W, Cin, Cout, 3, code))
        ws = torch.empty(wsn, dtype=torch.float32, device=G.dev())
        if out is None:
            out = torch.full((M, ldout), float('nan'), dtype=torch.float32, device=G.dev())
        _lib.call('ssp_conv_fwd', xd.data_ptr(), wt.data_ptr(), G.p(out, off), bd.data_ptr() if bias else None,
                  stats.data_ptr() if stats is not None else None, B, H, W, Cin, Cout, Cin, ldout, 3, accumulate, code,
                  ws.data_ptr(), wsn, G.stream())
        torch.cuda.synchronize()
        return out

    tile_m = _lib.query('ssp_conv_stats_tile_m', B, H, W, Cin, Cout, 3, plan)      # 0: counted format
    ntile = _lib.query('ssp_conv_stats_tiles', B, H, W, Cin, Cout, 3, plan)
    stats = torch.zeros(_lib.query('ssp_conv_stats_floats', B, H, W, Cin, Cout, 3, plan), dtype=torch.float32, device=G.dev())
    out = run(plan, U, stats=stats)
    got = G.from_nhwc(out, B, Cout, H, W, off)
    direct = G.from_nhwc(run(0, wd), B, Cout, H, W, off)
    print('winograd vs F.conv2d %.2e, vs the direct plan %.2e' % (rel_err(got.numpy(), ref.numpy()),
                                                                 rel_err(got.numpy(), direct.numpy())))
    assert rel_err(got.numpy(), ref.numpy()) < TOL
    if xout:
        o = out.cpu()
        assert torch.isnan(o[:, :off]).all() and torch.isnan(o[:, off + Cout:]).all()
    got2 = G.from_nhwc(run(plan, U, accumulate=1, out=out), B, Cout, H, W, off)
    assert rel_err(got2.numpy(), (2 * ref).numpy()) < TOL
    if not bias:      # training-mode BatchNorm statistics from the finishing pass's per-tile partials
        vec = torch.zeros(4, Cout, dtype=torch.float32, device=G.dev())
        ones, zeros = torch.ones(Cout, device=G.dev()), torch.zeros(Cout, device=G.dev())
        rm, rv = torch.zeros(Cout, device=G.dev()), torch.ones(Cout, device=G.dev())
        _lib.call('ssp_bn_fwd_finalize', stats.data_ptr(), ntile, tile_m, M, Cout, ones.data_ptr(), zeros.data_ptr(),
                  rm.data_ptr(), rv.data_ptr(), 0.1, 1e-4, vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(),
                  vec[3].data_ptr(), G.stream())
        torch.cuda.synchronize()
        r64 = ref.double()
        np.testing.assert_allclose(vec[0].cpu().numpy(), r64.mean(dim=(0, 2, 3)).numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(vec[1].cpu().numpy(),
                                   (1 / torch.sqrt(r64.var(dim=(0, 2, 3), unbiased=False) + 1e-4)).numpy(), rtol=1e-4)


@pytest.mark.parametrize("WINO", [WINO, WINO4, FUSED])
def test_wino_conv_fwd_affine_eval_block(WINO):
    """Inference form: BatchNorm affine + leaky applied by the finishing pass (ssp_conv_fwd_affine with a Winograd plan)."""
    G, _lib = _imports()
    B, H, W, Cin, Cout = 1, 21, 21, 128, 256
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.standard_normal((B, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rs.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32))
    sc = torch.from_numpy(rs.uniform(0.5, 1.5, Cout).astype(np.float32))
    sh = torch.from_numpy(rs.standard_normal(Cout).astype(np.float32))
    ref = F.leaky_relu(F.conv2d(x, w, None, padding=1) * sc[None, :, None, None] + sh[None, :, None, None], 0.1)
    xd, wd = G.to_nhwc(x), G.pack_fwd(w)
    U = _wino_filters(G, _lib, wd, Cout, Cin, _tile(WINO))
    wsn = _lib.query('ssp_conv_workspace_floats', B, H, W, Cin, Cout, 3, WINO)
    ws = torch.empty(wsn, dtype=torch.float32, device=G.dev())
    out = torch.empty(B * H * W, Cout, dtype=torch.float32, device=G.dev())
    scd, shd = sc.to(G.dev()), sh.to(G.dev())
    _lib.call('ssp_conv_fwd_affine', xd.data_ptr(), U.data_ptr(), out.data_ptr(), scd.data_ptr(), shd.data_ptr(), 0.1, B, H, W,
              Cin, Cout, Cin, Cout, 3, WINO, ws.data_ptr(), wsn, G.stream())
    torch.cuda.synchronize()
    assert rel_err(G.from_nhwc(out, B, Cout, H, W).numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("WINO", [WINO, WINO4, FUSED])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 13, 13, 128, 256), (3, 10, 14, 256, 128), (64, 13, 13, 128, 512), (7, 13, 9, 128, 128),
                                            (4, 104, 104, 64, 128)])
def test_wino_conv_dgrad(B, H, W, Cin, Cout, WINO):
    """Data gradient with a Winograd plan: filters from the ssp_repack_dgrad layout; plain, accumulating, and with the
    BatchNorm-backward reductions of the producing block folded into the finishing pass (ssp_conv_dgrad_bnbwd)."""
    G, _lib = _imports()
    rs = np.random.RandomState(Cin + Cout + H)
    x = torch.from_numpy(rs.standard_normal((B, Cin, H, W)).astype(np.float32)).requires_grad_(True)
    w = torch.from_numpy((rs.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32))
    dy = torch.from_numpy(rs.standard_normal((B, Cout, H, W)).astype(np.float32))
    F.conv2d(x, w, None, padding=1).backward(dy)
    want = x.grad
    dyd = G.to_nhwc(dy)
    wd = G.pack_dgrad(w, Cout)                                   # [Cin][tap'][Cout]
    U = _wino_filters(G, _lib, wd, Cin, Cout, _tile(WINO))
    M = B * H * W
    wsn = _lib.query('ssp_conv_workspace_floats', B, H, W, Cout, Cin, 3, WINO)
    ws = torch.empty(wsn, dtype=torch.float32, device=G.dev())
    dx = torch.full((M, Cin), float('nan'), dtype=torch.float32, device=G.dev())
    _lib.call('ssp_conv_dgrad', dyd.data_ptr(), U.data_ptr(), dx.data_ptr(), B, H, W, Cout, Cin, Cout, Cin, 3, 0, WINO,
              ws.data_ptr(), wsn, G.stream())
    torch.cuda.synchronize()
    assert rel_err(G.from_nhwc(dx, B, Cin, H, W).numpy(), want.numpy()) < TOL
    _lib.call('ssp_conv_dgrad', dyd.data_ptr(), U.data_ptr(), dx.data_ptr(), B, H, W, Cout, Cin, Cout, Cin, 3, 1, WINO,
              ws.data_ptr(), wsn, G.stream())
    torch.cuda.synchronize()
    assert rel_err(G.from_nhwc(dx, B, Cin, H, W).numpy(), (2 * want).numpy()) < TOL
    # fused BatchNorm-backward reductions: sum(dy'), sum(dy' * xhat) of the block that produced this conv's input, with
    # dy' = g * leaky'(scale * raw + shift); reference = the same sums from the CPU gradient
    raw = torch.from_numpy(rs.standard_normal((B, Cin, H, W)).astype(np.float32))
    scale = torch.from_numpy(rs.uniform(0.5, 1.5, Cin).astype(np.float32))
    shift = torch.from_numpy(rs.standard_normal(Cin).astype(np.float32) * 0.3)
    mean = torch.from_numpy(rs.standard_normal(Cin).astype(np.float32) * 0.1)
    invstd = torch.from_numpy(rs.uniform(0.5, 2.0, Cin).astype(np.float32))
    rows = _lib.query('ssp_conv_stats_tiles', B, H, W, Cout, Cin, 3, WINO)
    part = torch.zeros(rows * Cin * 2, dtype=torch.float32, device=G.dev())
    rawd = G.to_nhwc(raw)
    dv = [t.to(G.dev()) for t in (scale, shift, mean, invstd)]
    dx2 = torch.empty(M, Cin, dtype=torch.float32, device=G.dev())
    _lib.call('ssp_conv_dgrad_bnbwd', dyd.data_ptr(), U.data_ptr(), dx2.data_ptr(), B, H, W, Cout, Cin, Cout, Cin, 3, WINO,
              ws.data_ptr(), wsn, rawd.data_ptr(), Cin, dv[0].data_ptr(), dv[1].data_ptr(), dv[2].data_ptr(),
              dv[3].data_ptr(), 0.1, part.data_ptr(), rows, G.stream())
    torch.cuda.synchronize()
    assert rel_err(G.from_nhwc(dx2, B, Cin, H, W).numpy(), want.numpy()) < TOL
    g64 = want.double()
    y = raw.double() * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    dyp = torch.where(y > 0, g64, g64 * 0.1)
    xhat = (raw.double() - mean.double()[None, :, None, None]) * invstd.double()[None, :, None, None]
    s1, s2 = dyp.sum(dim=(0, 2, 3)), (dyp * xhat).sum(dim=(0, 2, 3))
    got = part.view(rows, Cin, 2).double().sum(dim=0).cpu()
    assert rel_err(got[:, 0].numpy(), s1.numpy()) < TOL and rel_err(got[:, 1].numpy(), s2.numpy()) < TOL


@pytest.mark.parametrize("tile", [2, 4, 12])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(4, 13, 13, 128, 256), (3, 10, 14, 256, 64), (64, 13, 13, 256, 512), (16, 7, 9, 64, 192),
                                            (64, 13, 13, 512, 1024), (7, 21, 13, 64, 128), (4, 104, 104, 64, 128), (2, 208, 208, 32, 64),
                                            (1, 5, 3, 32, 32)])
def test_wino_conv_wgrad(B, H, W, Cin, Cout, tile):
    """Filter gradient in the Winograd domain (ssp_conv_wgrad_wino) against autograd of F.conv2d and against the direct
    kernel; accumulates into dw (a second call doubles it).  tile 12 = F(2x2) with both transforms on the chip
    (csrc/conv_wino_wgrad_fused.hip: 32-channel granularity; the HBM forms need >= 64 channels)."""
    if tile != 12 and (min(Cin, Cout) < 64 or B * ((H + tile - 1) // tile) * ((W + tile - 1) // tile) < 16):
        pytest.skip("the HBM Winograd filter gradient needs >= 64 channels and >= 16 tiles")
    G, _lib = _imports()
    rs = np.random.RandomState(Cin * 5 + Cout + W)
    x = torch.from_numpy(rs.standard_normal((B, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rs.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)).requires_grad_(True)
    dy = torch.from_numpy(rs.standard_normal((B, Cout, H, W)).astype(np.float32))
    F.conv2d(x, w, None, padding=1).backward(dy)
    xd, dyd = G.to_nhwc(x), G.to_nhwc(dy)
    wsn = _lib.query('ssp_conv_wgrad_wino_workspace_floats_t', B, H, W, Cin, Cout, tile)
    # (NaN-poisoned: the transform-domain gradient inside is written, not accumulated, whatever the launch's split)
    ws = torch.full((wsn,), float('nan'), dtype=torch.float32, device=G.dev())
    dwp = torch.zeros(Cout * 9 * Cin, dtype=torch.float32, device=G.dev())

    def unpack(buf):
        gw = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=G.dev())
        _lib.call('ssp_unpack_grad', buf.data_ptr(), gw.data_ptr(), Cout, Cin, Cin, 3, G.stream())
        torch.cuda.synchronize()
        return gw.cpu().numpy()

    _lib.call('ssp_conv_wgrad_wino_t', dyd.data_ptr(), xd.data_ptr(), dwp.data_ptr(), B, H, W, Cin, Cout, Cout, Cin, tile,
              ws.data_ptr(), wsn, G.stream())
    got = unpack(dwp)
    direct = torch.zeros_like(dwp)
    _lib.call('ssp_conv_wgrad', dyd.data_ptr(), xd.data_ptr(), direct.data_ptr(), B, H, W, Cin, Cout, Cout, Cin, 3, G.stream())
    print('winograd wgrad vs autograd %.2e, vs the direct kernel %.2e' % (rel_err(got, w.grad.numpy()), rel_err(got, unpack(direct))))
    assert rel_err(got, w.grad.numpy()) < TOL
    _lib.call('ssp_conv_wgrad_wino_t', dyd.data_ptr(), xd.data_ptr(), dwp.data_ptr(), B, H, W, Cin, Cout, Cout, Cin, tile,
              ws.data_ptr(), wsn, G.stream())
    assert rel_err(unpack(dwp), 2 * w.grad.numpy()) < TOL


@pytest.mark.parametrize("WINO", [WINO, WINO4])
def test_wino_wgrad_reuses_the_forward_launch_transformed_input(WINO):
    """ssp_conv_wgrad_wino with x == NULL: the transformed input V left at the head of the workspace by the layer's own
    Winograd forward launch (same buffer) is used instead of transforming x again - same gradient as with x."""
    G, _lib = _imports()
    B, H, W, Cin, Cout = 8, 13, 13, 128, 256
    rs = np.random.RandomState(77)
    x = torch.from_numpy(rs.standard_normal((B, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rs.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)).requires_grad_(True)
    dy = torch.from_numpy(rs.standard_normal((B, Cout, H, W)).astype(np.float32))
    F.conv2d(x, w, None, padding=1).backward(dy)
    xd, dyd = G.to_nhwc(x), G.to_nhwc(dy)
    tile = _tile(WINO)
    wd = G.pack_fwd(w.detach())
    U = _wino_filters(G, _lib, wd, Cout, Cin, tile)
    wsn = max(_lib.query('ssp_conv_wgrad_wino_workspace_floats_t', B, H, W, Cin, Cout, tile),
              _lib.query('ssp_conv_workspace_floats', B, H, W, Cin, Cout, 3, WINO))
    ws = torch.full((wsn,), float('nan'), dtype=torch.float32, device=G.dev())
    out = torch.empty(B * H * W, Cout, dtype=torch.float32, device=G.dev())
    _lib.call('ssp_conv_fwd', xd.data_ptr(), U.data_ptr(), out.data_ptr(), None, None, B, H, W, Cin, Cout, Cin, Cout, 3, 0, WINO,
              ws.data_ptr(), wsn, G.stream())
    dwp = torch.zeros(Cout * 9 * Cin, dtype=torch.float32, device=G.dev())
    _lib.call('ssp_conv_wgrad_wino_t', dyd.data_ptr(), None, dwp.data_ptr(), B, H, W, Cin, Cout, Cout, Cin, tile, ws.data_ptr(), wsn,
              G.stream())
    gw = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=G.dev())
    _lib.call('ssp_unpack_grad', dwp.data_ptr(), gw.data_ptr(), Cout, Cin, Cin, 3, G.stream())
    torch.cuda.synchronize()
    assert rel_err(gw.cpu().numpy(), w.grad.numpy()) < TOL


@pytest.mark.parametrize("tile", [2, 4])
def test_wino_input_transform_alone_feeds_the_filter_gradient(tile):
    """ssp_wino_input_transform_t + ssp_conv_wgrad_wino_t(x == NULL): what the engine does for a layer whose forward runs a
    direct code while its filter gradient runs in the Winograd domain (the transform is queued during the forward pass, on
    the second stream) - the same gradient, bit for bit, as the launch that transforms x itself (a mosaic-tiled 13 x 13 map)."""
    G, _lib = _imports()
    B, H, W, Cin, Cout = 8, 13, 13, 64, 128
    rs = np.random.RandomState(5)
    xd = G.to_nhwc(torch.from_numpy(rs.standard_normal((B, Cin, H, W)).astype(np.float32)))
    dyd = G.to_nhwc(torch.from_numpy(rs.standard_normal((B, Cout, H, W)).astype(np.float32)))
    wsn = _lib.query('ssp_conv_wgrad_wino_workspace_floats_t', B, H, W, Cin, Cout, tile)
    res = []
    for early in (False, True):
        ws = torch.full((wsn,), float('nan'), dtype=torch.float32, device=G.dev())
        dw = torch.zeros(Cout * 9 * Cin, dtype=torch.float32, device=G.dev())
        if early:
            _lib.call('ssp_wino_input_transform_t', xd.data_ptr(), Cin, ws.data_ptr(), B, H, W, Cin, tile, G.stream())
        _lib.call('ssp_conv_wgrad_wino_t', dyd.data_ptr(), None if early else xd.data_ptr(), dw.data_ptr(), B, H, W, Cin, Cout, Cout,
                  Cin, tile, ws.data_ptr(), wsn, G.stream())
        torch.cuda.synchronize()
        res.append(dw.cpu())
    assert torch.isfinite(res[0]).all() and float(res[0].abs().max()) > 0
    assert rel_err(res[1].numpy(), res[0].numpy()) < 1e-6      # (the filter-gradient kernel's fp32 atomics: last-bit differences)


def test_wino_plan_on_a_shape_it_does_not_fit_is_an_error():
    G, _lib = _imports()
    x = torch.zeros(4 * 4 * 4, 64, device=G.dev())
    w = torch.zeros(16 * 128 * 64, device=G.dev())
    out = torch.zeros(4 * 4 * 4, 128, device=G.dev())
    ws = torch.zeros(1 << 20, device=G.dev())
    with pytest.raises(_lib.SspError):      # 1x1 filter
        _lib.call('ssp_conv_fwd', x.data_ptr(), w.data_ptr(), out.data_ptr(), None, None, 4, 4, 4, 64, 128, 64, 128, 1, 0,
                  WINO, ws.data_ptr(), 1 << 20, G.stream())
    with pytest.raises(_lib.SspError):      # no workspace
        _lib.call('ssp_conv_fwd', x.data_ptr(), w.data_ptr(), out.data_ptr(), None, None, 4, 4, 4, 64, 128, 64, 128, 3, 0,
                  WINO, None, 0, G.stream())
