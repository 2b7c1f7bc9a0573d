"""Per-kernel parity through the C ABI (libssp_hip.so) against PyTorch-CPU fp32 references of the same op.

Tolerance: SURVEY.md section 8 / BASELINE.json: bit-exact for index-only kernels (reorg, route copy, repack, max-pool
selection), max|a-b|/max|b| <= 1e-4 for fp32 arithmetic.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _imports():
    import gpu_util as G
    from singleshotpose_amd import _lib
    return G, _lib


CONV_CASES = [
    # B, H, W, Cin, Cout, R, ldin_extra, ldout_extra, bias
    (2, 13, 13, 64, 128, 3, 0, 0, False),
    (1, 20, 24, 3, 32, 3, 0, 0, False),      # first layer: Cin 3 padded to 4, BK=4 path, 256x32 tile
    (2, 10, 12, 128, 64, 1, 0, 0, False),    # 256x64 tile
    (3, 7, 9, 32, 20, 1, 0, 0, True),        # head: bias, ragged Cout
    (2, 13, 13, 48, 160, 3, 16, 32, False),  # channel slices of wider buffers; Cout not a tile multiple
    (1, 26, 26, 20, 1024, 3, 0, 0, False),   # K chunk of 4 (Cin % 16 != 0), many N tiles
    (5, 13, 13, 256, 256, 3, 0, 0, False),   # several M tiles + ragged M
    (64, 13, 13, 128, 1024, 3, 0, 0, False), # 680 tiles: split-K path + splitk_reduce statistics
    (64, 13, 13, 64, 1024, 3, 0, 32, True),  # split-K with bias and a sliced output
    (64, 13, 13, 1024, 20, 1, 0, 0, True),   # the head conv at the benchmark batch: 43 thin 256-row tiles, split-K x8 + bias
    (1, 42, 42, 512, 64, 1, 0, 0, False),    # route conv at batch 1, 672 x 672: 14 thin tiles, split-K x4 + statistics
    (1, 21, 21, 1024, 20, 1, 0, 4, True),    # thin split declined (unaligned output slice): un-split fallback
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,R,xin,xout,bias", CONV_CASES)
def test_conv_fwd(B, H, W, Cin, Cout, R, xin, xout, bias):
    G, _lib = _imports()
    rs = np.random.RandomState(Cin * 7 + Cout)
    x = torch.from_numpy(rs.standard_normal((B, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rs.standard_normal((Cout, Cin, R, R)) / np.sqrt(Cin * R * R)).astype(np.float32))
    bvec = torch.from_numpy(rs.standard_normal(Cout).astype(np.float32)) if bias else None
    ref = F.conv2d(x, w, bvec, padding=R // 2)
    cinp = (Cin + 3) // 4 * 4
    xp = torch.zeros(B, cinp, H, W)
    xp[:, :Cin] = x
    ldin, ldout = cinp + xin, Cout + xout
    xin_off, xout_off = (xin // 2) // 4 * 4, xout // 2
    xd = G.to_nhwc(xp, ldin, xin_off)
    wd = G.pack_fwd(w, cinp)
    out = torch.full((B * H * W, ldout), float('nan'), dtype=torch.float32, device=G.dev())
    bd = bvec.to(G.dev()) if bias else None
    tile_m = _lib.query('ssp_conv_stats_tile_m', B, H, W, cinp, Cout, R, 0)
    wsn = max(1, _lib.query('ssp_conv_workspace_floats', B, H, W, cinp, Cout, R, 0))
    ws = torch.empty(wsn, dtype=torch.float32, device=G.dev())
    ntile = (B * H * W + tile_m - 1) // tile_m
    stats = torch.zeros(ntile * Cout * 2, dtype=torch.float32, device=G.dev())
    _lib.call('ssp_conv_fwd', G.p(xd, xin_off), wd.data_ptr(), G.p(out, xout_off), bd.data_ptr() if bias else None,
              stats.data_ptr(), B, H, W, cinp, Cout, ldin, ldout, R, 0, 0, ws.data_ptr(), wsn, G.stream())
    torch.cuda.synchronize()
    got = G.from_nhwc(out, B, Cout, H, W, xout_off)
    assert rel_err(got.numpy(), ref.numpy()) < TOL
    # untouched columns stay NaN: the kernel writes only its channel slice
    if xout:
        o = out.cpu()
        assert torch.isnan(o[:, :xout_off]).all() and torch.isnan(o[:, xout_off + Cout:]).all()
    # accumulate mode
    _lib.call('ssp_conv_fwd', G.p(xd, xin_off), wd.data_ptr(), G.p(out, xout_off), bd.data_ptr() if bias else None,
              None, B, H, W, cinp, Cout, ldin, ldout, R, 1, 0, ws.data_ptr(), wsn, G.stream())
    torch.cuda.synchronize()
    got2 = G.from_nhwc(out, B, Cout, H, W, xout_off)
    assert rel_err(got2.numpy(), (2 * ref).numpy()) < TOL

    # BatchNorm statistics from the epilogue partials (bias-free raw output)
    if not bias:
        M = B * H * W
        vec = torch.zeros(4, Cout, dtype=torch.float32, device=G.dev())
        gamma = torch.from_numpy(rs.uniform(0.5, 1.5, Cout).astype(np.float32))
        beta = torch.from_numpy(rs.standard_normal(Cout).astype(np.float32))
        rm = torch.from_numpy(rs.standard_normal(Cout).astype(np.float32))
        rv = torch.from_numpy(rs.uniform(0.5, 1.5, Cout).astype(np.float32))
        gd, bd2, rmd, rvd = gamma.to(G.dev()), beta.to(G.dev()), rm.clone().to(G.dev()), rv.clone().to(G.dev())
        _lib.call('ssp_bn_fwd_finalize', stats.data_ptr(), ntile, tile_m, M, Cout, gd.data_ptr(), bd2.data_ptr(),
                  rmd.data_ptr(), rvd.data_ptr(), 0.1, 1e-4, vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(),
                  vec[3].data_ptr(), G.stream())
        torch.cuda.synchronize()
        r64 = ref.double()
        mean = r64.mean(dim=(0, 2, 3))
        var = r64.var(dim=(0, 2, 3), unbiased=False)
        np.testing.assert_allclose(vec[0].cpu().numpy(), mean.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(vec[1].cpu().numpy(), (1 / torch.sqrt(var + 1e-4)).numpy(), rtol=1e-4)
        rm_ref, rv_ref = rm.clone(), rv.clone()
        F.batch_norm(ref, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-4)
        np.testing.assert_allclose(rmd.cpu().numpy(), rm_ref.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(rvd.cpu().numpy(), rv_ref.numpy(), rtol=1e-4, atol=1e-5)


GRAD_CASES = [
    (2, 13, 13, 64, 128, 3),
    (1, 20, 24, 3, 32, 3),      # first layer (wgrad only)
    (2, 10, 12, 128, 64, 1),
    (3, 7, 9, 32, 20, 1),       # head: dY with 20 channels
    (2, 12, 12, 32, 64, 3),     # 64x32 wgrad tile with in-workgroup K split
    (4, 13, 13, 256, 256, 3),
    (2, 26, 26, 512, 64, 1),
    (64, 13, 13, 64, 512, 3),   # dgrad/wgrad at the 13x13 benchmark grid (split-K dgrad)
    (4, 3, 3, 64, 64, 3),       # tiny maps: fewer pixels than one staged chunk, W < 8
    (3, 9, 11, 128, 64, 3),     # W < 16: the LDS-direct wgrad loader wraps image rows twice per chunk
    (1, 9, 11, 32, 128, 3),     # Cin 32: two filter taps folded into one 64-column wgrad tile (128-cout tile)
    (2, 16, 20, 32, 64, 3),     # same, 64-cout tile (layer 2 of yolo-pose.cfg)
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,R", GRAD_CASES)
def test_conv_dgrad_wgrad(B, H, W, Cin, Cout, R):
    G, _lib = _imports()
    rs = np.random.RandomState(Cin * 3 + Cout + R)
    x = torch.from_numpy(rs.standard_normal((B, Cin, H, W)).astype(np.float32)).requires_grad_(True)
    w = torch.from_numpy((rs.standard_normal((Cout, Cin, R, R)) / np.sqrt(Cin * R * R)).astype(np.float32)).requires_grad_(True)
    dy = torch.from_numpy(rs.standard_normal((B, Cout, H, W)).astype(np.float32))
    F.conv2d(x, w, None, padding=R // 2).backward(dy)
    cinp, coutp = (Cin + 3) // 4 * 4, (Cout + 3) // 4 * 4
    xp = torch.zeros(B, cinp, H, W)
    xp[:, :Cin] = x.detach()
    xd = G.to_nhwc(xp)
    dyp = torch.zeros(B, coutp, H, W)
    dyp[:, :Cout] = dy
    dyd = G.to_nhwc(dyp)
    # wgrad
    dwp = torch.zeros(Cout * R * R * cinp, dtype=torch.float32, device=G.dev())
    _lib.call('ssp_conv_wgrad', dyd.data_ptr(), xd.data_ptr(), dwp.data_ptr(), B, H, W, cinp, Cout, coutp, cinp, R, G.stream())
    gw = torch.empty(Cout, Cin, R, R, dtype=torch.float32, device=G.dev())
    _lib.call('ssp_unpack_grad', dwp.data_ptr(), gw.data_ptr(), Cout, Cin, cinp, R, G.stream())
    torch.cuda.synchronize()
    assert rel_err(gw.cpu().numpy(), w.grad.numpy()) < TOL
    # dgrad
    if Cin % 4 == 0:
        wd = G.pack_dgrad(w.detach(), coutp)
        dx = torch.full((B * H * W, Cin), float('nan'), dtype=torch.float32, device=G.dev())
        wsn = max(1, _lib.query('ssp_conv_workspace_floats', B, H, W, coutp, Cin, R, 0))
        ws = torch.empty(wsn, dtype=torch.float32, device=G.dev())
        _lib.call('ssp_conv_dgrad', dyd.data_ptr(), wd.data_ptr(), dx.data_ptr(), B, H, W, coutp, Cin, coutp, Cin, R, 0, 0, ws.data_ptr(), wsn, G.stream())
        torch.cuda.synchronize()
        assert rel_err(G.from_nhwc(dx, B, Cin, H, W).numpy(), x.grad.numpy()) < TOL


def test_repack_bit_exact():
    G, _lib = _imports()
    rs = np.random.RandomState(0)
    for (co, ci, k, cip) in [(20, 12, 3, 12), (32, 3, 3, 4), (70, 100, 1, 100), (33, 65, 3, 68)]:
        w = torch.from_numpy(rs.standard_normal((co, ci, k, k)).astype(np.float32))
        got = G.pack_fwd(w, cip).cpu().reshape(co, k * k, cip)
        ref = torch.zeros(co, k * k, cip)
        ref[:, :, :ci] = w.reshape(co, ci, k * k).permute(0, 2, 1)
        assert torch.equal(got, ref)
        back = torch.empty(co, ci, k, k, dtype=torch.float32, device=G.dev())
        _lib.call('ssp_unpack_grad', G.pack_fwd(w, cip).data_ptr(), back.data_ptr(), co, ci, cip, k, G.stream())
        assert torch.equal(back.cpu(), w)
        cop = (co + 3) // 4 * 4
        gotd = G.pack_dgrad(w, cop).cpu().reshape(ci, k * k, cop)
        refd = torch.zeros(ci, k * k, cop)
        refd[:, :, :co] = torch.flip(w.reshape(co, ci, k * k), dims=[2]).permute(1, 2, 0)
        assert torch.equal(gotd, refd)
        # the same operand from channels-last filters ([Cout][kh][kw][Cin] in memory, what Darknet keeps)
        wcl = w.to(G.dev()).contiguous(memory_format=torch.channels_last)
        if k == 1:
            wcl = w.to(G.dev()).permute(0, 2, 3, 1).contiguous()       # explicit [Cout][1][1][Cin] bytes
        outp = torch.full((ci * k * k * cop,), -1.0, device=G.dev())
        _lib.call('ssp_repack_dgrad_packed', wcl.data_ptr(), outp.data_ptr(), co, ci, cop, k, G.stream())
        assert torch.equal(outp.cpu().reshape(ci, k * k, cop), refd)


@pytest.mark.parametrize("pool", [0, 1])
@pytest.mark.parametrize("C,B,H,W", [(32, 2, 8, 12), (1024, 1, 4, 6), (20, 3, 6, 6), (64, 5, 26, 26)])
def test_bn_act_fwd_bwd(C, B, H, W, pool):
    G, _lib = _imports()
    rs = np.random.RandomState(C + pool)
    x = torch.from_numpy(rs.standard_normal((B, C, H, W)).astype(np.float32) * 2 + 0.3).requires_grad_(True)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)).requires_grad_(True)
    beta = torch.from_numpy((rs.standard_normal(C) * 0.3).astype(np.float32)).requires_grad_(True)
    y = F.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-4)
    y = F.leaky_relu(y, 0.1)
    if pool:
        y = F.max_pool2d(y, 2, 2)
    g = torch.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32))
    y.backward(g)
    xd = G.to_nhwc(x.detach())
    M = B * H * W
    x64 = x.detach().double()
    mean = x64.mean(dim=(0, 2, 3))
    var = x64.var(dim=(0, 2, 3), unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-4)
    vec = torch.zeros(8, C, dtype=torch.float32, device=G.dev())
    vec[0] = mean.float().to(G.dev())
    vec[1] = invstd.float().to(G.dev())
    vec[2] = (gamma.detach() * invstd.float()).to(G.dev())
    vec[3] = (beta.detach() - mean.float() * gamma.detach() * invstd.float()).to(G.dev())
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = torch.empty(B * Ho * Wo, C, dtype=torch.float32, device=G.dev())
    _lib.call('ssp_bn_act_fwd', xd.data_ptr(), C, out.data_ptr(), C, vec[2].data_ptr(), vec[3].data_ptr(), C, B, H, W, pool, 0.1, G.stream())
    torch.cuda.synchronize()
    assert rel_err(G.from_nhwc(out, B, C, Ho, Wo).numpy(), y.detach().numpy()) < TOL
    gd = G.to_nhwc(g)
    partial = torch.empty(_lib.query('ssp_bn_bwd_blocks') * C * 2, dtype=torch.float32, device=G.dev())
    dx = torch.empty_like(xd)
    _lib.call('ssp_bn_act_bwd', xd.data_ptr(), C, gd.data_ptr(), C, dx.data_ptr(), C, vec[2].data_ptr(), vec[3].data_ptr(),
              vec[0].data_ptr(), vec[1].data_ptr(), C, B, H, W, pool, 0.1, 1, partial.data_ptr(), vec[6].data_ptr(),
              vec[7].data_ptr(), vec[4].data_ptr(), vec[5].data_ptr(), G.stream())
    torch.cuda.synchronize()
    assert rel_err(G.from_nhwc(dx, B, C, H, W).numpy(), x.grad.numpy()) < 2e-4
    assert rel_err(vec[6].cpu().numpy(), gamma.grad.numpy()) < 2e-4
    assert rel_err(vec[7].cpu().numpy(), beta.grad.numpy()) < 2e-4
    # in-place form (dx aliases x), as the engine uses it
    x2 = xd.clone()
    _lib.call('ssp_bn_act_bwd', x2.data_ptr(), C, gd.data_ptr(), C, x2.data_ptr(), C, vec[2].data_ptr(), vec[3].data_ptr(),
              vec[0].data_ptr(), vec[1].data_ptr(), C, B, H, W, pool, 0.1, 1, partial.data_ptr(), vec[6].data_ptr(),
              vec[7].data_ptr(), vec[4].data_ptr(), vec[5].data_ptr(), G.stream())
    torch.cuda.synchronize()
    assert torch.equal(x2, dx)
    # single-pass form (partial = NULL): zeroed dgamma / dbeta accumulated with atomics, no finalize launch, in place
    x3 = xd.clone()
    dgam, dbet = torch.zeros(C, device=G.dev()), torch.zeros(C, device=G.dev())
    _lib.call('ssp_bn_act_bwd', x3.data_ptr(), C, gd.data_ptr(), C, x3.data_ptr(), C, vec[2].data_ptr(), vec[3].data_ptr(),
              vec[0].data_ptr(), vec[1].data_ptr(), C, B, H, W, pool, 0.1, 1, None, dgam.data_ptr(), dbet.data_ptr(),
              None, None, G.stream())
    torch.cuda.synchronize()
    assert rel_err(G.from_nhwc(x3, B, C, H, W).numpy(), x.grad.numpy()) < 2e-4
    assert rel_err(dgam.cpu().numpy(), gamma.grad.numpy()) < 2e-4
    assert rel_err(dbet.cpu().numpy(), beta.grad.numpy()) < 2e-4


def test_reorg_route_maxpool_bit_exact(golden_dir):
    G, _lib = _imports()
    from helpers import gold
    g = gold('reorg.npz')
    x = torch.from_numpy(g['x'])
    B, C, H, W = x.shape
    xd = G.to_nhwc(x, ld=C + 8, off=4)
    out = torch.full((B * (H // 2) * (W // 2), 4 * C + 12), float('nan'), dtype=torch.float32, device=G.dev())
    _lib.call('ssp_reorg', G.p(xd, 4), C + 8, G.p(out, 8), 4 * C + 12, C, B, H, W, 0, 0, G.stream())
    torch.cuda.synchronize()
    assert np.array_equal(G.from_nhwc(out, B, 4 * C, H // 2, W // 2, 8).numpy(), g['y'])   # golden from the reference's Reorg
    # backward = inverse permutation
    back = torch.zeros(B * H * W, C, dtype=torch.float32, device=G.dev())
    _lib.call('ssp_reorg', G.p(out, 8), 4 * C + 12, back.data_ptr(), C, C, B, H, W, 1, 0, G.stream())
    assert torch.equal(G.from_nhwc(back, B, C, H, W), x)
    _lib.call('ssp_reorg', G.p(out, 8), 4 * C + 12, back.data_ptr(), C, C, B, H, W, 1, 1, G.stream())
    assert torch.equal(G.from_nhwc(back, B, C, H, W), 2 * x)
    # route concat = two slice copies
    rs = np.random.RandomState(1)
    a = torch.from_numpy(rs.standard_normal((2, 8, 5, 5)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal((2, 12, 5, 5)).astype(np.float32))
    ad, bd = G.to_nhwc(a), G.to_nhwc(b)
    cat = torch.empty(2 * 25, 20, dtype=torch.float32, device=G.dev())
    _lib.call('ssp_copy_channels', ad.data_ptr(), 8, cat.data_ptr(), 20, 8, 50, 0, G.stream())
    _lib.call('ssp_copy_channels', bd.data_ptr(), 12, G.p(cat, 8), 20, 12, 50, 0, G.stream())
    assert torch.equal(G.from_nhwc(cat, 2, 20, 5, 5), torch.cat((a, b), 1))
    # layout round trip NCHW <-> NHWC
    xn = torch.from_numpy(rs.standard_normal((2, 3, 6, 8)).astype(np.float32)).to(G.dev())
    nh = torch.empty(2 * 48 * 4, dtype=torch.float32, device=G.dev())
    _lib.call('ssp_nchw_to_nhwc', xn.data_ptr(), nh.data_ptr(), 2, 3, 6, 8, 4, 4, G.stream())
    ref = torch.zeros(2, 6, 8, 4)
    ref[..., :3] = xn.cpu().permute(0, 2, 3, 1)
    assert torch.equal(nh.cpu().view(2, 6, 8, 4), ref)
    xb = torch.empty(2, 3, 6, 8, dtype=torch.float32, device=G.dev())
    _lib.call('ssp_nhwc_to_nchw', nh.data_ptr(), xb.data_ptr(), 2, 3, 6, 8, 4, G.stream())
    assert torch.equal(xb, xn)
    # max-pool fwd/bwd incl. ties (first maximum wins)
    xm = torch.from_numpy(rs.randint(0, 3, (2, 8, 6, 6)).astype(np.float32)).requires_grad_(True)
    ym = F.max_pool2d(xm, 2, 2)
    gm = torch.from_numpy(rs.standard_normal(tuple(ym.shape)).astype(np.float32))
    ym.backward(gm)
    xmd, gmd = G.to_nhwc(xm.detach()), G.to_nhwc(gm)
    ymd = torch.empty(2 * 9, 8, dtype=torch.float32, device=G.dev())
    _lib.call('ssp_maxpool_fwd', xmd.data_ptr(), 8, ymd.data_ptr(), 8, 8, 2, 6, 6, G.stream())
    assert torch.equal(G.from_nhwc(ymd, 2, 8, 3, 3), ym.detach())
    dxm = torch.empty(2 * 36, 8, dtype=torch.float32, device=G.dev())
    _lib.call('ssp_maxpool_bwd', xmd.data_ptr(), 8, gmd.data_ptr(), 8, dxm.data_ptr(), 8, 8, 2, 6, 6, 0, G.stream())
    assert torch.equal(G.from_nhwc(dxm, 2, 8, 6, 6), xm.grad)


def test_colsum():
    G, _lib = _imports()
    rs = np.random.RandomState(5)
    g = torch.from_numpy(rs.standard_normal((1000, 20)).astype(np.float32))
    gd = g.to(G.dev())
    out = torch.empty(20, dtype=torch.float32, device=G.dev())
    _lib.call('ssp_colsum', gd.data_ptr(), 20, 1000, 20, out.data_ptr(), G.stream())
    np.testing.assert_allclose(out.cpu().numpy(), g.double().sum(0).numpy(), rtol=1e-5, atol=1e-5)


def test_error_reporting():
    G, _lib = _imports()
    with pytest.raises(_lib.SspError, match="1x1 and 3x3"):
        _lib.call('ssp_conv_fwd', None, None, None, None, None, 1, 4, 4, 4, 4, 4, 4, 5, 0, 0, None, 0, G.stream())


@pytest.mark.parametrize("B,H,W,C,Cp,ld", [(2, 32, 32, 3, 4, 4), (1, 5, 7, 3, 4, 4), (1, 3, 3, 1, 4, 8), (2, 4, 6, 4, 4, 4),
                                          (1, 1, 1, 3, 4, 4)])
def test_u8_image_to_nhwc_bit_exact(B, H, W, C, Cp, ld):
    """uint8 HWC bytes -> fp32 NHWC/255: bit-exact against ToTensor's arithmetic (uint8 -> float32, true division by
    255; dataset.py:113-131 via torchvision.transforms.ToTensor), every byte value covered."""
    from singleshotpose_amd import _lib
    rs = np.random.RandomState(B * 1000 + H)
    img = rs.randint(0, 256, (B, H, W, C)).astype(np.uint8)
    img.reshape(-1)[:min(256, img.size)] = np.arange(min(256, img.size), dtype=np.uint8)
    src = torch.from_numpy(img).cuda()
    dst = torch.full((B * H * W, ld), -7.0, device='cuda')
    _lib.call('ssp_u8hwc_to_nhwc', src.data_ptr(), dst.data_ptr(), B, H, W, C, Cp, ld,
              torch.cuda.current_stream().cuda_stream)
    want = torch.from_numpy(img).to(torch.float32).div(255).numpy().reshape(-1, C)       # ToTensor: .float().div(255)
    got = dst.cpu().numpy()
    assert np.array_equal(got[:, :C], want)
    assert np.all(got[:, C:Cp] == 0.0) and np.all(got[:, Cp:] == -7.0)


@pytest.mark.parametrize("B,H,W,Cin,Cout,R,plan,slope,with_scale", [
    (2, 13, 13, 64, 128, 3, 0, 0.1, True),        # LDS-direct 128x128 tile, epilogue path
    (3, 10, 14, 128, 256, 3, 12834, 0.1, True),   # forced split-K x3: affine + leaky applied by the partial-sum pass
    (2, 9, 9, 64, 20, 1, 0, 1.0, False),          # linear head: shift only (bias), 256x32 tile
    (1, 12, 12, 3, 32, 3, 0, 0.1, True),          # first layer (4-channel register-staged kernel)
])
def test_conv_fwd_affine_eval_block(B, H, W, Cin, Cout, R, plan, slope, with_scale):
    """ssp_conv_fwd_affine = the whole inference-mode block (darknet.py:154-167): leaky(scale * conv + shift)."""
    G, _lib = _imports()
    rs = np.random.RandomState(Cin + Cout + R)
    x = torch.from_numpy(rs.standard_normal((B, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rs.standard_normal((Cout, Cin, R, R)) / np.sqrt(Cin * R * R)).astype(np.float32))
    scale = torch.from_numpy(rs.uniform(0.5, 1.5, Cout).astype(np.float32))
    shift = torch.from_numpy(rs.standard_normal(Cout).astype(np.float32))
    y = F.conv2d(x, w, None, padding=R // 2)
    if with_scale:
        y = y * scale.view(1, -1, 1, 1)
    ref = F.leaky_relu(y + shift.view(1, -1, 1, 1), slope) if slope != 1.0 else y + shift.view(1, -1, 1, 1)
    cinp = (Cin + 3) // 4 * 4
    xp = torch.zeros(B, cinp, H, W)
    xp[:, :Cin] = x
    xd, wd = G.to_nhwc(xp), G.pack_fwd(w, cinp)
    out = torch.full((B * H * W, Cout), float('nan'), dtype=torch.float32, device=G.dev())
    wsn = max(1, _lib.query('ssp_conv_workspace_floats', B, H, W, cinp, Cout, R, plan))
    ws = torch.empty(wsn, dtype=torch.float32, device=G.dev())
    sd, hd = scale.to(G.dev()), shift.to(G.dev())
    _lib.call('ssp_conv_fwd_affine', xd.data_ptr(), wd.data_ptr(), out.data_ptr(), sd.data_ptr() if with_scale else None,
              hd.data_ptr(), slope, B, H, W, cinp, Cout, cinp, Cout, R, plan, ws.data_ptr(), wsn, G.stream())
    torch.cuda.synchronize()
    assert rel_err(G.from_nhwc(out, B, Cout, H, W).numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("plan", [306413, 212814, 306414])
def test_conv_hybrid_launch_matches_plain(plan):
    """Hybrid launch (whole resident waves of un-split tiles + the tail tiles split over K, one launch + a partial-sum
    pass over the tail rows): output, accumulate mode and BatchNorm statistics against the plain launch of the same
    tiling.  16 x 52 x 52 pixels, 128 -> 256 channels: 1352 (64-row) / 676 (128-row) tiles, more than one resident wave."""
    G, _lib = _imports()
    B, H, W, Cin, Cout, R = 16, 52, 52, 128, 256, 3
    rs = np.random.RandomState(plan % 1000)
    x = torch.from_numpy(rs.standard_normal((B, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rs.standard_normal((Cout, Cin, R, R)) / np.sqrt(Cin * R * R)).astype(np.float32))
    xd, wd = G.to_nhwc(x), G.pack_fwd(w, Cin)
    M = B * H * W

    def run(code, accumulate_twice=False):
        tile_m = _lib.query('ssp_conv_stats_tile_m', B, H, W, Cin, Cout, R, code)
        wsn = max(1, _lib.query('ssp_conv_workspace_floats', B, H, W, Cin, Cout, R, code))
        ws = torch.empty(wsn, dtype=torch.float32, device=G.dev())
        ntile = (M + tile_m - 1) // tile_m
        stats = torch.full((ntile * Cout * 2,), float('nan'), dtype=torch.float32, device=G.dev())
        out = torch.zeros(M, Cout, dtype=torch.float32, device=G.dev())
        _lib.call('ssp_conv_fwd', xd.data_ptr(), wd.data_ptr(), out.data_ptr(), None, stats.data_ptr(), B, H, W, Cin,
                  Cout, Cin, Cout, R, 0, code, ws.data_ptr(), wsn, G.stream())
        if accumulate_twice:
            _lib.call('ssp_conv_fwd', xd.data_ptr(), wd.data_ptr(), out.data_ptr(), None, None, B, H, W, Cin, Cout,
                      Cin, Cout, R, 1, code, ws.data_ptr(), wsn, G.stream())
        torch.cuda.synchronize()
        return out.cpu().numpy(), stats.cpu().numpy().reshape(ntile, Cout, 2), tile_m, wsn
    base = plan % 100000
    o_plain, s_plain, tm_plain, ws_plain = run(base)
    o_hyb, s_hyb, tm_hyb, ws_hyb = run(plan)
    assert tm_plain == tm_hyb and ws_hyb >= (plan // 100000) * M * Cout > ws_plain
    assert rel_err(o_hyb, o_plain) < 1e-5
    assert not np.isnan(s_hyb).any()
    np.testing.assert_allclose(s_hyb[:, :, 0], s_plain[:, :, 0], rtol=1e-4, atol=1e-5)      # per-tile means
    np.testing.assert_allclose(s_hyb[:, :, 1], s_plain[:, :, 1], rtol=1e-3)                 # per-tile M2
    o2, _, _, _ = run(plan, accumulate_twice=True)
    assert rel_err(o2, 2 * o_plain) < 1e-5
    # the plain-PyTorch check of the same convolution on a slice of the batch (covers main and tail rows: last images)
    ref = F.conv2d(x[-2:], w, None, padding=1)
    got = G.from_nhwc(torch.from_numpy(o_hyb[-2 * H * W:]), 2, Cout, H, W)
    assert rel_err(got.numpy(), ref.numpy()) < TOL


def test_conv_plan_is_an_argument_two_threads_two_plans():
    """the boundary is re-entrant - the tile / split plan travels with the call, so two host
    threads running different plans on different streams (two models, two shapes) get bit-identical results to the
    same launches issued serially.  (Round 1 set a process-global knob before every launch.)"""
    import threading
    G, _lib = _imports()
    B, H, W, Cin, Cout, R = 8, 13, 13, 512, 1024, 3
    M = B * H * W
    g = torch.Generator(device='cuda').manual_seed(11)
    x = torch.empty(M * Cin, device=G.dev()).uniform_(-1, 1, generator=g)
    w = torch.empty(Cout * R * R * Cin, device=G.dev()).uniform_(-0.05, 0.05, generator=g)
    plans = (12834, 6464)           # split-K x3 on 128-row tiles / split-K x6 on 64-row tiles

    def run(code, stream, out, ws, wsn, reps):
        for _ in range(reps):
            _lib.call('ssp_conv_fwd', x.data_ptr(), w.data_ptr(), out.data_ptr(), None, None, B, H, W, Cin, Cout, Cin,
                      Cout, R, 0, code, ws.data_ptr(), wsn, stream.cuda_stream)

    def buffers(code):
        wsn = max(1, _lib.query('ssp_conv_workspace_floats', B, H, W, Cin, Cout, R, code))
        assert wsn >= (code // 10 % 10) * M * Cout
        return torch.zeros(M * Cout, device=G.dev()), torch.empty(wsn, device=G.dev()), wsn

    serial = []
    for code in plans:
        out, ws, wsn = buffers(code)
        run(code, torch.cuda.current_stream(), out, ws, wsn, 1)
        torch.cuda.synchronize()
        serial.append(out.clone())
    assert not torch.equal(serial[0], serial[1])            # different summation orders: the plans really differ
    assert rel_err(serial[1].cpu().numpy(), serial[0].cpu().numpy()) < 1e-5
    streams = [torch.cuda.Stream() for _ in plans]
    bufs = [buffers(code) for code in plans]
    torch.cuda.synchronize()
    threads = [threading.Thread(target=run, args=(code, s, b[0], b[1], b[2], 200)) for code, s, b in zip(plans, streams, bufs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    for got, want in zip(bufs, serial):
        assert torch.equal(got[0], want)


@pytest.mark.parametrize("B,H,W,Cdy,Cdx,R,plan", [
    (4, 13, 13, 64, 128, 3, 0),            # direct epilogue, 128 x 128 / 64 x 128 tiles
    (4, 13, 13, 128, 64, 1, 0),            # 1 x 1, 128 x 64 tiles
    (16, 13, 13, 512, 256, 3, 12834),      # split-K x3: the sums come out of splitk_reduce_kernel
    (16, 52, 52, 128, 256, 3, 306413),     # hybrid launch: un-split tiles (epilogue) + tail tiles (reduce kernel)
    (2, 13, 13, 20, 1024, 1, 0),           # the head's data gradient (20 channels: register-staged kernel)
])
def test_conv_dgrad_with_fused_bn_backward_reductions(B, H, W, Cdy, Cdx, R, plan):
    """ssp_conv_dgrad_bnbwd = ssp_conv_dgrad + the (sum dy, sum dy * xhat) reductions of the producing block, and
    ssp_bn_act_bwd_partials finishes that block: gradient, dgamma / dbeta and dx against torch autograd (fp64) of
    leaky(batch_norm(raw)) fed the same upstream gradient, and against the two-pass ssp_bn_act_bwd."""
    G, _lib = _imports()
    rs = np.random.RandomState(Cdy + Cdx + R)
    M = B * H * W
    dy = torch.from_numpy(rs.standard_normal((B, Cdy, H, W)).astype(np.float32))
    wt = torch.from_numpy((rs.standard_normal((Cdy, Cdx, R, R)) / np.sqrt(Cdy * R * R)).astype(np.float32))
    raw = torch.from_numpy((rs.standard_normal((B, Cdx, H, W)) * 1.5 + 0.3).astype(np.float32))   # the producer's conv output
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, Cdx).astype(np.float32))
    beta = torch.from_numpy((rs.standard_normal(Cdx) * 0.2).astype(np.float32))
    # reference: g = conv_transpose-style data gradient of y = conv(a, wt); then BN + leaky backward in double
    a = torch.zeros(B, Cdx, H, W, requires_grad=True)
    F.conv2d(a, wt, None, padding=R // 2).backward(dy)
    g_ref = a.grad.double()
    rawd = raw.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    out = F.leaky_relu(F.batch_norm(rawd, None, None, gd, bd, True, 0.1, 1e-4), 0.1)
    out.backward(g_ref)
    coutp = (Cdy + 3) // 4 * 4
    dyp = torch.zeros(B, coutp, H, W)
    dyp[:, :Cdy] = dy
    dyd = G.to_nhwc(dyp)
    wd = G.pack_dgrad(wt, coutp)
    rawd_dev = G.to_nhwc(raw)
    mean = raw.double().mean(dim=(0, 2, 3))
    var = raw.double().var(dim=(0, 2, 3), unbiased=False)
    istd = 1.0 / torch.sqrt(var + 1e-4)
    vec = torch.stack([mean, istd, gamma.double() * istd, beta.double() - mean * gamma.double() * istd]).float().to(G.dev())
    tm = _lib.query('ssp_conv_stats_tile_m', B, H, W, coutp, Cdx, R, plan)
    ntile = (M + tm - 1) // tm
    wsn = max(1, _lib.query('ssp_conv_workspace_floats', B, H, W, coutp, Cdx, R, plan))
    ws = torch.empty(wsn, device=G.dev())
    rows = min(ntile, 16)                       # 16 rows: the bigger cases fold their tiles with atomics
    partial = torch.zeros(rows * Cdx * 2, device=G.dev())
    gx = torch.full((M, Cdx), float('nan'), device=G.dev())
    _lib.call('ssp_conv_dgrad_bnbwd', dyd.data_ptr(), wd.data_ptr(), gx.data_ptr(), B, H, W, coutp, Cdx, coutp, Cdx, R,
              plan, ws.data_ptr(), wsn, rawd_dev.data_ptr(), Cdx, vec[2].data_ptr(), vec[3].data_ptr(),
              vec[0].data_ptr(), vec[1].data_ptr(), 0.1, partial.data_ptr(), rows, G.stream())
    torch.cuda.synchronize()
    assert rel_err(G.from_nhwc(gx, B, Cdx, H, W).numpy(), g_ref.numpy()) < TOL
    ps = partial.cpu().double().view(rows, Cdx, 2).sum(0)
    assert not torch.isnan(ps).any()
    assert rel_err(ps[:, 0].numpy(), bd.grad.numpy()) < TOL                 # sum dy = dbeta
    assert rel_err(ps[:, 1].numpy(), gd.grad.numpy()) < TOL                 # sum dy * xhat = dgamma
    # finish the block from the partials, in place over the raw output (as Plan.backward does)
    out_vec = torch.zeros(4, Cdx, device=G.dev())
    dx1 = rawd_dev.clone()
    _lib.call('ssp_bn_act_bwd_partials', dx1.data_ptr(), Cdx, gx.data_ptr(), Cdx, dx1.data_ptr(), Cdx, vec[2].data_ptr(),
              vec[3].data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), Cdx, B, H, W, 0.1, 1, partial.data_ptr(), rows, 1,
              out_vec[0].data_ptr(), out_vec[1].data_ptr(), out_vec[2].data_ptr(), out_vec[3].data_ptr(), G.stream())
    torch.cuda.synchronize()
    assert float(partial.abs().max()) == 0.0            # zero_after: ready for the next (atomically folded) launch
    assert rel_err(G.from_nhwc(dx1, B, Cdx, H, W).numpy(), rawd.grad.numpy()) < TOL
    assert rel_err(out_vec[0].cpu().numpy(), gd.grad.numpy()) < TOL and rel_err(out_vec[1].cpu().numpy(), bd.grad.numpy()) < TOL
    # and the two-pass form on the same inputs
    nblk = _lib.query('ssp_bn_bwd_blocks')
    p2 = torch.empty(nblk * Cdx * 2, device=G.dev())
    out2 = torch.zeros(4, Cdx, device=G.dev())
    dx2 = rawd_dev.clone()
    _lib.call('ssp_bn_act_bwd', dx2.data_ptr(), Cdx, gx.data_ptr(), Cdx, dx2.data_ptr(), Cdx, vec[2].data_ptr(),
              vec[3].data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), Cdx, B, H, W, 0, 0.1, 1, p2.data_ptr(),
              out2[0].data_ptr(), out2[1].data_ptr(), out2[2].data_ptr(), out2[3].data_ptr(), G.stream())
    torch.cuda.synchronize()
    assert rel_err(dx1.cpu().numpy(), dx2.cpu().numpy()) < 1e-5
    with pytest.raises(_lib.SspError):       # no silent fall-through: the fused form needs its output buffer
        _lib.call('ssp_conv_dgrad_bnbwd', dyd.data_ptr(), wd.data_ptr(), gx.data_ptr(), B, H, W, coutp, Cdx, coutp, Cdx,
                  R, plan, ws.data_ptr(), wsn, rawd_dev.data_ptr(), Cdx, vec[2].data_ptr(), vec[3].data_ptr(),
                  vec[0].data_ptr(), vec[1].data_ptr(), 0.1, None, rows, G.stream())
