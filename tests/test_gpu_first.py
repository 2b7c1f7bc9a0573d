"""The fused first block (csrc/conv_first.hip): conv 3x3 (3 -> 32) + BatchNorm(train) + leaky + 2x2 max-pool with the
convolution recomputed by every pass - against torch (float64 autograd of the same chain, darknet.py:154-176)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("B,H,W", [(2, 8, 32), (1, 12, 64), (3, 16, 96), (2, 64, 128)])
def test_first_block_forward_and_backward(B, H, W):
    import gpu_util as G
    from singleshotpose_amd import _lib
    rs = np.random.RandomState(B * 1000 + H + W)
    x = torch.from_numpy(rs.uniform(0, 1, (B, 3, H, W)).astype(np.float32))
    w = torch.from_numpy((rs.standard_normal((32, 3, 3, 3)) * 0.4).astype(np.float32))
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, 32).astype(np.float32))
    beta = torch.from_numpy((rs.standard_normal(32) * 0.2).astype(np.float32))
    gpool = torch.from_numpy(rs.standard_normal((B, 32, H // 2, W // 2)).astype(np.float32))
    # float64 reference of the whole chain
    xd = x.double()
    wd, gd, bd = w.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    raw = F.conv2d(xd, wd, None, padding=1)
    out = F.max_pool2d(F.leaky_relu(F.batch_norm(raw, None, None, gd, bd, True, 0.1, 1e-4), 0.1), 2, 2)
    out.backward(gpool.double())
    mean_ref = raw.detach().mean(dim=(0, 2, 3))
    var_ref = raw.detach().var(dim=(0, 2, 3), unbiased=False)

    st = G.stream()
    M = B * H * W
    xp = torch.zeros(B, 4, H, W)
    xp[:, :3] = x
    xdev = G.to_nhwc(xp)
    wdev = G.pack_fwd(w, 4)
    groups = _lib.query('ssp_first_groups', B, H, W)
    tile = _lib.query('ssp_first_tile_pixels')
    assert groups == -(-(B * (H // 2) * (W // 16)) // 64) and tile == 2048
    stats = torch.full((groups * 64,), float('nan'), device=G.dev())
    _lib.call('ssp_first_fwd_stats', xdev.data_ptr(), wdev.data_ptr(), stats.data_ptr(), B, H, W, st)
    vec = torch.zeros(8, 32, device=G.dev())
    gdev, bdev = gamma.to(G.dev()), beta.to(G.dev())
    rmean, rvar = torch.zeros(32, device=G.dev()), torch.ones(32, device=G.dev())
    _lib.call('ssp_bn_fwd_finalize', stats.data_ptr(), groups, tile, M, 32, gdev.data_ptr(), bdev.data_ptr(),
              rmean.data_ptr(), rvar.data_ptr(), 0.1, 1e-4, vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(),
              vec[3].data_ptr(), st)
    torch.cuda.synchronize()
    assert rel_err(vec[0].cpu().numpy(), mean_ref.numpy()) < 1e-5
    assert rel_err(vec[1].cpu().numpy(), (1.0 / torch.sqrt(var_ref + 1e-4)).numpy()) < 1e-5
    np.testing.assert_allclose(rmean.cpu().numpy(), 0.1 * mean_ref.numpy(), rtol=1e-4, atol=1e-6)
    # the raw map (checker entry point) against the float64 convolution
    rawdev = torch.full((M, 36), float('nan'), device=G.dev())
    _lib.call('ssp_first_conv_raw', xdev.data_ptr(), wdev.data_ptr(), rawdev.data_ptr(), 36, B, H, W, st)
    torch.cuda.synchronize()
    assert rel_err(G.from_nhwc(rawdev, B, 32, H, W).numpy(), raw.detach().numpy()) < 1e-5
    assert torch.isnan(rawdev.cpu()[:, 32:]).all()
    # forward apply: pooled activation into a wider (channel-sliced) buffer
    ldo = 40
    P = B * (H // 2) * (W // 2)
    odev = torch.full((P, ldo), float('nan'), device=G.dev())
    _lib.call('ssp_first_fwd_apply', xdev.data_ptr(), wdev.data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), 0.1,
              odev.data_ptr(), ldo, B, H, W, st)
    torch.cuda.synchronize()
    got = G.from_nhwc(odev, B, 32, H // 2, W // 2)
    assert rel_err(got.numpy(), out.detach().numpy()) < TOL
    assert torch.isnan(odev.cpu()[:, 32:]).all()
    # backward: reductions -> finalize -> filter gradient
    gdev_p = G.to_nhwc(gpool, ldo)
    partial = torch.full((groups * 64,), float('nan'), device=G.dev())
    _lib.call('ssp_first_bwd_reduce', xdev.data_ptr(), wdev.data_ptr(), gdev_p.data_ptr(), ldo, vec[2].data_ptr(),
              vec[3].data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), 0.1, partial.data_ptr(), B, H, W, st)
    _lib.call('ssp_bn_bwd_finalize', partial.data_ptr(), groups, 32, M, 1, 0, vec[6].data_ptr(), vec[7].data_ptr(),
              vec[4].data_ptr(), vec[5].data_ptr(), st)
    dw = torch.full((32 * 36,), float('nan'), device=G.dev())      # written, not accumulated (ABI 4): NaN-poisoned
    wsn = _lib.query('ssp_first_wgrad_workspace_floats', B, H, W)
    wsp = torch.full((wsn,), float('nan'), device=G.dev())
    _lib.call('ssp_first_bwd_wgrad', xdev.data_ptr(), wdev.data_ptr(), gdev_p.data_ptr(), ldo, vec[2].data_ptr(),
              vec[3].data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[4].data_ptr(), vec[5].data_ptr(), 0.1,
              dw.data_ptr(), wsp.data_ptr(), wsn, B, H, W, st)
    with pytest.raises(_lib.SspError, match="workspace"):
        _lib.call('ssp_first_bwd_wgrad', xdev.data_ptr(), wdev.data_ptr(), gdev_p.data_ptr(), ldo, vec[2].data_ptr(),
                  vec[3].data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[4].data_ptr(), vec[5].data_ptr(), 0.1,
                  dw.data_ptr(), wsp.data_ptr(), wsn - 1, B, H, W, st)
    torch.cuda.synchronize()
    assert rel_err(vec[6].cpu().numpy(), gd.grad.numpy()) < TOL            # dgamma
    assert rel_err(vec[7].cpu().numpy(), bd.grad.numpy()) < TOL            # dbeta
    dwp = dw.cpu().view(32, 9, 4)
    assert float(dwp[:, :, 3].abs().max()) == 0.0                          # the padding channel stays zero
    got_dw = dwp[:, :, :3].permute(0, 2, 1).reshape(32, 3, 3, 3)
    assert rel_err(got_dw.numpy(), wd.grad.numpy()) < 2e-4


def test_first_block_rejects_unsupported_shapes():
    import gpu_util as G
    from singleshotpose_amd import _lib
    x = torch.zeros(1 * 6 * 24 * 4, device=G.dev())
    w = torch.zeros(32 * 36, device=G.dev())
    s = torch.zeros(64, device=G.dev())
    with pytest.raises(_lib.SspError, match="multiple of 16"):
        _lib.call('ssp_first_fwd_stats', x.data_ptr(), w.data_ptr(), s.data_ptr(), 1, 6, 24, G.stream())
