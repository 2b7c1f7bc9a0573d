// Index-only kernels: tensor layout changes, filter repacks, reorg (space-to-depth), route copies and the
// standalone 2x2/2 max-pool.  All of them are HBM-bound copies; reorg / route / repacks are bit-exact.
//
// Reference semantics:
//   Reorg        /root/reference/darknet.py:16-35   out[b,(dy*2+dx)*C+c,hy,wx] = in[b,c,2hy+dy,2wx+dx]
//   route        /root/reference/darknet.py:96-106  alias or torch.cat((x1,x2),1)
//   MaxPool2d    /root/reference/darknet.py:168-172 (2x2 stride 2; first maximum wins in backward)
//   weights      /root/reference/cfg.py:153-176     conv.weight is (Cout,Cin,kh,kw) row-major
#include "ssp_common.h"

static int elem_grid(int64_t total) {
  int64_t blocks = (total + 255) / 256;
  const int64_t cap = 256 * 16;
  return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

// (B,C,H,W) -> [B*H*W][ld] with channels c < C copied (others untouched unless zero_pad: then c in [C,Cp) = 0)
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                           int Cp, int64_t HW, int64_t total_pix, int ld) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total_pix;
       p += (int64_t)gridDim.x * blockDim.x) {
    int64_t b = p / HW, hw = p % HW;
    const float* s = src + b * C * HW + hw;
    float* d = dst + p * ld;
    if (Cp == 4 && C <= 4) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < C; ++c) v[c] = s[c * HW];
      *reinterpret_cast<f32x4*>(d) = v;
    } else {
      for (int c = 0; c < C; ++c) d[c] = s[c * HW];
      for (int c = C; c < Cp; ++c) d[c] = 0.f;
    }
  }
}

// uint8 (B,H,W,C) image bytes -> fp32 [B*H*W][ld], value/255 exactly as torchvision's ToTensor (dataset.py:113-131 via
// transforms.ToTensor: `img.float().div(255)`), channels [C,Cp) zeroed.  Replaces the host-side ToTensor + NCHW float
// upload (133 MB per 64x416x416 batch) by a 33 MB byte upload and this pass.
__global__ void __launch_bounds__(256) u8hwc_to_nhwc_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst,
                                                            int C, int Cp, int64_t total_pix, int ld, int fast) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (fast) {
    // C == 3, Cp == 4, ld == 4, everything 16-byte aligned: 4 pixels = 3 dwords in, 4 float4 out per thread
    const int64_t quads = total_pix >> 2;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += stride) {
      const unsigned int* s = reinterpret_cast<const unsigned int*>(src) + q * 3;
      const unsigned int w0 = s[0], w1 = s[1], w2 = s[2];
      unsigned char b[12];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        b[i] = (w0 >> (8 * i)) & 0xff;
        b[4 + i] = (w1 >> (8 * i)) & 0xff;
        b[8 + i] = (w2 >> (8 * i)) & 0xff;
      }
      f32x4* d = reinterpret_cast<f32x4*>(dst) + q * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = {(float)b[3 * i] / 255.0f, (float)b[3 * i + 1] / 255.0f, (float)b[3 * i + 2] / 255.0f, 0.f};
        d[i] = v;
      }
    }
    for (int64_t p = (quads << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total_pix; p += stride) {
      f32x4 v = {(float)src[p * 3] / 255.0f, (float)src[p * 3 + 1] / 255.0f, (float)src[p * 3 + 2] / 255.0f, 0.f};
      reinterpret_cast<f32x4*>(dst)[p] = v;
    }
    return;
  }
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total_pix; p += stride) {
    const unsigned char* s = src + p * C;
    float* d = dst + p * ld;
    for (int c = 0; c < C; ++c) d[c] = (float)s[c] / 255.0f;
    for (int c = C; c < Cp; ++c) d[c] = 0.f;
  }
}

// [B*H*W][ld] (first C channels) -> (B,C,H,W)
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                           int64_t HW, int64_t total, int ld) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t hw = i % HW;
    int64_t t = i / HW;
    int c = (int)(t % C);
    int64_t b = t / C;
    dst[i] = src[(b * HW + hw) * ld + c];
  }
}

// OIHW (Cout,Cin,R,R) -> forward-conv operand [Cout][R*R][Cinp] (k contiguous; channels >= Cin are zero).
// One workgroup per (cout, 64-channel chunk): contiguous read of 64*taps floats, LDS transpose, contiguous writes.
__global__ void __launch_bounds__(256) repack_fwd_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout,
                                                         int Cin, int Cinp, int taps) {
  __shared__ float tile[64 * 9];
  const int co = blockIdx.y, c0 = blockIdx.x * 64;
  const int cc = min(64, Cin - c0);    // real channels in this chunk
  const int ccp = min(64, Cinp - c0);  // incl. zero padding
  const float* src = w + ((int64_t)co * Cin + c0) * taps;
  for (int i = threadIdx.x; i < cc * taps; i += 256) tile[i] = src[i];
  __syncthreads();
  for (int o = threadIdx.x; o < ccp * taps; o += 256) {
    int tap = o / ccp, ci = o % ccp;
    out[((int64_t)co * taps + tap) * Cinp + c0 + ci] = ci < cc ? tile[ci * taps + tap] : 0.f;
  }
}

// inverse of repack_fwd for gradients: [Cout][R*R][Cinp] -> OIHW (Cout,Cin,R,R)
__global__ void __launch_bounds__(256) unpack_grad_kernel(const float* __restrict__ dwp, float* __restrict__ grad,
                                                          int Cout, int Cin, int Cinp, int taps) {
  __shared__ float tile[64 * 9];
  const int co = blockIdx.y, c0 = blockIdx.x * 64;
  const int cc = min(64, Cin - c0);
  for (int o = threadIdx.x; o < cc * taps; o += 256) {
    int tap = o / cc, ci = o % cc;
    tile[ci * taps + tap] = dwp[((int64_t)co * taps + tap) * Cinp + c0 + ci];
  }
  __syncthreads();
  float* dst = grad + ((int64_t)co * Cin + c0) * taps;
  for (int i = threadIdx.x; i < cc * taps; i += 256) dst[i] = tile[i];
}

// OIHW -> data-gradient operand [Cin][R*R][Coutp]:  out[ci][t][co] = w[co][ci][taps-1-t]  (180-degree flip,
// in/out transposed); columns co >= Cout are zero.  Tile: 32 cout x 32 cin x taps through LDS.
__global__ void __launch_bounds__(256) repack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                           int Cout, int Cin, int Coutp, int taps) {
  __shared__ float tile[32][32 * 9 + 1];
  const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
  const int nci = min(32, Cin - ci0);
  const int row_len = nci * taps;
  for (int r = 0; r < 32; ++r) {
    int co = co0 + r;
    if (co < Cout) {
      const float* src = w + ((int64_t)co * Cin + ci0) * taps;
      for (int i = threadIdx.x; i < row_len; i += 256) tile[r][i] = src[i];
    }
  }
  __syncthreads();
  const int nco = min(32, Coutp - co0);
  // o enumerates (ci, tap, co) with co fastest -> 128-byte contiguous runs
  for (int o = threadIdx.x; o < nci * taps * nco; o += 256) {
    int co = o % nco;
    int t = (o / nco) % taps;
    int ci = o / (nco * taps);
    float v = (co0 + co < Cout) ? tile[co][ci * taps + (taps - 1 - t)] : 0.f;
    out[((int64_t)(ci0 + ci) * taps + t) * Coutp + co0 + co] = v;
  }
}

// Same operand from filters already stored channels-last, [Cout][R*R][Cin] (the forward operand layout, which is also
// what a torch.channels_last conv.weight is in memory): out[ci][t][co] = wp[co][taps-1-t][ci] - one plain 32x32 LDS
// transpose per filter tap, 128-byte runs on both sides.
__global__ void __launch_bounds__(256) repack_dgrad_packed_kernel(const float* __restrict__ wp, float* __restrict__ out,
                                                                  int Cout, int Cin, int Coutp, int taps) {
  __shared__ float tile[32][33];
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32, t = blockIdx.z;
  const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
  for (int r = r0; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + c;
    tile[r][c] = (co < Cout && ci < Cin) ? wp[((int64_t)co * taps + t) * Cin + ci] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = r0; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + c;
    if (ci < Cin && co < Coutp) out[((int64_t)ci * taps + (taps - 1 - t)) * Coutp + co] = tile[c][r];
  }
}

// space-to-depth, stride 2, NHWC: out[b,hy,wx,(dy*2+dx)*C + c] = in[b,2hy+dy,2wx+dx,c]
// dir = 0: forward gather (in -> out); dir = 1: backward scatter (gradient: out-layout -> in-layout)
__global__ void __launch_bounds__(256) reorg_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst,
                                                    int ldd, int C, int B, int H, int W, int dir, int accumulate) {
  const int G = C >> 2;
  const int Ho = H >> 1, Wo = W >> 1;
  const int64_t total = (int64_t)B * H * W * G;  // one item per 4 input-layout channels
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int g = (int)(idx % G);
    int64_t p = idx / G;
    int x = (int)(p % W);
    int64_t t = p / W;
    int y = (int)(t % H);
    int64_t b = t / H;
    int64_t in_off = p;  // pixel index in the fine (H x W) map
    int64_t out_pix = (b * Ho + (y >> 1)) * Wo + (x >> 1);
    int oc = ((y & 1) * 2 + (x & 1)) * C + g * 4;
    if (dir == 0) {
      f32x4 v = *reinterpret_cast<const f32x4*>(src + in_off * lds_ + g * 4);
      *reinterpret_cast<f32x4*>(dst + out_pix * ldd + oc) = v;
    } else {
      f32x4 v = *reinterpret_cast<const f32x4*>(src + out_pix * lds_ + oc);
      float* d = dst + in_off * ldd + g * 4;
      if (accumulate) {
        f32x4 o = *reinterpret_cast<const f32x4*>(d);
        v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
      }
      *reinterpret_cast<f32x4*>(d) = v;
    }
  }
}

// strided channel-slice copy / accumulate: dst[m][0..C) (+)= src[m][0..C)
__global__ void __launch_bounds__(256) copy_channels_kernel(const float* __restrict__ src, int lds_,
                                                            float* __restrict__ dst, int ldd, int C, int64_t M,
                                                            int accumulate) {
  const int G = C >> 2;
  const int64_t total = M * G;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int g = (int)(idx % G);
    int64_t m = idx / G;
    f32x4 v = *reinterpret_cast<const f32x4*>(src + m * lds_ + g * 4);
    float* d = dst + m * ldd + g * 4;
    if (accumulate) {
      f32x4 o = *reinterpret_cast<const f32x4*>(d);
      v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
    }
    *reinterpret_cast<f32x4*>(d) = v;
  }
}

// standalone 2x2/2 max-pool (used when the un-pooled activation is also routed elsewhere)
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out,
                                                          int ldo, int C, int B, int H, int W) {
  const int G = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const int64_t total = (int64_t)B * Ho * Wo * G;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int g = (int)(idx % G);
    int64_t po = idx / G;
    int xo = (int)(po % Wo);
    int64_t t = po / Wo;
    int yo = (int)(t % Ho);
    int64_t b = t / Ho;
    int64_t p00 = (b * H + 2 * yo) * W + 2 * xo;
    int c = g * 4;
    f32x4 v0 = *reinterpret_cast<const f32x4*>(x + p00 * ldx + c);
    f32x4 v1 = *reinterpret_cast<const f32x4*>(x + (p00 + 1) * ldx + c);
    f32x4 v2 = *reinterpret_cast<const f32x4*>(x + (p00 + W) * ldx + c);
    f32x4 v3 = *reinterpret_cast<const f32x4*>(x + (p00 + W + 1) * ldx + c);
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float m = v0[k];
      if (v1[k] > m) m = v1[k];
      if (v2[k] > m) m = v2[k];
      if (v3[k] > m) m = v3[k];
      r[k] = m;
    }
    *reinterpret_cast<f32x4*>(out + po * ldo + c) = r;
  }
}

// dx[window] (+)= g at the first maximum of the window, 0 elsewhere
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ g, int ldg, float* __restrict__ dx,
                                                          int lddx, int C, int B, int H, int W, int accumulate) {
  const int G = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const int64_t total = (int64_t)B * Ho * Wo * G;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int gg = (int)(idx % G);
    int64_t po = idx / G;
    int xo = (int)(po % Wo);
    int64_t t = po / Wo;
    int yo = (int)(t % Ho);
    int64_t b = t / Ho;
    int64_t p00 = (b * H + 2 * yo) * W + 2 * xo;
    const int64_t poff[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
    int c = gg * 4;
    f32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(x + poff[q] * ldx + c);
    f32x4 gv = *reinterpret_cast<const f32x4*>(g + po * ldg + c);
    int sel[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float m = v[0][k];
      int s = 0;
#pragma unroll
      for (int q = 1; q < 4; ++q)
        if (v[q][k] > m) { m = v[q][k]; s = q; }
      sel[k] = s;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 r;
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = sel[k] == q ? gv[k] : 0.f;
      float* d = dx + poff[q] * lddx + c;
      if (accumulate) {
        f32x4 o = *reinterpret_cast<const f32x4*>(d);
        r[0] += o[0]; r[1] += o[1]; r[2] += o[2]; r[3] += o[3];
      }
      *reinterpret_cast<f32x4*>(d) = r;
    }
  }
}

int ssp_nchw_to_nhwc_launch(const float* src, float* dst, int B, int C, int H, int W, int Cp, int ld,
                            hipStream_t stream) {
  SSP_CHECK_ARG(Cp >= C && ld >= Cp, "nchw_to_nhwc: need ld >= Cp >= C");
  SSP_CHECK_ARG(!(Cp == 4 && C <= 4) || (ld % 4 == 0), "nchw_to_nhwc: ld must be a multiple of 4");
  const int64_t total = (int64_t)B * H * W;
  SspProfScope prof(SSP_PROF_LAYOUT, stream, 4.0 * total * (C + Cp));
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(elem_grid(total)), dim3(256), 0, stream, src, dst, C, Cp, (int64_t)H * W,
                     total, ld);
  SSP_CHECK_LAUNCH("nchw_to_nhwc");
  return SSP_OK;
}

int ssp_u8hwc_to_nhwc_launch(const unsigned char* src, float* dst, int B, int H, int W, int C, int Cp, int ld,
                             hipStream_t stream) {
  SSP_CHECK_ARG(src != nullptr && dst != nullptr, "u8hwc_to_nhwc: null buffer");
  SSP_CHECK_ARG(C > 0 && Cp >= C && ld >= Cp, "u8hwc_to_nhwc: need ld >= Cp >= C > 0");
  const int64_t total = (int64_t)B * H * W;
  SSP_CHECK_ARG(total > 0, "u8hwc_to_nhwc: empty batch");
  const int fast = (C == 3 && Cp == 4 && ld == 4 && (((uintptr_t)src & 3) == 0) && (((uintptr_t)dst & 15) == 0)) ? 1 : 0;
  SspProfScope prof(SSP_PROF_LAYOUT, stream, (double)total * (C + 4.0 * Cp));
  hipLaunchKernelGGL(u8hwc_to_nhwc_kernel, dim3(elem_grid(fast ? (total + 3) / 4 : total)), dim3(256), 0, stream, src, dst,
                     C, Cp, total, ld, fast);
  SSP_CHECK_LAUNCH("u8hwc_to_nhwc");
  return SSP_OK;
}

int ssp_nhwc_to_nchw_launch(const float* src, float* dst, int B, int C, int H, int W, int ld, hipStream_t stream) {
  SSP_CHECK_ARG(ld >= C, "nhwc_to_nchw: need ld >= C");
  const int64_t total = (int64_t)B * C * H * W;
  SspProfScope prof(SSP_PROF_LAYOUT, stream, 8.0 * total);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(elem_grid(total)), dim3(256), 0, stream, src, dst, C, (int64_t)H * W,
                     total, ld);
  SSP_CHECK_LAUNCH("nhwc_to_nchw");
  return SSP_OK;
}

int ssp_repack_fwd_launch(const float* w, float* out, int Cout, int Cin, int Cinp, int R, hipStream_t stream) {
  SSP_CHECK_ARG(R == 1 || R == 3, "repack_fwd: R must be 1 or 3");
  SSP_CHECK_ARG(Cinp >= Cin, "repack_fwd: Cinp < Cin");
  SspProfScope prof(SSP_PROF_LAYOUT, stream, 8.0 * Cout * Cin * R * R);
  hipLaunchKernelGGL(repack_fwd_kernel, dim3(ssp_cdiv(Cinp, 64), Cout), dim3(256), 0, stream, w, out, Cout, Cin, Cinp,
                     R * R);
  SSP_CHECK_LAUNCH("repack_fwd");
  return SSP_OK;
}

int ssp_unpack_grad_launch(const float* dwp, float* grad, int Cout, int Cin, int Cinp, int R, hipStream_t stream) {
  SSP_CHECK_ARG(R == 1 || R == 3, "unpack_grad: R must be 1 or 3");
  SspProfScope prof(SSP_PROF_LAYOUT, stream, 8.0 * Cout * Cin * R * R);
  hipLaunchKernelGGL(unpack_grad_kernel, dim3(ssp_cdiv(Cin, 64), Cout), dim3(256), 0, stream, dwp, grad, Cout, Cin, Cinp,
                     R * R);
  SSP_CHECK_LAUNCH("unpack_grad");
  return SSP_OK;
}

int ssp_repack_dgrad_launch(const float* w, float* out, int Cout, int Cin, int Coutp, int R, hipStream_t stream) {
  SSP_CHECK_ARG(R == 1 || R == 3, "repack_dgrad: R must be 1 or 3");
  SSP_CHECK_ARG(Coutp >= Cout, "repack_dgrad: Coutp < Cout");
  SspProfScope prof(SSP_PROF_LAYOUT, stream, 8.0 * Cout * Cin * R * R);
  hipLaunchKernelGGL(repack_dgrad_kernel, dim3(ssp_cdiv(Cin, 32), ssp_cdiv(Coutp, 32)), dim3(256), 0, stream, w, out,
                     Cout, Cin, Coutp, R * R);
  SSP_CHECK_LAUNCH("repack_dgrad");
  return SSP_OK;
}

int ssp_repack_dgrad_packed_launch(const float* wp, float* out, int Cout, int Cin, int Coutp, int R, hipStream_t stream) {
  SSP_CHECK_ARG(R == 1 || R == 3, "repack_dgrad_packed: R must be 1 or 3");
  SSP_CHECK_ARG(Coutp >= Cout && Cout > 0 && Cin > 0, "repack_dgrad_packed: need Coutp >= Cout > 0, Cin > 0");
  SspProfScope prof(SSP_PROF_LAYOUT, stream, 8.0 * Cout * Cin * R * R);
  hipLaunchKernelGGL(repack_dgrad_packed_kernel, dim3(ssp_cdiv(Cin, 32), ssp_cdiv(Coutp, 32), R * R), dim3(256), 0, stream,
                     wp, out, Cout, Cin, Coutp, R * R);
  SSP_CHECK_LAUNCH("repack_dgrad_packed");
  return SSP_OK;
}

int ssp_reorg_launch(const float* src, int lds_, float* dst, int ldd, int C, int B, int H, int W, int backward,
                     int accumulate, hipStream_t stream) {
  SSP_CHECK_ARG(C % 4 == 0 && lds_ % 4 == 0 && ldd % 4 == 0, "reorg: C and strides must be multiples of 4");
  SSP_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "reorg: H and W must be even");
  const int64_t total = (int64_t)B * H * W * (C / 4);
  SspProfScope prof(SSP_PROF_LAYOUT, stream, 32.0 * total);
  hipLaunchKernelGGL(reorg_kernel, dim3(elem_grid(total)), dim3(256), 0, stream, src, lds_, dst, ldd, C, B, H, W,
                     backward, accumulate);
  SSP_CHECK_LAUNCH("reorg");
  return SSP_OK;
}

int ssp_copy_channels_launch(const float* src, int lds_, float* dst, int ldd, int C, int64_t M, int accumulate,
                             hipStream_t stream) {
  SSP_CHECK_ARG(C % 4 == 0 && lds_ % 4 == 0 && ldd % 4 == 0, "copy_channels: C and strides must be multiples of 4");
  const int64_t total = M * (C / 4);
  SspProfScope prof(SSP_PROF_LAYOUT, stream, 32.0 * total);
  hipLaunchKernelGGL(copy_channels_kernel, dim3(elem_grid(total)), dim3(256), 0, stream, src, lds_, dst, ldd, C, M,
                     accumulate);
  SSP_CHECK_LAUNCH("copy_channels");
  return SSP_OK;
}

int ssp_maxpool_fwd_launch(const float* x, int ldx, float* out, int ldo, int C, int B, int H, int W,
                           hipStream_t stream) {
  SSP_CHECK_ARG(C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0, "maxpool_fwd: C and strides must be multiples of 4");
  SSP_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "maxpool_fwd: H and W must be even");
  const int64_t total = (int64_t)B * (H / 2) * (W / 2) * (C / 4);
  SspProfScope prof(SSP_PROF_LAYOUT, stream, 80.0 * total);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(elem_grid(total)), dim3(256), 0, stream, x, ldx, out, ldo, C, B, H, W);
  SSP_CHECK_LAUNCH("maxpool_fwd");
  return SSP_OK;
}

int ssp_maxpool_bwd_launch(const float* x, int ldx, const float* g, int ldg, float* dx, int lddx, int C, int B, int H,
                           int W, int accumulate, hipStream_t stream) {
  SSP_CHECK_ARG(C % 4 == 0 && ldx % 4 == 0 && ldg % 4 == 0 && lddx % 4 == 0, "maxpool_bwd: C and strides must be multiples of 4");
  SSP_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "maxpool_bwd: H and W must be even");
  const int64_t total = (int64_t)B * (H / 2) * (W / 2) * (C / 4);
  SspProfScope prof(SSP_PROF_LAYOUT, stream, 144.0 * total);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(elem_grid(total)), dim3(256), 0, stream, x, ldx, g, ldg, dx, lddx, C, B, H,
                     W, accumulate);
  SSP_CHECK_LAUNCH("maxpool_bwd");
  return SSP_OK;
}
