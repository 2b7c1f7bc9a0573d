// BatchNorm(eps=1e-4) + LeakyReLU(0.1) (+ fused 2x2/2 max-pool) forward and backward, HBM-bound kernels.
//
// Replaces nn.BatchNorm2d / nn.LeakyReLU / nn.MaxPool2d of the reference conv blocks
// (/root/reference/darknet.py:157,162,172) and their autograd backward.
//
// Forward (training): the conv epilogue (conv_igemm.hip) leaves one (mean, M2) pair per channel per
// M tile; bn_fwd_finalize Chan-combines them in fp64 (PyTorch's CPU batch_norm accumulates in double),
// emits mean / invstd / scale=gamma*invstd / shift=beta-mean*scale and updates the running statistics
// (momentum 0.1, unbiased variance).  bn_act_fwd then reads the raw conv output once and writes
// leaky(scale*x+shift), optionally max-pooled, into a (possibly channel-sliced) NHWC buffer.
//
// Backward: nothing but the raw conv output is kept.  bn_act_bwd_reduce recomputes y = scale*x+shift,
// the leaky slope and (when pooled) the arg-max of each 2x2 window (first maximum in scan order wins,
// as in ATen's max_pool2d) and reduces sum(dy), sum(dy*xhat) per channel; bn_act_bwd_apply forms
//   dx = scale * (dy - sum(dy)/N - xhat * sum(dy*xhat)/N)
// in place over the raw conv output.
#include "ssp_common.h"

__device__ __forceinline__ void chan_combine_d(double& n, double& mean, double& m2, double nb, double mb, double m2b) {
  double nt = n + nb;
  if (nt > 0.0) {
    double d = mb - mean;
    double f = nb / nt;
    mean += d * f;
    m2 += m2b + d * d * n * f;
    n = nt;
  }
}

// one workgroup per channel
__global__ void __launch_bounds__(256) bn_fwd_finalize_kernel(const float* __restrict__ stats, int ntile, int BM, int M,
                                                              int C, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* rmean, float* rvar,
                                                              float momentum, float eps, float* mean_out,
                                                              float* invstd_out, float* scale, float* shift) {
  const int c = blockIdx.x, tid = threadIdx.x;
  double n = 0.0, mean = 0.0, m2 = 0.0;
  // BM == 0: counted format (Winograd finishing pass, conv_wino.hip) - the tiles' pixel counts follow the (mean, M2) pairs
  const float* counts = stats + (int64_t)ntile * C * 2;
  for (int t = tid; t < ntile; t += 256) {
    const double cnt = BM > 0 ? (double)min(BM, M - t * BM) : (double)counts[t];
    const float* s = stats + ((int64_t)t * C + c) * 2;
    chan_combine_d(n, mean, m2, cnt, (double)s[0], (double)s[1]);
  }
  __shared__ double sn[256], smean[256], sm2[256];
  sn[tid] = n; smean[tid] = mean; sm2[tid] = m2;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      double a = sn[tid], b = smean[tid], d = sm2[tid];
      chan_combine_d(a, b, d, sn[tid + off], smean[tid + off], sm2[tid + off]);
      sn[tid] = a; smean[tid] = b; sm2[tid] = d;
    }
    __syncthreads();
  }
  if (tid == 0) {
    double N = sn[0], mu = smean[0], var = sm2[0] / N;
    double istd = 1.0 / sqrt(var + (double)eps);
    float g = gamma[c], b = beta[c];
    mean_out[c] = (float)mu;
    invstd_out[c] = (float)istd;
    float sc = g * (float)istd;
    scale[c] = sc;
    shift[c] = b - (float)mu * sc;
    double unbiased = N > 1.0 ? sm2[0] / (N - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mu;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
  }
}

// eval mode: normalise with the running statistics
__global__ void bn_eval_prepare_kernel(int C, const float* gamma, const float* beta, const float* rmean,
                                       const float* rvar, float eps, float* mean_out, float* invstd_out,
                                       float* scale, float* shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    float istd = 1.f / sqrtf(rvar[c] + eps);
    float sc = gamma[c] * istd;
    mean_out[c] = rmean[c];
    invstd_out[c] = istd;
    scale[c] = sc;
    shift[c] = beta[c] - rmean[c] * sc;
  }
}

__device__ __forceinline__ float leaky(float y, float slope) { return y > 0.f ? y : y * slope; }

// thread = 4 channels of one output pixel
__global__ void __launch_bounds__(256) bn_act_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out,
                                                         int ldo, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int C, int B, int H, int W,
                                                         int pool, float slope) {
  const int G = C >> 2;
  const int Ho = pool ? H >> 1 : H, Wo = pool ? W >> 1 : W;
  const int64_t total = (int64_t)B * Ho * Wo * G;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int g = (int)(idx % G);
    int64_t po = idx / G;
    int c = g * 4;
    f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
    f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c);
    f32x4 r;
    if (!pool) {
      f32x4 v = *reinterpret_cast<const f32x4*>(x + po * ldx + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = leaky(v[k] * sc[k] + sh[k], slope);
    } else {
      int xo = (int)(po % Wo);
      int64_t t = po / Wo;
      int yo = (int)(t % Ho);
      int64_t b = t / Ho;
      int64_t p00 = (b * H + 2 * yo) * W + 2 * xo;
      f32x4 v0 = *reinterpret_cast<const f32x4*>(x + p00 * ldx + c);
      f32x4 v1 = *reinterpret_cast<const f32x4*>(x + (p00 + 1) * ldx + c);
      f32x4 v2 = *reinterpret_cast<const f32x4*>(x + (p00 + W) * ldx + c);
      f32x4 v3 = *reinterpret_cast<const f32x4*>(x + (p00 + W + 1) * ldx + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float a0 = leaky(v0[k] * sc[k] + sh[k], slope), a1 = leaky(v1[k] * sc[k] + sh[k], slope);
        float a2 = leaky(v2[k] * sc[k] + sh[k], slope), a3 = leaky(v3[k] * sc[k] + sh[k], slope);
        float m = a0;
        if (a1 > m) m = a1;
        if (a2 > m) m = a2;
        if (a3 > m) m = a3;
        r[k] = m;
      }
    }
    *reinterpret_cast<f32x4*>(out + po * ldo + c) = r;
  }
}

// Recompute the activation-side gradient of one (output pixel, 4 channels) item.
// Returns, per channel k: dyv[k] (gradient wrt the BN output at the selected raw pixel), xh[k] (its xhat) and
// sel[k] (which of the 4 raw pixels it belongs to; 0 when not pooled).
struct BwdItem {
  f32x4 dyv, xh;
  int sel[4];
};

__device__ __forceinline__ BwdItem bwd_item(const float* __restrict__ x, int ldx, const float* __restrict__ g,
                                            int ldg, int64_t po, int64_t p00, int W, int c, int pool, float slope,
                                            const f32x4& sc, const f32x4& sh, const f32x4& mu, const f32x4& is) {
  BwdItem r;
  f32x4 gv = *reinterpret_cast<const f32x4*>(g + po * ldg + c);
  if (!pool) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + po * ldx + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float y = v[k] * sc[k] + sh[k];
      r.dyv[k] = y > 0.f ? gv[k] : gv[k] * slope;
      r.xh[k] = (v[k] - mu[k]) * is[k];
      r.sel[k] = 0;
    }
  } else {
    f32x4 v0 = *reinterpret_cast<const f32x4*>(x + p00 * ldx + c);
    f32x4 v1 = *reinterpret_cast<const f32x4*>(x + (p00 + 1) * ldx + c);
    f32x4 v2 = *reinterpret_cast<const f32x4*>(x + (p00 + W) * ldx + c);
    f32x4 v3 = *reinterpret_cast<const f32x4*>(x + (p00 + W + 1) * ldx + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float vv[4] = {v0[k], v1[k], v2[k], v3[k]};
      float y0 = vv[0] * sc[k] + sh[k];
      float best = leaky(y0, slope), ybest = y0, xbest = vv[0];
      int s = 0;
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        float y = vv[q] * sc[k] + sh[k];
        float a = leaky(y, slope);
        if (a > best) { best = a; ybest = y; xbest = vv[q]; s = q; }
      }
      r.dyv[k] = ybest > 0.f ? gv[k] : gv[k] * slope;
      r.xh[k] = (xbest - mu[k]) * is[k];
      r.sel[k] = s;
    }
  }
  return r;
}

// grid.x workgroups stride over output pixels; partial[blk][C][2] = (sum dy, sum dy*xhat)
__global__ void __launch_bounds__(256) bn_act_bwd_reduce_kernel(const float* __restrict__ x, int ldx,
                                                                const float* __restrict__ g, int ldg,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, int C, int B, int H,
                                                                int W, int pool, float slope, float* partial,
                                                                float* acc_dgamma, float* acc_dbeta) {
  const int G = C >> 2;
  const int gpb = G < 256 ? G : 256;     // channel groups per block
  const int ppb = 256 / gpb;             // pixel lanes per block
  const int tid = threadIdx.x;
  const int gl = tid % gpb, pp = tid / gpb;
  const int g4 = blockIdx.y * gpb + gl;
  const bool active = (pp < ppb) && (g4 < G);
  const int Ho = pool ? H >> 1 : H, Wo = pool ? W >> 1 : W;
  const int64_t npix = (int64_t)B * Ho * Wo;
  // fp64 accumulators: the two sums cancel heavily (dy has mixed signs) and feed mean(dy) / mean(dy*xhat), which the
  // apply pass subtracts from EVERY pixel - an error there is coherent, and a following filter gradient against a
  // non-centred input (the image itself for the first layer) amplifies it by sqrt(#pixels): with fp32 per-thread sums
  // the first layer's dW sat 1.6e-3 from the reference at B = 64 (PyTorch's CPU batch_norm_backward sums in double too)
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  if (active) {
    const int c = g4 * 4;
    f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
    f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    for (int64_t po = (int64_t)blockIdx.x * ppb + pp; po < npix; po += (int64_t)gridDim.x * ppb) {
      int64_t p00 = 0;
      if (pool) {
        int xo = (int)(po % Wo);
        int64_t t = po / Wo;
        int yo = (int)(t % Ho);
        int64_t b = t / Ho;
        p00 = (b * H + 2 * yo) * W + 2 * xo;
      }
      BwdItem it = bwd_item(x, ldx, g, ldg, po, p00, W, c, pool, slope, sc, sh, mu, is);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s1[k] += (double)it.dyv[k];
        s2[k] += (double)it.dyv[k] * (double)it.xh[k];
      }
    }
  }
  __shared__ double r1[256][4], r2[256][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { r1[tid][k] = s1[k]; r2[tid][k] = s2[k]; }
  __syncthreads();
  if (tid < gpb && g4 < G) {
    double a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = r1[tid][k]; b[k] = r2[tid][k]; }
    for (int q = 1; q < ppb; ++q) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { a[k] += r1[tid + q * gpb][k]; b[k] += r2[tid + q * gpb][k]; }
    }
    if (acc_dbeta != nullptr) {
      // single-pass mode: the (<= 1024) workgroup sums go straight into the zero-initialised gradients with fp32
      // hardware atomics; no finalize launch sits between this kernel and the apply pass
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        atomicAdd(acc_dbeta + g4 * 4 + k, (float)a[k]);
        atomicAdd(acc_dgamma + g4 * 4 + k, (float)b[k]);
      }
    } else {
      float* dst = partial + ((int64_t)blockIdx.x * C + g4 * 4) * 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) { dst[2 * k] = (float)a[k]; dst[2 * k + 1] = (float)b[k]; }
    }
  }
}

// fp64 sum of the per-block (or per-tile) partial pairs.  One workgroup per 4 channels: a partial row holds their 8
// floats contiguously (two float4), thread = (row lane 0..127, float4 half), fp64 accumulation, LDS tree over the row
// lanes.  zero_after: the rows are cleared once read (the fused data-gradient path accumulates into them with atomics
// when it has more tiles than rows, and expects zeros the next time).
__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(float* __restrict__ partial, int nblk, int C,
                                                              double inv_n, int training, int zero_after, float* dgamma,
                                                              float* dbeta, float* c1, float* c2) {
  const int tid = threadIdx.x, half = tid & 1, lane = tid >> 1;
  const int c0 = blockIdx.x * 4;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int t = lane; t < nblk; t += 128) {
    float* s = partial + ((int64_t)t * C + c0) * 2 + half * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(s);
    acc[0] += (double)v[0]; acc[1] += (double)v[1]; acc[2] += (double)v[2]; acc[3] += (double)v[3];
    if (zero_after) *reinterpret_cast<f32x4*>(s) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __shared__ double red[256][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) red[tid][k] = acc[k];
  __syncthreads();
  for (int off = 128; off >= 2; off >>= 1) {       // keeps the float4 half (tid & 1) apart
    if (tid < off) {
#pragma unroll
      for (int k = 0; k < 4; ++k) red[tid][k] += red[tid + off][k];
    }
    __syncthreads();
  }
  if (tid < 4) {
    // channel c0 + tid: (sum dy, sum dy*xhat) = floats 2*tid, 2*tid+1 of the 8 -> half = tid >> 1, k = (tid & 1) * 2
    const int c = c0 + tid;
    if (c < C) {
      const double a = red[tid >> 1][(tid & 1) * 2], b = red[tid >> 1][(tid & 1) * 2 + 1];
      dbeta[c] = (float)a;
      dgamma[c] = (float)b;
      c1[c] = training ? (float)(a * inv_n) : 0.f;
      c2[c] = training ? (float)(b * inv_n) : 0.f;
    }
  }
}

// dx may alias x (each thread reads its own raw pixels before writing them)
__global__ void __launch_bounds__(256) bn_act_bwd_apply_kernel(const float* x, int ldx, const float* __restrict__ g,
                                                               int ldg, float* dx, int lddx,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ c1,
                                                               const float* __restrict__ c2, float kscale, int C, int B,
                                                               int H, int W, int pool, float slope) {
  // c1 / c2: per-channel mean(dy), mean(dy * xhat) (kscale = 1), or the SUMS dbeta / dgamma with kscale = 1 / N
  // (single-pass mode; kscale = 0 in eval mode, where the statistics are constants)
  const int G = C >> 2;
  const int Ho = pool ? H >> 1 : H, Wo = pool ? W >> 1 : W;
  const int64_t total = (int64_t)B * Ho * Wo * G;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int g4 = (int)(idx % G);
    int64_t po = idx / G;
    int c = g4 * 4;
    f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
    f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    f32x4 k1 = *reinterpret_cast<const f32x4*>(c1 + c), k2 = *reinterpret_cast<const f32x4*>(c2 + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) { k1[k] *= kscale; k2[k] *= kscale; }
    if (!pool) {
      BwdItem it = bwd_item(x, ldx, g, ldg, po, 0, W, c, 0, slope, sc, sh, mu, is);
      f32x4 r;
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = sc[k] * (it.dyv[k] - k1[k] - it.xh[k] * k2[k]);
      *reinterpret_cast<f32x4*>(dx + po * lddx + c) = r;
    } else {
      int xo = (int)(po % Wo);
      int64_t t = po / Wo;
      int yo = (int)(t % Ho);
      int64_t b = t / Ho;
      int64_t p00 = (b * H + 2 * yo) * W + 2 * xo;
      const int64_t poff[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
      f32x4 xv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) xv[q] = *reinterpret_cast<const f32x4*>(x + poff[q] * ldx + c);
      BwdItem it = bwd_item(x, ldx, g, ldg, po, p00, W, c, 1, slope, sc, sh, mu, is);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float xh = (xv[q][k] - mu[k]) * is[k];
          float dyq = (it.sel[k] == q) ? it.dyv[k] : 0.f;
          r[k] = sc[k] * (dyq - k1[k] - xh * k2[k]);
        }
        *reinterpret_cast<f32x4*>(dx + poff[q] * lddx + c) = r;
      }
    }
  }
}

// per-channel sum over pixels (bias gradient of the linear head conv); one workgroup per channel
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ g, int ldg, int64_t M, float* out) {
  const int c = blockIdx.x, tid = threadIdx.x;
  double s = 0.0;
  for (int64_t m = tid; m < M; m += 256) s += (double)g[m * ldg + c];
  __shared__ double red[256];
  red[tid] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) red[tid] += red[tid + off];
    __syncthreads();
  }
  if (tid == 0) out[c] = (float)red[0];
}

static int elem_grid(int64_t total) {
  int64_t blocks = (total + 255) / 256;
  const int64_t cap = 256 * 16;  // 16 resident 256-thread workgroups' worth per CU, grid-stride beyond that
  return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

int ssp_bn_fwd_finalize_launch(const float* stats, int ntile, int BM, int M, int C, const float* gamma,
                               const float* beta, float* rmean, float* rvar, float momentum, float eps, float* mean,
                               float* invstd, float* scale, float* shift, hipStream_t stream) {
  SSP_CHECK_ARG(C > 0 && ntile > 0 && M > 0, "bn_fwd_finalize: bad sizes");
  SspProfScope prof(SSP_PROF_BN_ACT, stream, 0.0);
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(C), dim3(256), 0, stream, stats, ntile, BM, M, C, gamma, beta, rmean,
                     rvar, momentum, eps, mean, invstd, scale, shift);
  SSP_CHECK_LAUNCH("bn_fwd_finalize");
  return SSP_OK;
}

int ssp_bn_eval_prepare_launch(int C, const float* gamma, const float* beta, const float* rmean, const float* rvar,
                               float eps, float* mean, float* invstd, float* scale, float* shift, hipStream_t stream) {
  SSP_CHECK_ARG(C > 0, "bn_eval_prepare: bad sizes");
  SspProfScope prof(SSP_PROF_BN_ACT, stream, 0.0);
  hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(ssp_cdiv(C, 256)), dim3(256), 0, stream, C, gamma, beta, rmean, rvar,
                     eps, mean, invstd, scale, shift);
  SSP_CHECK_LAUNCH("bn_eval_prepare");
  return SSP_OK;
}

int ssp_bn_act_fwd_launch(const float* x, int ldx, float* out, int ldo, const float* scale, const float* shift, int C,
                          int B, int H, int W, int pool, float slope, hipStream_t stream) {
  SSP_CHECK_ARG(C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0, "bn_act_fwd: C, ldx, ldo must be multiples of 4");
  SSP_CHECK_ARG(!pool || (H % 2 == 0 && W % 2 == 0), "bn_act_fwd: pooled maps need even H, W");
  const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
  const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
  SspProfScope prof(SSP_PROF_BN_ACT, stream, 4.0 * C * ((double)B * H * W + (double)B * Ho * Wo));
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(elem_grid(total)), dim3(256), 0, stream, x, ldx, out, ldo, scale, shift, C,
                     B, H, W, pool, slope);
  SSP_CHECK_LAUNCH("bn_act_fwd");
  return SSP_OK;
}

// workspace floats needed by ssp_bn_act_bwd (partials): nblk * C * 2
int ssp_bn_bwd_blocks_impl(void) { return 1024; }

int ssp_bn_act_bwd_launch(const float* x, int ldx, const float* g, int ldg, float* dx, int lddx, const float* scale,
                          const float* shift, const float* mean, const float* invstd, int C, int B, int H, int W,
                          int pool, float slope, int training, float* partial, float* dgamma, float* dbeta, float* c1,
                          float* c2, hipStream_t stream) {
  SSP_CHECK_ARG(C % 4 == 0 && ldx % 4 == 0 && ldg % 4 == 0 && lddx % 4 == 0, "bn_act_bwd: C and strides must be multiples of 4");
  SSP_CHECK_ARG(!pool || (H % 2 == 0 && W % 2 == 0), "bn_act_bwd: pooled maps need even H, W");
  const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
  const int G = C / 4;
  const int gpb = G < 256 ? G : 256;
  const int ppb = 256 / gpb;
  const int64_t npix = (int64_t)B * Ho * Wo;
  int nblk = (int)((npix + ppb - 1) / ppb);
  if (nblk > ssp_bn_bwd_blocks_impl()) nblk = ssp_bn_bwd_blocks_impl();
  if (nblk < 1) nblk = 1;
  SspProfScope prof(SSP_PROF_BN_ACT, stream, 4.0 * C * (3.0 * (double)B * H * W + 2.0 * (double)npix));
  const int64_t total = npix * G;
  if (partial == nullptr) {
    // single pass: dgamma / dbeta zero-initialised by the caller, accumulated with atomics, read back by the apply pass
    SSP_CHECK_ARG((((uintptr_t)dgamma | (uintptr_t)dbeta) & 15) == 0, "bn_act_bwd: dgamma / dbeta must be 16-byte aligned");
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3(nblk, ssp_cdiv(G, gpb)), dim3(256), 0, stream, x, ldx, g, ldg,
                       scale, shift, mean, invstd, C, B, H, W, pool, slope, (float*)nullptr, dgamma, dbeta);
    SSP_CHECK_LAUNCH("bn_act_bwd_reduce");
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(elem_grid(total)), dim3(256), 0, stream, x, ldx, g, ldg, dx, lddx,
                       scale, shift, mean, invstd, dbeta, dgamma, training ? (float)(1.0 / ((double)B * H * W)) : 0.f, C, B,
                       H, W, pool, slope);
  } else {
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3(nblk, ssp_cdiv(G, gpb)), dim3(256), 0, stream, x, ldx, g, ldg,
                       scale, shift, mean, invstd, C, B, H, W, pool, slope, partial, (float*)nullptr, (float*)nullptr);
    SSP_CHECK_LAUNCH("bn_act_bwd_reduce");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ssp_cdiv(C, 4)), dim3(256), 0, stream, partial, nblk, C,
                       1.0 / ((double)B * H * W), training, 0, dgamma, dbeta, c1, c2);
    SSP_CHECK_LAUNCH("bn_bwd_finalize");
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(elem_grid(total)), dim3(256), 0, stream, x, ldx, g, ldg, dx, lddx,
                       scale, shift, mean, invstd, c1, c2, 1.f, C, B, H, W, pool, slope);
  }
  SSP_CHECK_LAUNCH("bn_act_bwd_apply");
  return SSP_OK;
}

// The two reductions were already done by the producing data-gradient launch (ssp_conv_dgrad_bnbwd: one (sum dy,
// sum dy * xhat) pair per M tile and channel): fp64 finalize over the tiles, then the apply pass.  Un-pooled blocks only.
int ssp_bn_act_bwd_partials_launch(const float* x, int ldx, const float* g, int ldg, float* dx, int lddx,
                                   const float* scale, const float* shift, const float* mean, const float* invstd,
                                   int C, int B, int H, int W, float slope, int training, float* partial,
                                   int npartial, int zero_after, float* dgamma, float* dbeta, float* c1, float* c2,
                                   hipStream_t stream) {
  SSP_CHECK_ARG(C % 4 == 0 && ldx % 4 == 0 && ldg % 4 == 0 && lddx % 4 == 0, "bn_act_bwd_partials: C and strides must be multiples of 4");
  SSP_CHECK_ARG(partial != nullptr && npartial > 0 && (((uintptr_t)partial) & 15) == 0, "bn_act_bwd_partials: no (aligned) partial sums");
  const int64_t npix = (int64_t)B * H * W;
  SspProfScope prof(SSP_PROF_BN_ACT, stream, 4.0 * C * 3.0 * (double)npix);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ssp_cdiv(C, 4)), dim3(256), 0, stream, partial, npartial, C,
                     1.0 / (double)npix, training, zero_after, dgamma, dbeta, c1, c2);
  SSP_CHECK_LAUNCH("bn_bwd_finalize");
  hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(elem_grid(npix * (C / 4))), dim3(256), 0, stream, x, ldx, g, ldg, dx,
                     lddx, scale, shift, mean, invstd, c1, c2, 1.f, C, B, H, W, 0, slope);
  SSP_CHECK_LAUNCH("bn_act_bwd_apply");
  return SSP_OK;
}

int ssp_bn_bwd_finalize_launch(float* partial, int npartial, int C, int64_t npix, int training, int zero_after,
                               float* dgamma, float* dbeta, float* c1, float* c2, hipStream_t stream) {
  SSP_CHECK_ARG(C % 4 == 0 && partial != nullptr && npartial > 0 && npix > 0 && (((uintptr_t)partial) & 15) == 0,
                "bn_bwd_finalize: C %% 4 == 0, aligned partial rows");
  SspProfScope prof(SSP_PROF_BN_ACT, stream, 0.0);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ssp_cdiv(C, 4)), dim3(256), 0, stream, partial, npartial, C,
                     1.0 / (double)npix, training, zero_after, dgamma, dbeta, c1, c2);
  SSP_CHECK_LAUNCH("bn_bwd_finalize");
  return SSP_OK;
}

int ssp_colsum_launch(const float* g, int ldg, int64_t M, int C, float* out, hipStream_t stream) {
  SSP_CHECK_ARG(C > 0 && M > 0, "colsum: bad sizes");
  SspProfScope prof(SSP_PROF_BN_ACT, stream, 0.0);
  hipLaunchKernelGGL(colsum_kernel, dim3(C), dim3(256), 0, stream, g, ldg, M, out);
  SSP_CHECK_LAUNCH("colsum");
  return SSP_OK;
}
