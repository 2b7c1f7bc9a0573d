// Winograd F(2x2, 3x3) transforms for the deep 3x3 layers (the 13x13 ... 52x52 maps with >= 128 channels).
//
// The fp32 MFMA is the binding resource of the training step (conv_igemm_dma.hip runs it 88 - 93 % busy), and on gfx950 it
// has no faster fp32 form - so the remaining lever on those layers is arithmetic: a 3x3 stride-1 convolution evaluated on
// 2x2 output tiles needs 16 multiplies per tile and channel pair instead of 36 (Lavin & Gray, "Fast Algorithms for
// Convolutional Neural Networks", F(2x2, 3x3)).  All of it stays in fp32, the transforms use the constants 1, -1 and 1/2
// only (exact in binary), and the result agrees with the direct kernel to ~1e-6 of the output's range - inside the 1e-5
// the engine's verify-after-tune demands of any plan before it admits it.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, summed over input channels
//
// becomes, with the channel sum pulled inside, 16 independent GEMMs (one per position xi of the 4x4 transform domain):
//   M_xi [T][Cout] = V_xi [T][Cin] * U_xi [Cout][Cin]^T,     T = B * ceil(H/2) * ceil(W/2) tiles
// which run as ONE batched launch of the LDS-direct implicit-GEMM kernel (R = 1, gridDim.y = 16).  Around it:
//   * wino_filter_kernel   U = G g G^T   from the [rows][tap][K] filter layout both passes already use (forward: the
//     channels-last parameter itself; data gradient: the flipped / transposed operand ssp_repack_dgrad_packed builds) -
//     weights only, so it runs on the side stream next to the repacks (202 MB -> 360 MB per pass at most);
//   * wino_input_kernel    V = B^T d B   one thread per (tile, 4 channels): 16 float4 loads (zero padding = the tile's
//     out-of-image positions), 32 adds per channel, 16 float4 stores into the 16 planes;  HBM-bound;
//   * the inverse transform A^T M A is the gather step of reduce_kernel<true> (conv_igemm.hip): the same finishing pass
//     as split-K (bias, BatchNorm statistics, accumulate, fused BatchNorm-backward reductions), reading 9 of the 16
//     planes per output pixel instead of summing K partials.
// Algorithmic FLOPs stay 2 * M * Cout * 9 * Cin (what the profiler books); the MFMA executes 16/36 of them (x 49/42.25
// for the tile padding of a 13 x 13 map).
#include "ssp_common.h"

struct WinoInArgs {
  const float* in;   // [B*H*W][ldin]
  float* V;          // [16][T][C]
  int H, W, C, ldin, th, tw;
  int64_t T;
  SspFastDiv div_c4, div_tw, div_th;
};

__global__ void __launch_bounds__(256) wino_input_kernel(WinoInArgs p) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c4n = p.C >> 2;
  if (gid >= p.T * c4n) return;
  const unsigned t = ssp_div((unsigned)gid, p.div_c4);      // T * C/4 < 2^31 is checked by the launcher
  const int c = (int)((unsigned)gid - t * (unsigned)c4n) * 4;
  const unsigned q = ssp_div(t, p.div_tw);                  // b * th + ty
  const int tx = (int)(t - q * (unsigned)p.tw);
  const unsigned b = ssp_div(q, p.div_th);
  const int ty = (int)(q - b * (unsigned)p.th);
  const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
  const float* base = p.in + ((int64_t)b * p.H * p.W) * p.ldin + c;
  f32x4 d[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = y0 + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = x0 + j;
      const bool ok = ((unsigned)y < (unsigned)p.H) && ((unsigned)x < (unsigned)p.W);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      d[i][j] = ok ? *reinterpret_cast<const f32x4*>(base + ((int64_t)y * p.W + x) * p.ldin) : z;
    }
  }
  // B^T d: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3), then the same combination over the columns
  f32x4 r[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r[0][j] = d[0][j] - d[2][j];
    r[1][j] = d[1][j] + d[2][j];
    r[2][j] = d[2][j] - d[1][j];
    r[3][j] = d[1][j] - d[3][j];
  }
  float* dst = p.V + (int64_t)t * p.C + c;
  const int64_t plane = p.T * p.C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<f32x4*>(dst + (int64_t)(i * 4 + 0) * plane) = r[i][0] - r[i][2];
    *reinterpret_cast<f32x4*>(dst + (int64_t)(i * 4 + 1) * plane) = r[i][1] + r[i][2];
    *reinterpret_cast<f32x4*>(dst + (int64_t)(i * 4 + 2) * plane) = r[i][2] - r[i][1];
    *reinterpret_cast<f32x4*>(dst + (int64_t)(i * 4 + 3) * plane) = r[i][1] - r[i][3];
  }
}

// U[xi][row][k] = (G g G^T)[xi] of the 3x3 filter g[tap] = w[row][tap][k]; thread = (row, 4 k's)
__global__ void __launch_bounds__(256) wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int rows, int K) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int k4n = K >> 2;
  if (gid >= (int64_t)rows * k4n) return;
  const int row = (int)(gid / k4n);
  const int k = (int)(gid - (int64_t)row * k4n) * 4;
  const float* src = w + ((int64_t)row * 9) * K + k;
  f32x4 g[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) g[i][j] = *reinterpret_cast<const f32x4*>(src + (int64_t)(i * 3 + j) * K);
  // G g: rows (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2); then the same over the columns
  f32x4 h[4][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    h[0][j] = g[0][j];
    h[1][j] = (g[0][j] + g[1][j] + g[2][j]) * 0.5f;
    h[2][j] = (g[0][j] - g[1][j] + g[2][j]) * 0.5f;
    h[3][j] = g[2][j];
  }
  float* dst = U + (int64_t)row * K + k;
  const int64_t plane = (int64_t)rows * K;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<f32x4*>(dst + (int64_t)(i * 4 + 0) * plane) = h[i][0];
    *reinterpret_cast<f32x4*>(dst + (int64_t)(i * 4 + 1) * plane) = (h[i][0] + h[i][1] + h[i][2]) * 0.5f;
    *reinterpret_cast<f32x4*>(dst + (int64_t)(i * 4 + 2) * plane) = (h[i][0] - h[i][1] + h[i][2]) * 0.5f;
    *reinterpret_cast<f32x4*>(dst + (int64_t)(i * 4 + 3) * plane) = h[i][2];
  }
}

// ---- filter gradient in the transform domain --------------------------------------------------------------------------
// dL/dU_xi [Cout][Cin] = sum_t dM_xi[t][Cout] (x) V_xi[t][Cin]   with   dM = A dY A^T  (the adjoint of the inverse transform:
// a 2x2 tile of the output gradient spread over the 4x4 domain, A = [1 0; 1 1; 1 -1; 0 -1]) and V the transformed input of
// the forward pass; then dL/dg = G^T (dL/dU) G.  The 16 sums over the tiles are 16 pixel-contraction GEMMs - one batched
// launch of the LDS-direct filter-gradient kernel (conv_wgrad_dma.hip, R = 1, gridDim.y = 16) - with 16/36 of the direct
// kernel's multiplies.
struct WinoOutGradArgs {
  const float* dy;   // [B*H*W][lddy]
  float* dM;         // [16][T][C]
  int H, W, C, lddy, th, tw;
  int64_t T;
  SspFastDiv div_c4, div_tw, div_th;
};

__global__ void __launch_bounds__(256) wino_outgrad_kernel(WinoOutGradArgs p) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c4n = p.C >> 2;
  if (gid >= p.T * c4n) return;
  const unsigned t = ssp_div((unsigned)gid, p.div_c4);
  const int c = (int)((unsigned)gid - t * (unsigned)c4n) * 4;
  const unsigned q = ssp_div(t, p.div_tw);
  const int tx = (int)(t - q * (unsigned)p.tw);
  const unsigned b = ssp_div(q, p.div_th);
  const int ty = (int)(q - b * (unsigned)p.th);
  const float* base = p.dy + ((int64_t)b * p.H * p.W) * p.lddy + c;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 d[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int y = 2 * ty + i, x = 2 * tx + j;
      d[i][j] = (y < p.H && x < p.W) ? *reinterpret_cast<const f32x4*>(base + ((int64_t)y * p.W + x) * p.lddy) : z;
    }
  // A d: rows (d0, d0 + d1, d0 - d1, -d1); then the same over the columns
  f32x4 r[4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    r[0][j] = d[0][j];
    r[1][j] = d[0][j] + d[1][j];
    r[2][j] = d[0][j] - d[1][j];
    r[3][j] = z - d[1][j];
  }
  float* dst = p.dM + (int64_t)t * p.C + c;
  const int64_t plane = p.T * p.C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<f32x4*>(dst + (int64_t)(i * 4 + 0) * plane) = r[i][0];
    *reinterpret_cast<f32x4*>(dst + (int64_t)(i * 4 + 1) * plane) = r[i][0] + r[i][1];
    *reinterpret_cast<f32x4*>(dst + (int64_t)(i * 4 + 2) * plane) = r[i][0] - r[i][1];
    *reinterpret_cast<f32x4*>(dst + (int64_t)(i * 4 + 3) * plane) = z - r[i][1];
  }
}

// dw[row][tap][k] += (G^T dU G)[tap]; thread = (row, 4 k's); dw is this launch's own (no other writer): plain read-add-write
__global__ void __launch_bounds__(256) wino_wgrad_finish_kernel(const float* __restrict__ dU, float* dw, int rows, int K) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int k4n = K >> 2;
  if (gid >= (int64_t)rows * k4n) return;
  const int row = (int)(gid / k4n);
  const int k = (int)(gid - (int64_t)row * k4n) * 4;
  const float* src = dU + (int64_t)row * K + k;
  const int64_t plane = (int64_t)rows * K;
  f32x4 u[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) u[i][j] = *reinterpret_cast<const f32x4*>(src + (int64_t)(i * 4 + j) * plane);
  // G^T u: rows (u0 + (u1 + u2) / 2, (u1 - u2) / 2, (u1 + u2) / 2 + u3); then the same over the columns
  f32x4 h[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[0][j] = u[0][j] + (u[1][j] + u[2][j]) * 0.5f;
    h[1][j] = (u[1][j] - u[2][j]) * 0.5f;
    h[2][j] = (u[1][j] + u[2][j]) * 0.5f + u[3][j];
  }
  float* dst = dw + ((int64_t)row * 9) * K + k;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    f32x4* o0 = reinterpret_cast<f32x4*>(dst + (int64_t)(i * 3 + 0) * K);
    f32x4* o1 = reinterpret_cast<f32x4*>(dst + (int64_t)(i * 3 + 1) * K);
    f32x4* o2 = reinterpret_cast<f32x4*>(dst + (int64_t)(i * 3 + 2) * K);
    *o0 = *o0 + (h[i][0] + (h[i][1] + h[i][2]) * 0.5f);
    *o1 = *o1 + (h[i][1] - h[i][2]) * 0.5f;
    *o2 = *o2 + ((h[i][1] + h[i][2]) * 0.5f + h[i][3]);
  }
}

int ssp_wino_outgrad_launch(const float* dy, int lddy, float* dM, int B, int H, int W, int C, hipStream_t stream) {
  WinoOutGradArgs a;
  a.dy = dy; a.dM = dM; a.H = H; a.W = W; a.C = C; a.lddy = lddy;
  a.th = (H + 1) / 2; a.tw = (W + 1) / 2;
  a.T = (int64_t)B * a.th * a.tw;
  SSP_CHECK_ARG(C % 4 == 0 && lddy % 4 == 0 && (((uintptr_t)dy) & 15) == 0 && (((uintptr_t)dM) & 15) == 0,
                "wino_outgrad: channels must be a multiple of 4 and the operands 16-byte aligned");
  SSP_CHECK_ARG(a.T * (C / 4) < (1ll << 31), "wino_outgrad: too many (tile, channel) pairs");
  a.div_c4 = ssp_fastdiv((unsigned)(C / 4)); a.div_tw = ssp_fastdiv((unsigned)a.tw); a.div_th = ssp_fastdiv((unsigned)a.th);
  const int64_t n = a.T * (C / 4);
  hipLaunchKernelGGL(wino_outgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
  SSP_CHECK_LAUNCH("wino_outgrad");
  return SSP_OK;
}

int ssp_wino_wgrad_finish_launch(const float* dU, float* dw, int rows, int K, hipStream_t stream) {
  const int64_t n = (int64_t)rows * (K / 4);
  hipLaunchKernelGGL(wino_wgrad_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dU, dw, rows, K);
  SSP_CHECK_LAUNCH("wino_wgrad_finish");
  return SSP_OK;
}

int ssp_wino_input_launch(const float* in, int ldin, float* V, int B, int H, int W, int C, hipStream_t stream) {
  WinoInArgs a;
  a.in = in; a.V = V; a.H = H; a.W = W; a.C = C; a.ldin = ldin;
  a.th = (H + 1) / 2; a.tw = (W + 1) / 2;
  a.T = (int64_t)B * a.th * a.tw;
  SSP_CHECK_ARG(C % 4 == 0 && ldin % 4 == 0 && (((uintptr_t)in) & 15) == 0 && (((uintptr_t)V) & 15) == 0,
                "wino_input: channels must be a multiple of 4 and the operands 16-byte aligned");
  SSP_CHECK_ARG(a.T * (C / 4) < (1ll << 31), "wino_input: too many (tile, channel) pairs");
  a.div_c4 = ssp_fastdiv((unsigned)(C / 4)); a.div_tw = ssp_fastdiv((unsigned)a.tw); a.div_th = ssp_fastdiv((unsigned)a.th);
  const int64_t n = a.T * (C / 4);
  hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
  SSP_CHECK_LAUNCH("wino_input");
  return SSP_OK;
}

int ssp_wino_filter_launch(const float* w, float* U, int rows, int K, hipStream_t stream) {
  SSP_CHECK_ARG(w != nullptr && U != nullptr && rows > 0 && K > 0 && K % 4 == 0 && (((uintptr_t)w) & 15) == 0 &&
                    (((uintptr_t)U) & 15) == 0,
                "wino_filter: [rows][9][K] filters with K % 4 == 0, 16-byte aligned operands");
  SspProfScope prof(SSP_PROF_LAYOUT, stream, 4.0 * (9.0 + 16.0) * (double)rows * K);
  const int64_t n = (int64_t)rows * (K / 4);
  hipLaunchKernelGGL(wino_filter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, U, rows, K);
  SSP_CHECK_LAUNCH("wino_filter");
  return SSP_OK;
}
