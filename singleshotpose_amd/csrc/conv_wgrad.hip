// Filter-gradient of the stride-1 "same" convolution on the CDNA4 fp32 matrix cores.
//
// Replaces the autograd wgrad that loss.backward() runs for every nn.Conv2d of the reference
// (/root/reference/train.py:103 through /root/reference/darknet.py:156,160).
//
//   dW[co][tap][ci] = sum_m dY[m][co] * X[m shifted by tap][ci]      m = (b,y,x)
//
// GEMM view per filter tap: rows = cout, cols = cin, reduction = pixels.  Both operands are
// channel-contiguous per pixel (NHWC), so a [pixels][channels] LDS image is filled with coalesced
// float4 loads and the MFMA operands (lane l: A[i=l&31][k=l>>5]) are conflict-free ds_read_b32
// of 32 consecutive channels of pixel row 2*kk + (l>>5).
//
// The pixel reduction is split across workgroups (grid.x = tiles * taps * nsplit) and - for
// thin layers - across the waves of a workgroup (WK); partial tiles are summed into dW with
// hardware fp32 atomics, so the caller zeroes dW first.
#include "ssp_common.h"

struct WgradArgs {
  const float* dy;
  const float* x;
  float* dw;
  int H, W, Cin, Cout, lddy, ldx, R, M;
  int ntile_co, ntile_ci, nsplit, chunk_m;
};

template <int BMO, int BNI, int WM, int WN, int WK>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(WgradArgs p) {
  static_assert(WM * WN * WK == 4, "256-thread workgroups");
  constexpr int NT = 256;
  constexpr int RA = 16 * WK;  // pixel rows staged per iteration (16 per k-group of waves)
  constexpr int WTM = BMO / WM, WTN = BNI / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int F4A = RA * BMO / 4, F4B = RA * BNI / 4;
  constexpr int APASS = (F4A + NT - 1) / NT, BPASS = (F4B + NT - 1) / NT;
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of 32x32");

  __shared__ __attribute__((aligned(16))) float smem[2 * RA * (BMO + BNI)];
  float* As = smem;                  // [2][RA][BMO]
  float* Bs = smem + 2 * RA * BMO;   // [2][RA][BNI]

  const int taps = p.R * p.R;
  int bid = blockIdx.x;
  const int split = bid % p.nsplit; bid /= p.nsplit;
  const int tap = bid % taps; bid /= taps;
  const int tile_ci = bid % p.ntile_ci;
  const int tile_co = bid / p.ntile_ci;
  const int co0 = tile_co * BMO, ci0 = tile_ci * BNI;
  const int pad = p.R >> 1;
  const int dy = tap / p.R - pad, dx = tap % p.R - pad;
  const int64_t xshift = ((int64_t)dy * p.W + dx) * p.ldx;

  const int m_begin = split * p.chunk_m;
  const int m_end = min(p.M, m_begin + p.chunk_m);
  const int niter = (m_end - m_begin + RA - 1) / RA;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wk = wid / (WM * WN);
  const int wmn = wid % (WM * WN);
  const int wm = wmn / WN, wn = wmn % WN;
  const int li = lane & 31, lh = lane >> 5;

  f32x4 a_reg[APASS], b_reg[BPASS];
  auto load_global = [&](int it) {
    const int mb = m_begin + it * RA;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      int idx = tid + i * NT;
      int row = idx / (BMO / 4), c4 = (idx % (BMO / 4)) * 4;
      int m = mb + row;
      int co = co0 + c4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < F4A && m < m_end && co < p.Cout) {
        const float* src = p.dy + (int64_t)m * p.lddy + co;
        if (co + 3 < p.Cout) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {  // ragged channel tail (the 20-channel head)
          v[0] = src[0];
          if (co + 1 < p.Cout) v[1] = src[1];
          if (co + 2 < p.Cout) v[2] = src[2];
        }
      }
      a_reg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      int idx = tid + i * NT;
      int row = idx / (BNI / 4), c4 = (idx % (BNI / 4)) * 4;
      int m = mb + row;
      int ci = ci0 + c4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < F4B && m < m_end && ci < p.Cin) {  // Cin is a multiple of 4
        int xx = m % p.W + dx;
        int yy = (m / p.W) % p.H + dy;
        if (((unsigned)yy < (unsigned)p.H) && ((unsigned)xx < (unsigned)p.W))
          v = *reinterpret_cast<const f32x4*>(p.x + (int64_t)m * p.ldx + xshift + ci);
      }
      b_reg[i] = v;
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      int idx = tid + i * NT;
      if (idx < F4A) *reinterpret_cast<f32x4*>(As + buf * RA * BMO + idx * 4) = a_reg[i];
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      int idx = tid + i * NT;
      if (idx < F4B) *reinterpret_cast<f32x4*>(Bs + buf * RA * BNI + idx * 4) = b_reg[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (niter > 0) {
    load_global(0);
    store_lds(0);
  }
  __syncthreads();

  for (int it = 0; it < niter; ++it) {
    const int buf = it & 1;
    if (it + 1 < niter) load_global(it + 1);
    const float* Ab = As + (buf * RA + wk * 16 + lh) * BMO + wm * WTM + li;
    const float* Bb = Bs + (buf * RA + wk * 16 + lh) * BNI + wn * WTN + li;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = Ab[kk * 2 * BMO + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = Bb[kk * 2 * BNI + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (it + 1 < niter) store_lds(buf ^ 1);
    __syncthreads();
  }

  if (niter <= 0) return;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      int ci = ci0 + wn * WTN + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int co = co0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < p.Cout && ci < p.Cin)
          atomicAdd(p.dw + ((int64_t)co * taps + tap) * p.Cin + ci, acc[i][j][r]);
      }
    }
}

template <int BMO, int BNI, int WM, int WN, int WK>
static int launch_wgrad(WgradArgs a, hipStream_t stream) {
  constexpr int RA = 16 * WK;
  a.ntile_co = ssp_cdiv(a.Cout, BMO);
  a.ntile_ci = ssp_cdiv(a.Cin, BNI);
  const int64_t tiles = (int64_t)a.ntile_co * a.ntile_ci * a.R * a.R;
  // enough workgroups for ~6 per CU, but keep >= 8 staging iterations per workgroup
  int64_t want = (1536 + tiles - 1) / tiles;
  int64_t max_split = (a.M + RA * 8 - 1) / (RA * 8);
  int64_t nsplit = want < 1 ? 1 : (want > max_split ? max_split : want);
  if (nsplit < 1) nsplit = 1;
  int64_t chunk = (a.M + nsplit - 1) / nsplit;
  chunk = (chunk + RA - 1) / RA * RA;
  nsplit = (a.M + chunk - 1) / chunk;
  a.nsplit = (int)nsplit;
  a.chunk_m = (int)chunk;
  dim3 grid((unsigned)(tiles * nsplit)), block(256);
  hipLaunchKernelGGL((conv_wgrad_kernel<BMO, BNI, WM, WN, WK>), grid, block, 0, stream, a);
  SSP_CHECK_LAUNCH("conv_wgrad");
  return SSP_OK;
}

int ssp_conv_wgrad_launch(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout,
                          int lddy, int ldx, int R, hipStream_t stream) {
  SSP_CHECK_ARG(R == 1 || R == 3, "wgrad: only 1x1 and 3x3 filters are supported (got %d)", R);
  SSP_CHECK_ARG(Cin % 4 == 0 && Cin > 0, "wgrad: Cin must be a positive multiple of 4 (got %d)", Cin);
  SSP_CHECK_ARG(ldx % 4 == 0 && ldx >= Cin, "wgrad: ldx must be a multiple of 4 and >= Cin");
  SSP_CHECK_ARG(lddy % 4 == 0 && lddy >= Cout, "wgrad: lddy must be a multiple of 4 and >= Cout");
  SSP_CHECK_ARG((int64_t)B * H * W < (1ll << 31), "wgrad: too many pixels");
  SSP_CHECK_ARG((((uintptr_t)dy) & 15) == 0 && (((uintptr_t)x) & 15) == 0, "wgrad: dy/x must be 16-byte aligned");
  WgradArgs a;
  a.dy = dy; a.x = x; a.dw = dw;
  a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.lddy = lddy; a.ldx = ldx; a.R = R; a.M = B * H * W;
  SspProfScope prof(SSP_PROF_CONV_WGRAD, stream, 2.0 * (double)a.M * Cout * (double)(R * R * Cin));
  const int bo = Cout >= 128 ? 128 : (Cout >= 64 ? 64 : 32);
  const int bi = Cin >= 128 ? 128 : (Cin >= 64 ? 64 : 32);
  if (bo == 128 && bi == 128) return launch_wgrad<128, 128, 2, 2, 1>(a, stream);
  if (bo == 128 && bi == 64) return launch_wgrad<128, 64, 2, 2, 1>(a, stream);
  if (bo == 128 && bi == 32) return launch_wgrad<128, 32, 4, 1, 1>(a, stream);
  if (bo == 64 && bi == 128) return launch_wgrad<64, 128, 2, 2, 1>(a, stream);
  if (bo == 64 && bi == 64) return launch_wgrad<64, 64, 2, 2, 1>(a, stream);
  if (bo == 64 && bi == 32) return launch_wgrad<64, 32, 2, 1, 2>(a, stream);
  if (bo == 32 && bi == 128) return launch_wgrad<32, 128, 1, 4, 1>(a, stream);
  if (bo == 32 && bi == 64) return launch_wgrad<32, 64, 1, 2, 2>(a, stream);
  return launch_wgrad<32, 32, 1, 1, 4>(a, stream);
}
