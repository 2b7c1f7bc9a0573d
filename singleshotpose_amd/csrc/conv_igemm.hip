// Implicit-GEMM convolution (stride 1, "same" padding, R in {1,3}) on the CDNA4 fp32 matrix cores.
//
// Replaces the cuDNN/ATen kernels behind nn.Conv2d in the reference's conv blocks
// (/root/reference/darknet.py:145-167) for BOTH the forward pass and the data-gradient pass:
// dgrad is the same contraction run on dY with spatially flipped, in/out-transposed filters
// (the repack lives in layout.hip).
//
//   out[m][n] = sum_k A[m][k] * Wt[n][k]      m = (b,y,x) pixel, n = cout, k = (tap, cin)
//   A[m][(tap,c)] = in[b][y + tap/R - R/2][x + tap%R - R/2][c]   (zero outside the image)
//
// Data layout: activations NHWC fp32 with an explicit per-pixel stride (so a conv can read or
// write a channel slice of a wider buffer, which is how route/concat is realised without a copy),
// filters [Cout][R*R][Cin] (k contiguous) - the classic "TN" GEMM where both operands are
// k-contiguous, so global->LDS staging is 16-byte coalesced on both sides.
//
// Matrix core use: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = 157.3 TF chip peak).
// A-operand lane l holds A[i=l&31][k=l>>5], B-operand lane l holds B[k=l>>5][j=l&31]
// (cdna_hip_programming.md section 3).  The k order inside a BK chunk is arbitrary as long as A and B
// agree, so lane (i,h) reads VEC consecutive floats at column h*VEC of its LDS row with one
// ds_read_b128/b64 and feeds them to VEC successive MFMAs.
#include "ssp_common.h"

struct ConvArgs {
  const float* in;
  const float* wt;
  float* out;
  const float* bias;  // [Cout] or nullptr (added in the epilogue; the linear head conv)
  float* stats;       // [ntile_m][Cout][2] = per-M-tile (mean, M2) of the raw output, or nullptr
  int H, W, Cin, Cout, ldin, ldout, R, M;
  int accumulate;     // out += result (second consumer of a routed activation in dgrad)
  int ntile_m, ntile_n;
};

__device__ __forceinline__ void chan_combine(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
  float nt = n + nb;
  if (nt > 0.f) {
    float d = mb - mean;
    float f = nb / nt;
    mean += d * f;
    m2 += m2b + d * d * n * f;
    n = nt;
  }
}

template <int BM, int BN, int WM, int WN, int BK>
__global__ void __launch_bounds__(WM* WN * 64) conv_igemm_kernel(ConvArgs p) {
  constexpr int NT = WM * WN * 64;
  constexpr int LS = BK + 4;        // LDS row stride in floats (16-B aligned, conflict-free b128 reads)
  constexpr int TPR = BK / 4;       // loader threads per tile row (one float4 each)
  constexpr int RPP = NT / TPR;     // tile rows covered per loader pass
  constexpr int APASS = (BM + RPP - 1) / RPP;
  constexpr int BPASS = (BN + RPP - 1) / RPP;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int VEC = (BK >= 8) ? 4 : 2;
  constexpr int NQ = BK / (2 * VEC);
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA");
  static_assert(NQ >= 1, "BK too small");

  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LS];
  float* As = smem;
  float* Bs = smem + 2 * BM * LS;

  const int nwg = p.ntile_m * p.ntile_n;
  const int lid = ssp_xcd_remap(blockIdx.x, nwg);
  const int tile_n = lid % p.ntile_n, tile_m = lid / p.ntile_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int li = lane & 31, lh = lane >> 5;

  const int pad = p.R >> 1;
  const int K = p.R * p.R * p.Cin;
  const int cpt = p.Cin / BK;  // K chunks per filter tap
  const int niter = p.R * p.R * cpt;

  // ---- loader setup: rows are fixed for the whole K loop ----
  const int lrow = tid / TPR, lcol = (tid % TPR) * 4;
  const float* a_ptr[APASS];
  int a_y[APASS], a_x[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    int row = lrow + i * RPP;
    int m = m0 + row;
    if (row < BM && m < p.M) {
      int x = m % p.W;
      int t = m / p.W;
      a_y[i] = t % p.H;
      a_x[i] = x;
      a_ptr[i] = p.in + (int64_t)m * p.ldin + lcol;
    } else {
      a_y[i] = -(1 << 20);
      a_x[i] = 0;
      a_ptr[i] = p.in;
    }
  }
  const float* b_ptr[BPASS];
  bool b_ok[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    int row = lrow + i * RPP;
    int n = n0 + row;
    b_ok[i] = (row < BN) && (n < p.Cout);
    b_ptr[i] = b_ok[i] ? (p.wt + (int64_t)n * K + lcol) : p.wt;
  }

  f32x4 a_reg[APASS], b_reg[BPASS];
  auto load_global = [&](int it) {
    int tap = it / cpt;
    int c0 = (it - tap * cpt) * BK;
    int dy = tap / p.R - pad, dx = tap % p.R - pad;
    int64_t shift = ((int64_t)dy * p.W + dx) * p.ldin + c0;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      int yy = a_y[i] + dy, xx = a_x[i] + dx;
      bool ok = ((unsigned)yy < (unsigned)p.H) && ((unsigned)xx < (unsigned)p.W);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(a_ptr[i] + shift);
      a_reg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (b_ok[i]) v = *reinterpret_cast<const f32x4*>(b_ptr[i] + (int64_t)it * BK);
      b_reg[i] = v;
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      int row = lrow + i * RPP;
      if (row < BM) *reinterpret_cast<f32x4*>(As + (buf * BM + row) * LS + lcol) = a_reg[i];
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      int row = lrow + i * RPP;
      if (row < BN) *reinterpret_cast<f32x4*>(Bs + (buf * BN + row) * LS + lcol) = b_reg[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_global(0);
  store_lds(0);
  __syncthreads();

  for (int it = 0; it < niter; ++it) {
    const int buf = it & 1;
    if (it + 1 < niter) load_global(it + 1);  // in flight while the matrix cores work on `buf`

    const float* Ab = As + (buf * BM + wm * WTM + li) * LS + lh * VEC;
    const float* Bb = Bs + (buf * BN + wn * WTN + li) * LS + lh * VEC;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      float av[TM][VEC], bv[TN][VEC];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (VEC == 4) {
          f32x4 t = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LS + q * 2 * VEC);
          av[i][0] = t[0]; av[i][1] = t[1]; av[i][2] = t[2]; av[i][3] = t[3];
        } else {
          f32x2 t = *reinterpret_cast<const f32x2*>(Ab + i * 32 * LS + q * 2 * VEC);
          av[i][0] = t[0]; av[i][1] = t[1];
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (VEC == 4) {
          f32x4 t = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LS + q * 2 * VEC);
          bv[j][0] = t[0]; bv[j][1] = t[1]; bv[j][2] = t[2]; bv[j][3] = t[3];
        } else {
          f32x2 t = *reinterpret_cast<const f32x2*>(Bb + j * 32 * LS + q * 2 * VEC);
          bv[j][0] = t[0]; bv[j][1] = t[1];
        }
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
    }

    if (it + 1 < niter) store_lds(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WTN + j * 32 + li;
    const bool n_ok = n < p.Cout;
    const float bias = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
    float cnt = 0.f, sum = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        int m = m0 + row;
        float v = acc[i][j][r] + bias;
        if (m < p.M && n_ok) {
          float* o = p.out + (int64_t)m * p.ldout + n;
          if (p.accumulate) v += *o;
          *o = v;
          cnt += 1.f;
          sum += acc[i][j][r];
        }
      }
    }
    if (p.stats != nullptr) {
      // per-lane (count, mean, M2) of this lane's column over its valid rows, then Chan-combine:
      // lane halves (rows +4) -> waves along M (through LDS) -> one (mean, M2) pair per column per M tile.
      float mean = cnt > 0.f ? sum / cnt : 0.f;
      float m2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m0 + row < p.M && n_ok) {
            float d = acc[i][j][r] - mean;
            m2 += d * d;
          }
        }
      }
      float ocnt = __shfl_xor(cnt, 32), omean = __shfl_xor(mean, 32), om2 = __shfl_xor(m2, 32);
      chan_combine(cnt, mean, m2, ocnt, omean, om2);
      // smem is free again: the K loop ended with a barrier
      float* red = smem;  // [WM][BN][3]
      int col = wn * WTN + j * 32 + li;
      if (lh == 0) {
        red[(wm * BN + col) * 3 + 0] = cnt;
        red[(wm * BN + col) * 3 + 1] = mean;
        red[(wm * BN + col) * 3 + 2] = m2;
      }
    }
  }
  if (p.stats != nullptr) {
    __syncthreads();
    for (int col = tid; col < BN; col += NT) {
      int n = n0 + col;
      if (n < p.Cout) {
        float cnt = smem[col * 3 + 0], mean = smem[col * 3 + 1], m2 = smem[col * 3 + 2];
#pragma unroll
        for (int w = 1; w < WM; ++w)
          chan_combine(cnt, mean, m2, smem[(w * BN + col) * 3 + 0], smem[(w * BN + col) * 3 + 1],
                       smem[(w * BN + col) * 3 + 2]);
        float* s = p.stats + ((int64_t)tile_m * p.Cout + n) * 2;
        s[0] = mean;
        s[1] = m2;
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int BK>
static int launch_cfg(ConvArgs a, hipStream_t stream) {
  a.ntile_m = ssp_cdiv(a.M, BM);
  a.ntile_n = ssp_cdiv(a.Cout, BN);
  dim3 grid(a.ntile_m * a.ntile_n), block(WM * WN * 64);
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, BK>), grid, block, 0, stream, a);
  SSP_CHECK_LAUNCH("conv_igemm");
  return SSP_OK;
}

// Rows of the per-M-tile statistics buffer a forward launch will write (host sizes the workspace).
int ssp_conv_tile_m(int Cout) { return Cout > 64 ? 128 : 256; }

int ssp_conv_igemm_launch(const float* in, const float* wt, float* out, const float* bias, float* stats,
                          int B, int H, int W, int Cin, int Cout, int ldin, int ldout, int R, int accumulate,
                          int prof_kind, hipStream_t stream) {
  SSP_CHECK_ARG(R == 1 || R == 3, "conv: only 1x1 and 3x3 filters are supported (got %d)", R);
  SSP_CHECK_ARG(Cin % 4 == 0 && Cin > 0, "conv: Cin must be a positive multiple of 4 (got %d)", Cin);
  SSP_CHECK_ARG(ldin % 4 == 0 && ldin >= Cin, "conv: ldin must be a multiple of 4 and >= Cin");
  SSP_CHECK_ARG(ldout >= Cout && Cout > 0, "conv: ldout < Cout");
  SSP_CHECK_ARG((int64_t)B * H * W < (1ll << 31), "conv: too many pixels");
  SSP_CHECK_ARG((((uintptr_t)in) & 15) == 0 && (((uintptr_t)wt) & 15) == 0, "conv: in/wt must be 16-byte aligned");
  ConvArgs a;
  a.in = in; a.wt = wt; a.out = out; a.bias = bias; a.stats = stats;
  a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ldin = ldin; a.ldout = ldout; a.R = R;
  a.M = B * H * W; a.accumulate = accumulate;
  SspProfScope prof(prof_kind, stream, 2.0 * (double)a.M * Cout * (double)(R * R * Cin));
  const bool k16 = (Cin % 16) == 0;
  if (Cout > 64) {
    return k16 ? launch_cfg<128, 128, 2, 2, 16>(a, stream) : launch_cfg<128, 128, 2, 2, 4>(a, stream);
  } else if (Cout > 32) {
    return k16 ? launch_cfg<256, 64, 4, 1, 16>(a, stream) : launch_cfg<256, 64, 4, 1, 4>(a, stream);
  } else {
    return k16 ? launch_cfg<256, 32, 4, 1, 16>(a, stream) : launch_cfg<256, 32, 4, 1, 4>(a, stream);
  }
}
