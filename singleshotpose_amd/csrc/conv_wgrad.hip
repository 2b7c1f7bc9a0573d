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
#include "conv_wino.h"

struct WgradArgs {
  const float* dy;
  const float* x;
  float* dw;
  int H, W, Cin, Cout, lddy, ldx, R, M;
  int ntile_co, ntile_ci, nsplit, chunk_m;
};

// Pipeline: as conv_igemm.hip - LDS ring of 3 slots, one barrier per staged chunk of RA pixel rows; the registers hold
// chunk it+2 (loaded during the previous chunk's MFMAs), are written to slot (it+2)%3 at the top of the iteration and
// immediately refilled with chunk it+3.  Loads are branch-free (clamped address, zeroed at LDS-store time).
template <int BMO, int BNI, int WM, int WN, int WK>
__global__ void __launch_bounds__(256, 2) conv_wgrad_kernel(WgradArgs p) {
  static_assert(WM * WN * WK == 4, "256-thread workgroups");
  constexpr int NT = 256;
  constexpr int RA = 16 * WK;  // pixel rows staged per iteration (16 per k-group of waves)
  constexpr int WTM = BMO / WM, WTN = BNI / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int F4A = RA * BMO / 4, F4B = RA * BNI / 4;
  constexpr int APASS = (F4A + NT - 1) / NT, BPASS = (F4B + NT - 1) / NT;
  constexpr int SLOT = RA * (BMO + BNI);
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of 32x32");

  extern __shared__ __attribute__((aligned(16))) float smem[];   // 3 * SLOT floats: per slot [RA][BMO] then [RA][BNI]

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
  if (niter <= 0) return;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wk = wid / (WM * WN);
  const int wmn = wid % (WM * WN);
  const int wm = wmn / WN, wn = wmn % WN;
  const int li = lane & 31, lh = lane >> 5;

  // loader constants: (row, channel) of each pass; channels beyond the tensor read channel 0 and are zeroed
  int a_row[APASS], a_col[APASS], b_row[BPASS], b_col[BPASS];
  bool a_cok[APASS], b_cok[BPASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    int idx = tid + i * NT;
    a_row[i] = idx / (BMO / 4);
    int co = co0 + (idx % (BMO / 4)) * 4;
    a_cok[i] = (idx < F4A) && (co + 3 < p.Cout || co < p.Cout);   // Cout is padded to a multiple of 4 in dy
    a_col[i] = a_cok[i] ? co : 0;
  }
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    int idx = tid + i * NT;
    b_row[i] = idx / (BNI / 4);
    int ci = ci0 + (idx % (BNI / 4)) * 4;
    b_cok[i] = (idx < F4B) && (ci < p.Cin);
    b_col[i] = b_cok[i] ? ci : 0;
  }

  f32x4 a_reg[APASS], b_reg[BPASS];
  unsigned okmask = 0;   // bits [0,APASS): dy rows valid; bits [8,8+BPASS): x rows valid
  int ld_m = m_begin;
  auto load_global = [&](f32x4 (&ar)[APASS], f32x4 (&br)[BPASS], unsigned& okm) {
    okm = 0;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      int m = ld_m + a_row[i];
      bool ok = a_cok[i] && (m < m_end);
      int mc = ok ? m : m_begin;
      ar[i] = *reinterpret_cast<const f32x4*>(p.dy + (int64_t)mc * p.lddy + a_col[i]);
      okm |= ok ? (1u << i) : 0u;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      int m = ld_m + b_row[i];
      int xx = m % p.W + dx;
      int yy = (m / p.W) % p.H + dy;
      bool ok = b_cok[i] && (m < m_end) && ((unsigned)yy < (unsigned)p.H) && ((unsigned)xx < (unsigned)p.W);
      int64_t off = ok ? ((int64_t)m * p.ldx + xshift) : ((int64_t)m_begin * p.ldx);
      br[i] = *reinterpret_cast<const f32x4*>(p.x + off + b_col[i]);
      okm |= ok ? (1u << (8 + i)) : 0u;
    }
    ld_m += RA;   // past m_end every row is masked off, so running ahead is harmless
  };
  auto store_lds = [&](int slot, const f32x4 (&ar)[APASS], const f32x4 (&br)[BPASS], unsigned okm) {
    float* As = smem + slot * SLOT;
    float* Bs = As + RA * BMO;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      int idx = tid + i * NT;
      const bool ok = (okm >> i) & 1u;
      f32x4 v;
      v[0] = ok ? ar[i][0] : 0.f; v[1] = ok ? ar[i][1] : 0.f; v[2] = ok ? ar[i][2] : 0.f; v[3] = ok ? ar[i][3] : 0.f;
      if (F4A % NT == 0 || idx < F4A) *reinterpret_cast<f32x4*>(As + idx * 4) = v;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      int idx = tid + i * NT;
      const bool ok = (okm >> (8 + i)) & 1u;
      f32x4 v;
      v[0] = ok ? br[i][0] : 0.f; v[1] = ok ? br[i][1] : 0.f; v[2] = ok ? br[i][2] : 0.f; v[3] = ok ? br[i][3] : 0.f;
      if (F4B % NT == 0 || idx < F4B) *reinterpret_cast<f32x4*>(Bs + idx * 4) = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  {
    f32x4 a0[APASS], b0[BPASS], a1[APASS], b1[BPASS];
    unsigned k0 = 0, k1 = 0;
    load_global(a0, b0, k0);
    load_global(a1, b1, k1);
    load_global(a_reg, b_reg, okmask);
    store_lds(0, a0, b0, k0);
    store_lds(1, a1, b1, k1);
  }
  __syncthreads();

  // MFMA operands of k-step kk: A[i=cout][k=pixel row 2kk+lh], B[k][j=cin]: conflict-free ds_read_b32 rows
  auto read_frag = [&](int slot, int kk, float (&av)[TM], float (&bv)[TN]) {
    const float* Ab = smem + slot * SLOT + (wk * 16 + lh + kk * 2) * BMO + wm * WTM + li;
    const float* Bb = smem + slot * SLOT + RA * BMO + (wk * 16 + lh + kk * 2) * BNI + wn * WTN + li;
#pragma unroll
    for (int i = 0; i < TM; ++i) av[i] = Ab[i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = Bb[j * 32];
  };
  float av[2][TM], bv[2][TN];
  read_frag(0, 0, av[0], bv[0]);

  int slot = 0;
  for (int it = 0; it < niter; ++it) {
    const int slot1 = (slot == 2) ? 0 : slot + 1;
    const int slot2 = (slot1 == 2) ? 0 : slot1 + 1;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      if (kk + 1 < 8) {
        read_frag(slot, kk + 1, av[(kk + 1) & 1], bv[(kk + 1) & 1]);
      } else {
        read_frag(slot1, 0, av[0], bv[0]);
      }
      if (kk == 0) {
        store_lds(slot2, a_reg, b_reg, okmask);
        load_global(a_reg, b_reg, okmask);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][i], bv[kk & 1][j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    slot = slot1;
  }

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

// ---- first-layer special case: Cin = 4 (RGB + zero pad), Cout = 32, 3x3 ------------------------------------------------
// The generic kernel would spend a 32-wide MFMA column tile on 4 input channels per tap (8x wasted issue slots and
// 3.7 ms at batch 64).  Here the nine taps are folded into the column dimension: cols = (tap, ci) = 36 -> three
// 16-wide tiles of v_mfma_f32_16x16x4_f32, rows = cout.  The layer is HBM-bound (dY alone is 1.4 GB at batch 64).
// Each of the 4 waves reduces its own 16 of the 64 staged pixels; the 4 partial 32x48 tiles are summed through LDS
// and one workgroup issues one atomic per filter element.
typedef float f32x4v __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) conv_wgrad_c4_kernel(WgradArgs p) {
  constexpr int RA = 64;        // pixels per staged chunk
  constexpr int CO = 32;        // couts
  constexpr int LSA = CO + 16;  // LDS row strides (floats): consecutive pixel rows land 16 banks apart
  constexpr int NCOL = 48;      // 36 real (tap, ci) columns padded to 3 MFMA tiles
  constexpr int LSB = NCOL;
  __shared__ __attribute__((aligned(16))) float smem[2 * RA * (LSA + LSB)];
  float* As = smem;                    // [2][RA][LSA]  dY
  float* Bs = smem + 2 * RA * LSA;     // [2][RA][LSB]  X gathered over the 9 taps

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int m_begin = blockIdx.x * p.chunk_m;
  const int m_end = min(p.M, m_begin + p.chunk_m);
  const int niter = (m_end - m_begin + RA - 1) / RA;
  if (niter <= 0) return;

  // zero the 12 padding columns once (both buffers); they are never written again
  for (int i = tid; i < 2 * RA * 12; i += 256) Bs[(i / 12) * LSB + 36 + i % 12] = 0.f;

  // loader roles: dY -> 2 float4 per thread; X -> pixel (tid & 63), taps {tid>>6, +4, +8}
  const int a_row0 = tid >> 3, a_c4 = (tid & 7) * 4;       // rows a_row0 and a_row0 + 32
  const int g_pix = tid & 63, g_tap0 = tid >> 6;
  f32x4v a_reg[2], g_reg[3];
  auto load_global = [&](int it) {
    const int mb = m_begin + it * RA;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int m = mb + a_row0 + i * 32;
      f32x4v v = {0.f, 0.f, 0.f, 0.f};
      if (m < m_end) v = *reinterpret_cast<const f32x4v*>(p.dy + (int64_t)m * p.lddy + a_c4);
      a_reg[i] = v;
    }
    const int m = mb + g_pix;
    const int x = m % p.W, y = (m / p.W) % p.H;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int tap = g_tap0 + 4 * i;
      f32x4v v = {0.f, 0.f, 0.f, 0.f};
      if (tap < 9 && m < m_end) {
        int dy = tap / 3 - 1, dx = tap % 3 - 1;
        int yy = y + dy, xx = x + dx;
        if (((unsigned)yy < (unsigned)p.H) && ((unsigned)xx < (unsigned)p.W))
          v = *reinterpret_cast<const f32x4v*>(p.x + ((int64_t)m + dy * p.W + dx) * p.ldx);
      }
      g_reg[i] = v;
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<f32x4v*>(As + (buf * RA + a_row0 + i * 32) * LSA + a_c4) = a_reg[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int tap = g_tap0 + 4 * i;
      if (tap < 9) *reinterpret_cast<f32x4v*>(Bs + (buf * RA + g_pix) * LSB + tap * 4) = g_reg[i];
    }
  };

  f32x4v acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

  load_global(0);
  store_lds(0);
  __syncthreads();
  const int li = lane & 15, lk = lane >> 4;
  for (int it = 0; it < niter; ++it) {
    const int buf = it & 1;
    if (it + 1 < niter) load_global(it + 1);
    // this wave's 16 pixels: 4 k-steps of 4 pixels; A[i=cout][k=pixel], B[k=pixel][j=col]
    const float* Ab = As + (buf * RA + wid * 16 + lk) * LSA + li;
    const float* Bb = Bs + (buf * RA + wid * 16 + lk) * LSB + li;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float av[2], bv[3];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = Ab[kk * 4 * LSA + i * 16];
#pragma unroll
      for (int j = 0; j < 3; ++j) bv[j] = Bb[kk * 4 * LSB + j * 16];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (it + 1 < niter) store_lds(buf ^ 1);
    __syncthreads();
  }

  // sum the 4 waves' partial tiles through LDS: C/D map of 16x16x4: col = lane&15, row = (lane>>4)*4 + reg
  float* red = smem;   // [4][32][48]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wid * CO + i * 16 + lk * 4 + r) * NCOL + j * 16 + li] = acc[i][j][r];
  __syncthreads();
  for (int e = tid; e < CO * 36; e += 256) {
    int co = e / 36, col = e % 36;
    float v = red[(0 * CO + co) * NCOL + col] + red[(1 * CO + co) * NCOL + col] + red[(2 * CO + co) * NCOL + col] +
              red[(3 * CO + co) * NCOL + col];
    atomicAdd(p.dw + co * 36 + col, v);   // dw[co][tap][ci], Cin = 4
  }
}

template <int BMO, int BNI, int WM, int WN, int WK>
static int launch_wgrad(WgradArgs a, hipStream_t stream) {
  constexpr int RA = 16 * WK;
  a.ntile_co = ssp_cdiv(a.Cout, BMO);
  a.ntile_ci = ssp_cdiv(a.Cin, BNI);
  const int64_t tiles = (int64_t)a.ntile_co * a.ntile_ci * a.R * a.R;
  const int lds_bytes = 3 * RA * (BMO + BNI) * 4;
  auto kern = conv_wgrad_kernel<BMO, BNI, WM, WN, WK>;
  static SspKernelCache cache;   // per instantiation, per device
  int slots = 0;                 // workgroups resident on the whole chip (occupancy x CUs)
  if (int rc = ssp_kernel_prepare((const void*)kern, lds_bytes, 256, &cache, &slots, "conv_wgrad")) return rc;
  // Split the pixel reduction so that the grid is (close to) a whole number of resident waves of workgroups - a
  // 2.25-wave grid runs as long as a 3-wave one - with 2..5 waves in total and >= 8 staged chunks per workgroup.
  const int64_t max_split = (a.M + RA * 8 - 1) / (RA * 8);
  int64_t lo = (2 * (int64_t)slots + tiles - 1) / tiles, hi = (5 * (int64_t)slots) / tiles;
  if (lo < 1) lo = 1;
  if (hi < lo) hi = lo;
  if (lo > max_split) lo = max_split;
  if (hi > max_split) hi = max_split;
  int64_t nsplit = lo;
  double best = -1.0;
  for (int64_t sp = lo; sp <= hi; ++sp) {
    const double waves = (double)(tiles * sp) / slots;
    const double eff = waves / (double)((tiles * sp + slots - 1) / slots);
    if (eff > best + 1e-3) { best = eff; nsplit = sp; }
  }
  int64_t chunk = (a.M + nsplit - 1) / nsplit;
  chunk = (chunk + RA - 1) / RA * RA;
  nsplit = (a.M + chunk - 1) / chunk;
  a.nsplit = (int)nsplit;
  a.chunk_m = (int)chunk;
  dim3 grid((unsigned)(tiles * nsplit)), block(256);
  hipLaunchKernelGGL(kern, grid, block, lds_bytes, stream, a);
  SSP_CHECK_LAUNCH("conv_wgrad");
  return SSP_OK;
}

int ssp_conv_wgrad_dma_try(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int lddy,
                           int ldx, int R, hipStream_t stream, int batch = 0, int64_t batch_dy = 0, int64_t batch_x = 0,
                           int64_t batch_dw = 0, int overwrite = 0);   // conv_wgrad_dma.hip

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
  if (ssp_option(SSP_OPT_WGRAD_VARIANT) != 2) {   // LDS-direct loader for tiles with >= 64 couts and >= 64 cins
    const int r = ssp_conv_wgrad_dma_try(dy, x, dw, B, H, W, Cin, Cout, lddy, ldx, R, stream);
    if (r != 0) return r < 0 ? r : SSP_OK;
  }
  if (Cin == 4 && Cout == 32 && R == 3 && ldx == 4 && ssp_option(SSP_OPT_WGRAD_VARIANT) != 1) {
    // ~8 workgroups per CU, each a multiple of 64 pixels
    int64_t chunk = (a.M + 2047) / 2048;
    chunk = (chunk + 63) / 64 * 64;
    a.chunk_m = (int)chunk;
    a.nsplit = ssp_cdiv(a.M, chunk);
    hipLaunchKernelGGL(conv_wgrad_c4_kernel, dim3(a.nsplit), dim3(256), 0, stream, a);
    SSP_CHECK_LAUNCH("conv_wgrad_c4");
    return SSP_OK;
  }
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


// Filter gradient of a 3x3 layer in the Winograd domain (conv_wino.hip): transform the input (B^T d B) and the output
// gradient (A dY A^T) into P = (tile + 2)^2 planes each, contract the planes pairwise over the tiles in ONE batched launch
// of the LDS-direct filter-gradient kernel, map the P results back onto the 9 taps (G^T . G) and add them to dw.
// workspace: V [P][T][Cin] | dM [P][T][Cout] | dU [P][Cout][Cin] (ssp_conv_wgrad_wino_ws_floats); V sits where a
// Winograd forward launch of the same tile size puts it (workspace head), so a caller that hands both launches the same
// buffer transforms the layer input once per step.  dU needs no clearing by the caller: an un-split batched launch writes
// it with plain stores, a split one zeroes it first.
int64_t ssp_conv_wgrad_wino_ws_floats(int B, int H, int W, int Cin, int Cout, int tile) {
  if (tile == SSP_WINO_WGRAD_FUSED) return ssp_wino_wgrad_fused_ws_floats(Cin, Cout);      // dU alone: V and dM stay on the chip
  if (tile != 2 && tile != 4) return 0;
  const int64_t T = ssp_wino_tiles(B, H, W, tile);
  return ssp_wino_planes(tile) * (T * ((int64_t)Cin + Cout) + (int64_t)Cin * Cout);
}
int ssp_conv_wgrad_wino_launch(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int lddy,
                               int ldx, int tile, float* ws, int64_t ws_floats, hipStream_t stream) {
  if (tile == SSP_WINO_WGRAD_FUSED) return ssp_wino_wgrad_fused_launch(dy, x, dw, B, H, W, Cin, Cout, lddy, ldx, ws, ws_floats, stream);
  SSP_CHECK_ARG(tile == 2 || tile == 4, "wgrad (Winograd): tile must be 2, 4 or 12 (F(2x2) with both transforms on the chip)");
  SSP_CHECK_ARG(Cin % 16 == 0 && Cout % 16 == 0 && Cin >= 64 && Cout >= 64, "wgrad (Winograd): needs Cin, Cout >= 64 and %% 16 == 0");
  SSP_CHECK_ARG(ldx % 4 == 0 && ldx >= Cin && lddy % 4 == 0 && lddy >= Cout, "wgrad (Winograd): bad leading dimensions");
  SSP_CHECK_ARG((((uintptr_t)dy) | ((uintptr_t)x) | ((uintptr_t)dw) | ((uintptr_t)ws)) % 16 == 0, "wgrad (Winograd): operands must be 16-byte aligned");
  const int64_t T = ssp_wino_tiles(B, H, W, tile);
  const int P = ssp_wino_planes(tile);
  SSP_CHECK_ARG(T >= 16 && T < (1ll << 31), "wgrad (Winograd): tile count out of range (>= 16 tiles)");
  SSP_CHECK_ARG(ws != nullptr && ws_floats >= ssp_conv_wgrad_wino_ws_floats(B, H, W, Cin, Cout, tile),
                "wgrad (Winograd): needs a workspace of %lld floats (ssp_conv_wgrad_wino_workspace_floats)",
                (long long)ssp_conv_wgrad_wino_ws_floats(B, H, W, Cin, Cout, tile));
  SspProfScope prof(SSP_PROF_CONV_WGRAD, stream, 2.0 * (double)B * H * W * Cout * 9.0 * Cin);      // algorithmic (direct) FLOPs
  float* V = ws;
  float* dM = V + P * T * Cin;
  float* dU = dM + P * T * Cout;
  // x == nullptr: the transformed input is already at the head of the workspace - the forward launch of the same layer
  // (a Winograd plan of ssp_conv_fwd with the same tile size, given THIS workspace) left V there, and nothing has written
  // the region since
  if (x != nullptr)
    if (int rc = ssp_wino_input_launch(x, ldx, V, B, H, W, Cin, tile, SSP_PROF_WINO_WGRAD, stream)) return rc;
  if (int rc = ssp_wino_outgrad_launch(dy, lddy, dM, B, H, W, Cout, tile, stream)) return rc;
  const int r = ssp_conv_wgrad_dma_try(dM, V, dU, 1, 1, (int)T, Cin, Cout, Cout, Cin, 1, stream, P, T * Cout, T * Cin,
                                       (int64_t)Cout * Cin, 1);
  if (r < 0) return r;
  SSP_CHECK_ARG(r == 1, "wgrad (Winograd): shape declined by the LDS-direct filter-gradient kernel");
  return ssp_wino_wgrad_finish_launch(dU, dw, Cout, Cin, tile, stream);
}
