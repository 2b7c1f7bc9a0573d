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
#include "conv_igemm_common.h"
#include "conv_wino.h"

// ABL: 0 = product kernel; ablation bits for tools/conv_bench.py only (results are garbage): 1 no global loads in
// the K loop, 2 no MFMA, 4 no barrier, 8 no fragment reads, 16 no LDS stores
//
// Pipeline (per K chunk `it`, LDS ring of 3 slots, one barrier per chunk):
//   registers (chunk it+2, loaded one chunk ago) -> LDS slot (it+2)%3 | global loads of chunk it+3 -> registers
//   | MFMAs on chunk it (its first fragments were read during chunk it-1)
//   | fragment reads of chunk it+1 (written + barrier-published one chunk ago) | barrier
// so after a barrier the matrix cores restart from registers, and neither the LDS round trip nor the HBM/L2 latency of
// a chunk sits between two MFMA blocks.  Loads are branch-free: out-of-image taps read a clamped address and are
// zeroed with v_cndmask; rows/columns beyond M/Cout read row 0 / the last filter and are never stored.
// PASS (0 forward, 1 data gradient) only gives the two uses distinct kernel names for profiles (as conv_igemm_dma.hip)
template <int BM, int BN, int WM, int WN, int BK, int ABL = 0, int PASS = 0>
__global__ void __launch_bounds__(WM* WN * 64, 2) conv_igemm_kernel(ConvArgs p) {
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
  constexpr int SLOT = (BM + BN) * LS;   // floats per ring slot: A rows then B rows
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA");
  static_assert(NQ >= 1, "BK too small");

  extern __shared__ __attribute__((aligned(16))) float smem[];   // 3 * SLOT floats

  const int ntiles = p.ntile_m * p.ntile_n;
  const int nwg = ntiles * p.ksplit;
  const int lid0 = p.xcd_remap ? ssp_xcd_remap(blockIdx.x, nwg) : (int)blockIdx.x;
  const int split = lid0 / ntiles, lid = lid0 - split * ntiles;
  const int tile_n = lid % p.ntile_n, tile_m = lid / p.ntile_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int li = lane & 31, lh = lane >> 5;

  const int pad = p.R >> 1;
  const int K = p.R * p.R * p.Cin;
  const int cpt = p.Cin / BK;  // K chunks per filter tap
  const int it_begin = split * p.it_per_split;
  const int niter = min(p.R * p.R * cpt, it_begin + p.it_per_split) - it_begin;   // this workgroup's K chunks

  // ---- loader setup: rows are fixed for the whole K loop ----
  const int lrow = tid / TPR, lcol = (tid % TPR) * 4;
  const float* a_ptr[APASS];
  int a_y[APASS], a_x[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    int row = lrow + i * RPP;
    int m = m0 + row;
    if (m >= p.M) m = p.M - 1;   // clamped duplicate row: computed, never stored
    int x = m % p.W;
    int t = m / p.W;
    a_y[i] = t % p.H;
    a_x[i] = x;
    a_ptr[i] = p.in + (int64_t)m * p.ldin + lcol;
  }
  const float* b_ptr[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    int n = n0 + lrow + i * RPP;
    if (n >= p.Cout) n = p.Cout - 1;
    b_ptr[i] = p.wt + (int64_t)n * K + lcol;
  }

  f32x4 a_reg[APASS], b_reg[BPASS];
  unsigned a_okmask = 0;   // bit i: pass i of the staged chunk is inside the image (applied at LDS-store time, so
                           // the zeroing select does not pull the s_waitcnt for the loads in front of the MFMAs)
  // K-chunk walker for the loader (chunks are visited in order; no per-chunk integer division)
  const int tap0 = it_begin / cpt;
  int ld_c0 = (it_begin - tap0 * cpt) * BK, ld_dy = tap0 / p.R - pad, ld_dx = tap0 % p.R - pad;
  int64_t ld_koff = (int64_t)it_begin * BK;   // it * BK
  auto load_global = [&](f32x4 (&a_reg)[APASS], f32x4 (&b_reg)[BPASS], unsigned& a_okmask) {
    const int64_t shift = ((int64_t)ld_dy * p.W + ld_dx) * p.ldin + ld_c0;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      int yy = a_y[i] + ld_dy, xx = a_x[i] + ld_dx;
      bool ok = ((unsigned)yy < (unsigned)p.H) && ((unsigned)xx < (unsigned)p.W);
      a_reg[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + (ok ? shift : (int64_t)ld_c0));
      a_okmask = ok ? (a_okmask | (1u << i)) : (a_okmask & ~(1u << i));
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) b_reg[i] = *reinterpret_cast<const f32x4*>(b_ptr[i] + ld_koff);
    // advance; past the last chunk the walker re-reads the last filter columns / an out-of-window tap (harmless:
    // those stagings land in a ring slot nobody reads), which keeps the K loop free of branches
    ld_koff = min(ld_koff + BK, (int64_t)(K - BK));
    ld_c0 += BK;
    const bool wrap = ld_c0 == p.Cin;
    ld_c0 = wrap ? 0 : ld_c0;
    ld_dx += wrap ? 1 : 0;
    const bool wrap2 = ld_dx > pad;
    ld_dx = wrap2 ? -pad : ld_dx;
    ld_dy += wrap2 ? 1 : 0;
  };
  auto store_lds = [&](int slot, const f32x4 (&a_reg)[APASS], const f32x4 (&b_reg)[BPASS], unsigned a_okmask) {
    float* As = smem + slot * SLOT;
    float* Bs = As + BM * LS;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      int row = lrow + i * RPP;
      const bool ok = (a_okmask >> i) & 1u;
      f32x4 v;
      v[0] = ok ? a_reg[i][0] : 0.f; v[1] = ok ? a_reg[i][1] : 0.f;
      v[2] = ok ? a_reg[i][2] : 0.f; v[3] = ok ? a_reg[i][3] : 0.f;
      if (BM % RPP == 0 || row < BM) *reinterpret_cast<f32x4*>(As + row * LS + lcol) = v;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      int row = lrow + i * RPP;
      if (BN % RPP == 0 || row < BN) *reinterpret_cast<f32x4*>(Bs + row * LS + lcol) = b_reg[i];
    }
  };
  // fragment read: lane (i,h) takes VEC consecutive floats at column q*2*VEC + h*VEC of its row
  auto read_frag = [&](int slot, int q, float (&fa)[TM][VEC], float (&fb)[TN][VEC]) {
    const float* Ab = smem + slot * SLOT + (wm * WTM + li) * LS + lh * VEC + q * 2 * VEC;
    const float* Bb = smem + slot * SLOT + (BM + wn * WTN + li) * LS + lh * VEC + q * 2 * VEC;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (VEC == 4) {
        f32x4 t = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LS);
        fa[i][0] = t[0]; fa[i][1] = t[1]; fa[i][2] = t[2]; fa[i][3] = t[3];
      } else {
        f32x2 t = *reinterpret_cast<const f32x2*>(Ab + i * 32 * LS);
        fa[i][0] = t[0]; fa[i][1] = t[1];
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (VEC == 4) {
        f32x4 t = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LS);
        fb[j][0] = t[0]; fb[j][1] = t[1]; fb[j][2] = t[2]; fb[j][3] = t[3];
      } else {
        f32x2 t = *reinterpret_cast<const f32x2*>(Bb + j * 32 * LS);
        fb[j][0] = t[0]; fb[j][1] = t[1];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto mma = [&](float (&fa)[TM][VEC], float (&fb)[TN][VEC]) {
#pragma unroll
    for (int e = 0; e < VEC; ++e)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr ((ABL & 2) != 0) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::"v"(fa[i][e]), "v"(fb[j][e]));
#endif
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
          }
        }
  };

  // ---- prologue: chunks 0 and 1 into slots 0 and 1, first fragments into registers ----
  {
    // all three stagings are issued back to back (one memory round trip, not three); past the last chunk the walker
    // re-reads in-bounds data that is never consumed
    f32x4 a0[APASS], b0[BPASS], a1[APASS], b1[BPASS];
    unsigned m0k = 0, m1k = 0;
    load_global(a0, b0, m0k);
    load_global(a1, b1, m1k);
    load_global(a_reg, b_reg, a_okmask);   // chunk 2 stays in registers until the first loop iteration stores it
    store_lds(0, a0, b0, m0k);
    store_lds(1, a1, b1, m1k);
  }
  __syncthreads();
  float fa[2][TM][VEC], fb[2][TN][VEC];   // fragment double buffer; set (q & 1) holds sub-step q
  read_frag(0, 0, fa[0], fb[0]);

  int slot = 0;   // it % 3
  for (int it = 0; it < niter; ++it) {
    const int slot1 = (slot == 2) ? 0 : slot + 1;          // (it+1) % 3
    const int slot2 = (slot1 == 2) ? 0 : slot1 + 1;        // (it+2) % 3
    // Branch-free body (one basic block, so the scheduler can interleave loads / LDS traffic with the MFMAs): in
    // the last two chunks the staging of "chunk it+2" and the fragment prefetch of "chunk it+1" are redundant
    // re-reads whose results are never consumed.
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      // fragments of the next sub-step (or of the next chunk) are fetched while this sub-step's MFMAs run
      if constexpr ((ABL & 8) == 0) {
        if (q + 1 < NQ) {
          read_frag(slot, q + 1, fa[(q + 1) & 1], fb[(q + 1) & 1]);
        } else {
          read_frag(slot1, 0, fa[NQ & 1], fb[NQ & 1]);
        }
      }
      if (q == 0) {
        if (!(ABL & 16)) store_lds(slot2, a_reg, b_reg, a_okmask);  // chunk it+2, loaded during the previous chunk's MFMAs -> its ring slot
        if (!(ABL & 1)) load_global(a_reg, b_reg, a_okmask);    // chunk it+3 -> registers; has a whole chunk of MFMAs to land
      }
      mma(fa[q & 1], fb[q & 1]);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr ((ABL & 64) != 0) {     // overlap probe: 64 independent VALU ops per chunk
      float t0 = 1.f, t1 = 2.f, t2 = 3.f, t3 = 4.f;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3"
                     : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
      asm volatile("" ::"v"(t0), "v"(t1), "v"(t2), "v"(t3));
    }
    if constexpr ((ABL & 128) != 0) {    // overlap probe: 64 SALU ops per chunk
      int u0 = 1, u1 = 2;
#pragma unroll
      for (int u = 0; u < 32; ++u) asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3" : "+s"(u0), "+s"(u1) : : "scc");
      asm volatile("" ::"s"(u0), "s"(u1));
    }
    if constexpr ((ABL & 256) != 0) {    // overlap probe: 8 x ds_read_b128 + wait per chunk, results unused
      f32x4 d0, d1, d2, d3, d4, d5, d6, d7;
      const float* base = smem + (wm * WTM + li) * LS + lh * VEC;
      d0 = *reinterpret_cast<const volatile f32x4*>(base);
      d1 = *reinterpret_cast<const volatile f32x4*>(base + 8);
      d2 = *reinterpret_cast<const volatile f32x4*>(base + 32 * LS);
      d3 = *reinterpret_cast<const volatile f32x4*>(base + 32 * LS + 8);
      d4 = *reinterpret_cast<const volatile f32x4*>(base + BM * LS);
      d5 = *reinterpret_cast<const volatile f32x4*>(base + BM * LS + 8);
      d6 = *reinterpret_cast<const volatile f32x4*>(base + (BM + 32) * LS);
      d7 = *reinterpret_cast<const volatile f32x4*>(base + (BM + 32) * LS + 8);
      asm volatile("" ::"v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));
    }
#endif
    if constexpr ((ABL & 4) == 0) __syncthreads();
    if constexpr ((NQ & 1) != 0) {   // odd NQ: the prefetched q=0 fragments sit in set 1, move them to set 0
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < VEC; ++e) fa[0][i][e] = fa[1][i][e];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < VEC; ++e) fb[0][j][e] = fb[1][j][e];
    }
    slot = slot1;
  }

  // ---- epilogue (conv_igemm_common.h) ----
  if constexpr ((ABL & 32) == 0) {
    igemm_epilogue<BM, BN, WM, WN, NT>(p, acc, smem, m0, n0, tile_m, split, tid, p.ksplit > 1);
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::"v"(acc[i][j]));   // keep the accumulators (and the MFMAs feeding them) alive
#endif
      }
  }
}

// Sum of the split-K partial tiles: out[m][n] (+)= sum_s ws[s][m][n] + bias[n], and - for BN layers - the per-row-tile
// (mean, M2) statistics in the same format the conv epilogue emits (tile_m rows per tile).  HBM-bound: reads
// ksplit*M*Cout floats once, writes M*Cout.  Grid = (row tiles, 64-channel slabs); thread = one float4 of channels
// (16 per slab) x one of 16 row lanes; Welford per thread, Chan-combine of the 16 row lanes through LDS.
// (Winograd plans have their own finishing pass: wino_output_kernel, conv_wino.hip.)
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ ws, int ksplit, float* out, int ldout,
                                                     const float* __restrict__ bias, float* stats, int M,
                                                     int Cout, int tile_m, int accumulate,
                                                     const float* __restrict__ escale, float act_slope,
                                                     int row0, const float* __restrict__ bn_raw, int bn_ld,
                                                     const float* __restrict__ bn_scale,
                                                     const float* __restrict__ bn_shift,
                                                     const float* __restrict__ bn_mean,
                                                     const float* __restrict__ bn_invstd, float bn_slope,
                                                     float* bn_partial, int bn_nslot, int ntile_all) {
  // rows [row0, M) (row0 a multiple of tile_m); workspace row m sits at m - row0, split stride (M - row0) rows
  const int tid = threadIdx.x, gl = tid & 15, pp = tid >> 4;
  const int c = blockIdx.y * 64 + gl * 4;
  const int m0 = row0 + blockIdx.x * tile_m, m1 = min(M, m0 + tile_m);
  const int64_t ws_rows = M - row0;
  ws -= (int64_t)row0 * Cout;
  const bool cok = c < Cout;           // Cout % 4 == 0 on this path (checked by the launcher)
  float cnt = 0.f;
  f32x4 mean = {0.f, 0.f, 0.f, 0.f}, m2 = {0.f, 0.f, 0.f, 0.f};
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (cok && bias != nullptr) b4 = *reinterpret_cast<const f32x4*>(bias + c);
  f32x4 e4 = {1.f, 1.f, 1.f, 1.f};
  if (cok && escale != nullptr) e4 = *reinterpret_cast<const f32x4*>(escale + c);
  // fused BatchNorm-backward reductions of the block that produced this gradient's activation (ConvArgs::bn_*)
  const bool bnb = bn_partial != nullptr;
  f32x4 bsc = {0.f, 0.f, 0.f, 0.f}, bsh = bsc, bmu = bsc, bis = bsc, s1 = bsc, s2 = bsc;
  if (bnb && cok) {
    bsc = *reinterpret_cast<const f32x4*>(bn_scale + c); bsh = *reinterpret_cast<const f32x4*>(bn_shift + c);
    bmu = *reinterpret_cast<const f32x4*>(bn_mean + c); bis = *reinterpret_cast<const f32x4*>(bn_invstd + c);
  }
  if (cok) {
    for (int m = m0 + pp; m < m1; m += 16) {
      f32x4 v;
      v = *reinterpret_cast<const f32x4*>(ws + (int64_t)m * Cout + c);
      int sp = 1;
      for (; sp + 3 < ksplit; sp += 4) {       // four partial loads in flight; summation order unchanged (sp ascending)
        f32x4 u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] = *reinterpret_cast<const f32x4*>(ws + ((int64_t)(sp + q) * ws_rows + m) * Cout + c);
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[0] += u[q][0]; v[1] += u[q][1]; v[2] += u[q][2]; v[3] += u[q][3]; }
      }
      for (; sp < ksplit; ++sp) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(ws + ((int64_t)sp * ws_rows + m) * Cout + c);
        v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
      }
      cnt += 1.f;
      const float inv = 1.f / cnt;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = v[k] - mean[k];     // Welford on the raw (bias-free) value
        mean[k] += d * inv;
        m2[k] += d * (v[k] - mean[k]);
      }
      float* dst = out + (int64_t)m * ldout + c;
      f32x4 o = {v[0] * e4[0] + b4[0], v[1] * e4[1] + b4[1], v[2] * e4[2] + b4[2], v[3] * e4[3] + b4[3]};
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.f ? o[k] : o[k] * act_slope;
      if (accumulate) {
        const f32x4 prev = *reinterpret_cast<const f32x4*>(dst);
        o[0] += prev[0]; o[1] += prev[1]; o[2] += prev[2]; o[3] += prev[3];
      }
      *reinterpret_cast<f32x4*>(dst) = o;
      if (bnb) {
        const f32x4 xr = *reinterpret_cast<const f32x4*>(bn_raw + (int64_t)m * bn_ld + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float y = xr[k] * bsc[k] + bsh[k];
          const float dyv = y > 0.f ? o[k] : o[k] * bn_slope;
          s1[k] += dyv;
          s2[k] += dyv * ((xr[k] - bmu[k]) * bis[k]);
        }
      }
    }
  }
  __shared__ float red[16][16][9];   // [row lane][channel quad][cnt, mean x4, m2 x4]  (or [-, s1 x4, s2 x4])
  if (bnb) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[pp][gl][1 + k] = s1[k]; red[pp][gl][5 + k] = s2[k]; }
    __syncthreads();
    if (tid < 64) {
      const int q = tid >> 2, k = tid & 3;
      const int ch = blockIdx.y * 64 + tid;
      if (ch < Cout) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < 16; ++w) { a += red[w][q][1 + k]; b += red[w][q][5 + k]; }
        const int t = m0 / tile_m;
        if (ntile_all > bn_nslot) {
          float* dst = bn_partial + ((int64_t)(t % bn_nslot) * Cout + ch) * 2;
          atomicAdd(dst, a);
          atomicAdd(dst + 1, b);
        } else {
          float* dst = bn_partial + ((int64_t)t * Cout + ch) * 2;
          dst[0] = a;
          dst[1] = b;
        }
      }
    }
    return;
  }
  if (stats == nullptr) return;
  red[pp][gl][0] = cnt;
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[pp][gl][1 + k] = mean[k]; red[pp][gl][5 + k] = m2[k]; }
  __syncthreads();
  if (tid < 64) {                     // one thread per channel of the slab
    const int q = tid >> 2, k = tid & 3;
    const int ch = blockIdx.y * 64 + tid;
    if (ch < Cout) {
      float n_ = red[0][q][0], mu = red[0][q][1 + k], s2 = red[0][q][5 + k];
      for (int w = 1; w < 16; ++w) chan_combine(n_, mu, s2, red[w][q][0], red[w][q][1 + k], red[w][q][5 + k]);
      float* st = stats + ((int64_t)(m0 / tile_m) * Cout + ch) * 2;
      st[0] = mu;
      st[1] = s2;
    }
  }
}

template <int BM, int BN, int WM, int WN, int BK, int ABL = 0, int PASS = 0>
static int launch_cfg(ConvArgs a, hipStream_t stream, int extra_lds = 0) {
  a.ntile_m = ssp_cdiv(a.M, BM);
  a.ntile_n = ssp_cdiv(a.Cout, BN);
  const int niter_total = a.R * a.R * (a.Cin / BK);
  a.it_per_split = ssp_cdiv(niter_total, a.ksplit);
  a.ksplit = ssp_cdiv(niter_total, a.it_per_split);
  dim3 grid(a.ntile_m * a.ntile_n * a.ksplit), block(WM * WN * 64);
  const int lds_bytes = 3 * (BM + BN) * (BK + 4) * 4 + extra_lds;
  auto kern = conv_igemm_kernel<BM, BN, WM, WN, BK, ABL, PASS>;
  static SspKernelCache cache;   // per instantiation, per device
  if (int rc = ssp_kernel_prepare((const void*)kern, lds_bytes, WM * WN * 64, &cache, nullptr, "conv_igemm")) return rc;
  hipLaunchKernelGGL(kern, grid, block, lds_bytes, stream, a);
  SSP_CHECK_LAUNCH("conv_igemm");
  return SSP_OK;
}

int ssp_conv_igemm_dma_launch(ConvArgs& a, int bm, int slots, int tail_ks, int is_dgrad, hipStream_t stream);   // conv_igemm_dma.hip

// Tile / split selection, a pure function of the layer shape (shared by the launcher and by the host-side queries
// that size the BN-statistics and split-K workspaces).  128x128 tiles are the workhorse.  A grid that is not a whole
// number of resident waves of workgroups (2 per CU = 512 slots for 128x128, 3 per CU = 768 for 64x128) idles the chip
// in its last wave: the 13x13 layers (680 tiles) would run 2 waves at 66 %.  Those layers split the K loop
// (ksplit workgroups per tile, partial tiles summed by splitk_reduce_kernel) so the grid becomes ~4 full waves;
// mid-size grids use 64-row tiles instead.
struct IgemmPlan { int bm, ksplit, slots, tail; };   // slots: LDS ring depth of the LDS-direct kernel (0 = its default);
                                                      // tail: K split of the last partial wave's tiles (hybrid launch), 0 = off
static IgemmPlan select_plan(int M, int Cin, int Cout, int R, int plan_code) {
  IgemmPlan pl = {256, 1, 0, 0};
  if (Cout <= 64) {
    if (Cout > 32 && Cin % 16 == 0 && ssp_option(SSP_OPT_IGEMM_VARIANT) != 50) pl.bm = 128;   // 128x64 LDS-direct tiles
    // Thin layers on small grids (the 1x1 head conv 1024 -> 20 and the 512 -> 64 route conv; at batch 1 also at 672 x 672):
    // one column of a few dozen 256- / 128-row tiles leaves most CUs idle while each workgroup walks the whole K loop
    // (44 us for 18 MFLOP at batch 1).  Split K so that ~2 workgroups per CU stream the filters; >= 8 chunks per split.
    if (Cin % 16 == 0 && Cout % 4 == 0 && ssp_option(SSP_OPT_IGEMM_VARIANT) != 50 && ssp_option(SSP_OPT_IGEMM_VARIANT) != 30) {
      const int64_t tiles = ssp_cdiv(M, pl.bm);
      const int niter = R * R * (Cin / 16);
      if (tiles < 256) {
        int64_t ks = 512 / tiles;
        if (ks > niter / 8) ks = niter / 8;
        if (ks > 8) ks = 8;
        if (ks >= 2) pl.ksplit = (int)ks;
      }
    }
    return pl;
  }
  pl.bm = 128;
  // explicit plan (the `plan` argument of the entry points, chosen per launch shape by the engine's autotuner:
  // tail*100000 + bm*100 + ksplit*10 + slots); 0 = this heuristic.  The process-wide "igemm_plan" knob only stands in
  // when the caller passed 0 (bench / tools experiments).  Invalid requests fall back to auto.
  const int forced = plan_code > 0 ? plan_code : ssp_option(SSP_OPT_IGEMM_PLAN);
  if (forced > 0) {
    const int ftail = forced / 100000, fbm = (forced / 100) % 1000, fks = (forced / 10) % 10, fsl = forced % 10;
    const int niter16 = (Cin % 16 == 0) ? R * R * (Cin / 16) : 0;
    const bool ok = (fbm == 64 || fbm == 128) && fks >= 1 && (fsl == 3 || fsl == 4 || fsl == 8) &&
                    (fks == 1 || (niter16 / fks >= 8 && Cout % 4 == 0)) && Cin % 16 == 0 &&
                    (ftail == 0 || (ftail >= 2 && ftail <= 9 && fks == 1 && Cout % 4 == 0 && niter16 / ftail >= 8));
    if (ok) { pl.bm = fbm; pl.ksplit = fks; pl.slots = fsl; pl.tail = ftail; return pl; }
  }
  const int variant = ssp_option(SSP_OPT_IGEMM_VARIANT);
  if (variant == 20 || variant == 22) { pl.bm = 64; return pl; }
  if (variant >= 70 && variant <= 75) {                      // A/B: forced split-K x(variant-68) on 128x128 tiles
    const int niter = R * R * (Cin / 16);
    if (Cin % 16 == 0 && Cout % 4 == 0 && niter / (variant - 68) >= 8) pl.ksplit = variant - 68;
    return pl;
  }
  if (variant != 0 && variant != 30 && variant != 50 && variant != 61 && variant != 62 && variant != 63) return pl;   // experiment variants: plain 128x128, no split
  const int64_t t128 = (int64_t)ssp_cdiv(M, 128) * ssp_cdiv(Cout, 128);
  const int64_t t64 = (int64_t)ssp_cdiv(M, 64) * ssp_cdiv(Cout, 128);
  if (t128 > 1400) return pl;
  auto wave_eff = [](int64_t wgs, int slots) { return (double)wgs / (double)(((wgs + slots - 1) / slots) * slots); };
  double best = wave_eff(t128, 512);
  const double e64 = 0.97 * wave_eff(t64, 768);
  if (e64 > best) { best = e64; pl.bm = 64; }
  if (variant == 30 || Cin % 16 != 0 || Cout % 4 != 0) return pl;   // variant 30: no split-K (A/B experiments)
  const int niter = R * R * (Cin / 16);
  const double flop_time = 2.0 * M * (double)Cout * R * R * Cin / 110e12;
  for (int sp = 2; sp <= 6; ++sp) {
    if (niter / sp < 128) break;   // short K per workgroup: prologue/epilogue and the reduce pass eat the gain (measured)
    const double traffic = (2.0 * sp + 1.0) * (double)M * Cout * 4.0 / 4.0e12;   // partial writes + reads + final write
    const double eff = wave_eff(t128 * sp, 512) / (1.0 + traffic / flop_time);
    if (eff > best + 0.02) { best = eff; pl.bm = 128; pl.ksplit = sp; }
  }
  return pl;
}
int ssp_conv_tile_m(int M, int Cin, int Cout, int R, int plan) {
  if (ssp_wino_plan_tile(plan)) return 0;      // Winograd plans: statistics per group of SSP_WINO_TG tiles, counted format
  return select_plan(M, Cin, Cout, R, plan).bm;
}
int64_t ssp_conv_ws_floats(int M, int Cin, int Cout, int R, int plan) {
  IgemmPlan pl = select_plan(M, Cin, Cout, R, plan);
  const int deepest = pl.ksplit > pl.tail ? pl.ksplit : pl.tail;      // a hybrid launch parks at most M rows x tail
  return deepest > 1 ? (int64_t)deepest * M * Cout : 0;
}

// ---- Winograd path (conv_wino.hip): plan codes 9000000 (F(2x2,3x3)) / 8000000 (F(4x4,3x3)) + bm * 100 + 10 + slots ----
// `wt` is then the TRANSFORMED filter U [P][Cout][Cin] (ssp_wino_filter_transform_t, P = 16 / 36), the workspace holds the
// transformed input V [P][T][Cin] followed by the GEMM output M [P][T][Cout] (ssp_conv_workspace_floats says how much),
// and the launch is three kernels: input transform, ONE batched launch of the LDS-direct GEMM kernel (gridDim.y = P), and
// the finishing pass wino_output_kernel (inverse transform + everything the split-K finish does).
int64_t ssp_wino_ws_floats(int B, int H, int W, int Cin, int Cout, int tile) {
  return ssp_wino_planes(tile) * ssp_wino_tiles(B, H, W, tile) * ((int64_t)Cin + Cout);
}

static int wino_launch(ConvArgs& a, int B, int H, int W, float* ws, int64_t ws_floats, int plan, int prof_kind,
                       hipStream_t stream) {
  const int bm = (plan / 100) % 1000, slots = plan % 10, tile = ssp_wino_plan_tile(plan), P = ssp_wino_planes(tile);
  SSP_CHECK_ARG(a.R == 3 && a.Cin % 16 == 0 && a.Cout >= 64 && a.Cout % 4 == 0 && a.ldout % 4 == 0 && (((uintptr_t)a.out) & 15) == 0,
                "conv (Winograd plan): needs a 3x3 filter, Cin %% 16 == 0, Cout >= 64 and %% 4 == 0, an aligned output");
  SSP_CHECK_ARG((bm == 64 || bm == 128) && (slots == 3 || slots == 4 || slots == 8) && (plan / 10) % 10 == 1 &&
                    (plan / 100000) % 10 == 0, "conv: bad Winograd plan code %d", plan);
  const int64_t T = ssp_wino_tiles(B, H, W, tile);
  SSP_CHECK_ARG(T < (1ll << 31) && P * T * (int64_t)a.Cin < (1ll << 40), "conv (Winograd plan): too many tiles");
  SSP_CHECK_ARG(ws != nullptr && ws_floats >= ssp_wino_ws_floats(B, H, W, a.Cin, a.Cout, tile) && (((uintptr_t)ws) & 15) == 0,
                "conv (Winograd plan): needs a workspace of %lld floats (ssp_conv_workspace_floats)",
                (long long)ssp_wino_ws_floats(B, H, W, a.Cin, a.Cout, tile));
  SSP_CHECK_ARG((int64_t)(128 + 2) * a.Cin * 4 + (int64_t)a.Cin * 4 < (1ll << 31), "conv (Winograd plan): Cin too large");
  if (a.bn_partial != nullptr) SSP_CHECK_ARG(a.bn_ld % 4 == 0, "conv (Winograd plan): bn_ld must be a multiple of 4");
  SspProfScope prof(prof_kind, stream, 2.0 * (double)a.M * a.Cout * 9.0 * a.Cin);      // algorithmic (direct) FLOPs
  float* V = ws;
  float* Mw = ws + P * T * a.Cin;
  const int wkind = prof_kind == SSP_PROF_CONV_DGRAD ? SSP_PROF_WINO_DGRAD : SSP_PROF_WINO_FWD;
  if (int rc = ssp_wino_input_launch(a.in, a.ldin, V, B, H, W, a.Cin, tile, wkind, stream)) return rc;
  ConvArgs g = a;
  g.in = V; g.wt = a.wt; g.out = Mw; g.bias = nullptr; g.escale = nullptr; g.act_slope = 1.f; g.stats = nullptr;
  g.H = 1; g.W = (int)T; g.ldin = a.Cin; g.ldout = a.Cout; g.R = 1; g.M = (int)T; g.accumulate = 0;
  g.divW = ssp_fastdiv((unsigned)T); g.divH = ssp_fastdiv(1u);
  g.ws = nullptr; g.ksplit = 1; g.probe = 0;
  g.bn_partial = nullptr; g.bn_raw = nullptr;
  g.batch = P; g.batch_in = T * a.Cin; g.batch_wt = (int64_t)a.Cout * a.Cin; g.batch_out = T * a.Cout;
  if (int rc = ssp_conv_igemm_dma_launch(g, bm, slots, 0, prof_kind == SSP_PROF_CONV_DGRAD, stream)) return rc;
  WinoOutArgs o;
  o.Mw = Mw; o.out = a.out; o.bias = a.bias; o.escale = a.escale; o.act_slope = a.act_slope; o.stats = a.stats;
  o.Cout = a.Cout; o.ldout = a.ldout; o.accumulate = a.accumulate;
  o.bn_raw = a.bn_raw; o.bn_scale = a.bn_scale; o.bn_shift = a.bn_shift; o.bn_mean = a.bn_mean; o.bn_invstd = a.bn_invstd;
  o.bn_partial = a.bn_partial; o.bn_nslot = a.bn_nslot; o.bn_ld = a.bn_ld; o.bn_slope = a.bn_slope;
  return ssp_wino_output_launch(o, B, H, W, tile, wkind, stream);
}

int ssp_conv_igemm_launch(const float* in, const float* wt, float* out, const float* bias, float* stats, int B, int H,
                          int W, int Cin, int Cout, int ldin, int ldout, int R, int accumulate, float* ws,
                          int64_t ws_floats, int plan, int prof_kind, hipStream_t stream, const float* escale,
                          float act_slope, const SspBnBwdFuse* bnb) {
  SSP_CHECK_ARG(R == 1 || R == 3, "conv: only 1x1 and 3x3 filters are supported (got %d)", R);
  SSP_CHECK_ARG(Cin % 4 == 0 && Cin > 0, "conv: Cin must be a positive multiple of 4 (got %d)", Cin);
  SSP_CHECK_ARG(ldin % 4 == 0 && ldin >= Cin, "conv: ldin must be a multiple of 4 and >= Cin");
  SSP_CHECK_ARG(ldout >= Cout && Cout > 0, "conv: ldout < Cout");
  SSP_CHECK_ARG((int64_t)B * H * W < (1ll << 31), "conv: too many pixels");
  SSP_CHECK_ARG((((uintptr_t)in) & 15) == 0 && (((uintptr_t)wt) & 15) == 0, "conv: in/wt must be 16-byte aligned");
  ConvArgs a;
  a.in = in; a.wt = wt; a.out = out; a.bias = bias; a.stats = stats; a.escale = escale; a.act_slope = act_slope;
  a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ldin = ldin; a.ldout = ldout; a.R = R;
  a.M = B * H * W; a.accumulate = accumulate;
  a.divW = ssp_fastdiv((unsigned)W); a.divH = ssp_fastdiv((unsigned)H);
  a.xcd_remap = ssp_option(SSP_OPT_IGEMM_XCD);
  a.probe = ssp_option(SSP_OPT_IGEMM_VARIANT) == 63 ? 2 : 0;      // 63: A/B switch - the generic (predicated) epilogue everywhere
  a.tail_begin = 0; a.tail_ks = 0; a.tail_it_per_split = 0; a.ws_row0 = 0; a.ws_rows = a.M; a.col_major = 0;
  a.bn_raw = nullptr; a.bn_scale = a.bn_shift = a.bn_mean = a.bn_invstd = nullptr; a.bn_partial = nullptr;
  a.bn_ld = 0; a.bn_slope = 1.f; a.bn_nslot = 1;
  a.batch = 0; a.batch_in = a.batch_wt = a.batch_out = 0;
  if (bnb != nullptr && bnb->partial != nullptr) {
    SSP_CHECK_ARG(accumulate == 0 && bias == nullptr && stats == nullptr && escale == nullptr,
                  "conv: the fused BatchNorm-backward reductions need a plain (non-accumulating) data-gradient launch");
    SSP_CHECK_ARG(bnb->raw != nullptr && bnb->scale != nullptr && bnb->shift != nullptr && bnb->mean != nullptr &&
                  bnb->invstd != nullptr && bnb->ldraw >= Cout && Cout % 4 == 0 && bnb->ldraw % 4 == 0 &&
                  ((((uintptr_t)bnb->raw) | ((uintptr_t)bnb->scale) | ((uintptr_t)bnb->shift) | ((uintptr_t)bnb->mean) |
                    ((uintptr_t)bnb->invstd)) & 15) == 0,
                  "conv: bad BatchNorm-backward operands (16-byte aligned, Cout %% 4 == 0, ldraw >= Cout)");
    a.bn_raw = bnb->raw; a.bn_ld = bnb->ldraw; a.bn_scale = bnb->scale; a.bn_shift = bnb->shift; a.bn_mean = bnb->mean;
    a.bn_invstd = bnb->invstd; a.bn_slope = bnb->slope; a.bn_partial = bnb->partial;
    SSP_CHECK_ARG(bnb->nslot >= 1, "conv: the BatchNorm-backward partial buffer needs at least one row");
    a.bn_nslot = bnb->nslot;
  }
  if (ssp_wino_plan_fused(plan)) {
    a.ksplit = 1; a.ws = nullptr;
    return ssp_wino_fused_launch(a, B, H, W, prof_kind, stream);
  }
  if (ssp_wino_plan_tile(plan)) return wino_launch(a, B, H, W, ws, ws_floats, plan, prof_kind, stream);
  IgemmPlan pl = select_plan(a.M, Cin, Cout, R, plan);
  if (Cout <= 64 && pl.ksplit > 1 &&
      (ws == nullptr || ws_floats < (int64_t)pl.ksplit * a.M * Cout || ldout % 4 != 0 || (((uintptr_t)out) & 15) != 0))
    pl.ksplit = 1;      // thin-layer split is an optimisation only: without a workspace the tile walks its whole K loop
  a.ksplit = pl.ksplit;
  a.ws = ws;
  if (pl.ksplit > 1)
    SSP_CHECK_ARG(ldout % 4 == 0 && (((uintptr_t)out) & 15) == 0, "conv: split-K output must be 16-byte aligned with ldout % 4 == 0");
  if (pl.ksplit > 1)
    SSP_CHECK_ARG(ws != nullptr && ws_floats >= (int64_t)pl.ksplit * a.M * Cout,
                  "conv: this shape runs split-K x%d and needs a workspace of %lld floats (ssp_conv_workspace_floats)",
                  pl.ksplit, (long long)pl.ksplit * a.M * Cout);
  if (pl.tail > 1) {
    SSP_CHECK_ARG(ws != nullptr && ws_floats >= (int64_t)pl.tail * a.M * Cout && ldout % 4 == 0 && (((uintptr_t)out) & 15) == 0,
                  "conv: the hybrid plan (tail split x%d) needs an aligned output and a workspace of %lld floats "
                  "(ssp_conv_workspace_floats)", pl.tail, (long long)pl.tail * a.M * Cout);
  }
  SspProfScope prof(prof_kind, stream, 2.0 * (double)a.M * Cout * (double)(R * R * Cin));
  const int variant = ssp_option(SSP_OPT_IGEMM_VARIANT);   // experiments: tools/conv_bench.py --opt igemm_variant=N
  const int bk = (Cin % 32 == 0 && (variant == 2 || variant == 4 || variant == 5)) ? 32 : ((Cin % 16 == 0) ? 16 : 4);
  int rc = SSP_OK;
  if (Cout > 64) {
    switch (variant) {
      case 3: if (bk >= 16) return launch_cfg<256, 128, 4, 2, 16>(a, stream); break;
      case 4: if (bk == 32) return launch_cfg<128, 128, 2, 2, 32, 1>(a, stream); break;
      case 5: if (bk == 32) return launch_cfg<128, 128, 2, 2, 32, 2>(a, stream); break;
#ifdef SSP_PROBES   // timing probes that skip loads / the epilogue: WRONG RESULTS on purpose, never in the shipped library (make PROBES=1)
      case 10: if (bk >= 16) return launch_cfg<128, 128, 2, 2, 16, 1>(a, stream); break;            // no loads
      case 14: if (bk >= 16) return launch_cfg<128, 128, 2, 2, 16, 1 | 4 | 8 | 16>(a, stream); break;  // MFMA only
      case 17: if (bk >= 16) return launch_cfg<128, 128, 2, 2, 16, 1 | 4 | 8 | 16 | 32>(a, stream); break;  // MFMA only, no epilogue
      case 40: if (bk >= 16) return launch_cfg<128, 128, 2, 2, 16, 29 | 64>(a, stream); break;    // MFMA only + VALU probe
      case 41: if (bk >= 16) return launch_cfg<128, 128, 2, 2, 16, 29 | 128>(a, stream); break;   // MFMA only + SALU probe
      case 42: if (bk >= 16) return launch_cfg<128, 128, 2, 2, 16, 29 | 256>(a, stream); break;   // MFMA only + LDS-read probe
      case 43: if (bk >= 16) return launch_cfg<128, 128, 2, 2, 16, 29 | 64 | 128 | 256>(a, stream); break;
      case 18: if (bk >= 16) return launch_cfg<128, 128, 2, 2, 16, 32>(a, stream); break;   // full loop, no epilogue
#endif
      case 21: if (bk >= 16) return launch_cfg<128, 64, 2, 2, 16>(a, stream); break;
      case 22: if (bk >= 16) return launch_cfg<64, 64, 2, 2, 16>(a, stream); break;
      default: break;
    }
    if (bk >= 16 && variant != 50 && ((int64_t)(128 + 2 * W + 2) * ldin * 4 + (int64_t)Cin * 4 < (1ll << 31))) {
      rc = ssp_conv_igemm_dma_launch(a, pl.bm, pl.slots, pl.tail, prof_kind == SSP_PROF_CONV_DGRAD, stream);      // LDS-direct loader (conv_igemm_dma.hip)
    } else if (pl.bm == 64) {
      rc = (bk >= 16) ? launch_cfg<64, 128, 2, 2, 16>(a, stream)
                      : (prof_kind == SSP_PROF_CONV_DGRAD ? launch_cfg<64, 128, 2, 2, 4, 0, 1>(a, stream) : launch_cfg<64, 128, 2, 2, 4>(a, stream));
    } else if (bk == 32) {
      rc = launch_cfg<128, 128, 2, 2, 32>(a, stream);
    } else if (bk == 16) {
      rc = launch_cfg<128, 128, 2, 2, 16>(a, stream);
    } else {
      rc = prof_kind == SSP_PROF_CONV_DGRAD ? launch_cfg<128, 128, 2, 2, 4, 0, 1>(a, stream) : launch_cfg<128, 128, 2, 2, 4>(a, stream);
    }
  } else if (Cout > 32) {
    if (pl.bm == 128) {
      SSP_CHECK_ARG((int64_t)(128 + 2 * W + 2) * ldin * 4 + (int64_t)Cin * 4 < (1ll << 31),
                    "conv: image rows too long for the 32-bit tile offsets of the LDS-direct loader");
      rc = ssp_conv_igemm_dma_launch(a, 128, 0, 0, prof_kind == SSP_PROF_CONV_DGRAD, stream);
    } else
      rc = (bk >= 16) ? launch_cfg<256, 64, 4, 1, 16>(a, stream) : launch_cfg<256, 64, 4, 1, 4>(a, stream);
  } else if (bk >= 16 && variant != 50 && ((int64_t)(256 + 2 * W + 2) * ldin * 4 + (int64_t)Cin * 4 < (1ll << 31))) {
    rc = ssp_conv_igemm_dma_launch(a, 256, 0, 0, prof_kind == SSP_PROF_CONV_DGRAD, stream);   // 256x32 LDS-direct tiles
  } else {
    rc = (bk >= 16) ? launch_cfg<256, 32, 4, 1, 16>(a, stream) : launch_cfg<256, 32, 4, 1, 4>(a, stream);
  }
  if (rc != SSP_OK) return rc;
  // rows per reduce workgroup: the statistics formats are per conv tile (pl.bm rows); a launch that produces none is free
  // to use 16-row blocks (one row per thread: 4-8x the workgroups on the small grids of batch-1 inference)
  const int rt = (stats == nullptr && a.bn_partial == nullptr && a.M <= 16384) ? 16 : pl.bm;
  if (pl.ksplit > 1) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ssp_cdiv(a.M, rt), ssp_cdiv(Cout, 64)), dim3(256), 0, stream, ws, pl.ksplit, out, ldout,
                       bias, stats, a.M, Cout, rt, accumulate, escale, act_slope, 0, a.bn_raw, a.bn_ld, a.bn_scale,
                       a.bn_shift, a.bn_mean, a.bn_invstd, a.bn_slope, a.bn_partial, a.bn_nslot, ssp_cdiv(a.M, pl.bm));
    SSP_CHECK_LAUNCH("splitk_reduce");
  } else if (a.tail_ks > 1) {     // hybrid launch: only the tail rows were left as partials
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ssp_cdiv(a.M - a.ws_row0, rt), ssp_cdiv(Cout, 64)), dim3(256), 0, stream, ws,
                       a.tail_ks, out, ldout, bias, stats, a.M, Cout, rt, accumulate, escale, act_slope, a.ws_row0,
                       a.bn_raw, a.bn_ld, a.bn_scale, a.bn_shift, a.bn_mean, a.bn_invstd, a.bn_slope, a.bn_partial, a.bn_nslot,
                       ssp_cdiv(a.M, pl.bm));
    SSP_CHECK_LAUNCH("splitk_reduce(tail)");
  }
  return SSP_OK;
}
