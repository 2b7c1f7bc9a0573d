// Filter gradient, LDS-direct loader variant (tiles with >= 64 couts and >= 64 cins).
//
// Same contraction, tiling, split-over-pixels and atomic accumulation as conv_wgrad.hip; the operands reach the LDS
// with `buffer_load_dwordx4 ... lds` so the K loop carries no VGPR staging, no ds_write and no zero-fill selects
// (the fp32 MFMA shares the SIMD's vector lanes with every VALU instruction - see conv_igemm_dma.hip):
//   * dY: lane offsets are constant; the descriptor's base advances by 16 pixel rows per chunk and its num_records
//     shrinks with the rows left, so the rows past the end of this workgroup's pixel range read as zeros for free;
//   * X (shifted by the workgroup's filter tap): the lane offset is constant too, but the image-border test depends
//     on the pixel, which moves every chunk: each lane walks its pixel's (x, y) incrementally (no division) and
//     swaps its offset for an out-of-range one when the tap leaves the image;
//   * [pixel][channel] LDS images are what the DMA writes naturally and what the conflict-free ds_read_b32
//     operand fetch wants, so no swizzle is needed here;
//   * 4-slot ring, loads 3 chunks ahead, one counted vmcnt + raw barrier per chunk, K loop unrolled x4.
#include <type_traits>

#include "ssp_common.h"

struct WgradArgs {
  const float* dy;
  const float* x;
  float* dw;
  int H, W, Cin, Cout, lddy, ldx, R, M;
  int ntile_co, ntile_ci, nsplit, chunk_m;
  int xcd_order;  // XCD-aware workgroup order (see the kernel): 1 = all tiles of a pixel range on one XCD, 2 = per (range, cout tile)
  int no_store;   // 2: generic epilogue (A/B switch, same results); 1: SSP_PROBES builds only - skip the atomic epilogue
  int fold;   // filter taps per cin tile: 1, or BNI / Cin when Cin < BNI (thin layers: two taps of 32 cins share a tile)
  // batched launch (the 16 transform positions of a Winograd filter gradient, conv_wino.hip): gridDim.y problems of one
  // shape; problem b reads dy + b * batch_dy, x + b * batch_x and accumulates into dw + b * batch_dw.  0 / 1: one problem.
  int batch;
  int64_t batch_dy, batch_x, batch_dw;
  // The caller promises that nothing else accumulates into dw during this launch and does not need the previous contents
  // (the transform-domain gradient dU of a Winograd filter gradient): a launch that does not split the pixel range then
  // writes its tiles with plain stores (store = 1, set by the launcher) and a split one zeroes `zero_bytes` bytes at
  // `zero_ptr` first - instead of a memset in front of every launch.
  int store;
  void* zero_ptr;
  size_t zero_bytes;
};

#define SSP_OOB 0x80000000u

typedef float f32x2 __attribute__((ext_vector_type(2)));

// T consecutive floats from the LDS in one ds_read_b32 / b64 / b128
template <int T>
__device__ __forceinline__ void lds_read_vec(const char* ptr, float (&out)[T]) {
  static_assert(T == 1 || T == 2 || T == 4, "1, 2 or 4 MFMA blocks per wave along a tile edge");
  if constexpr (T == 1) {
    out[0] = *reinterpret_cast<const float*>(ptr);
  } else if constexpr (T == 2) {
    const f32x2 v = *reinterpret_cast<const f32x2*>(ptr);
    out[0] = v[0]; out[1] = v[1];
  } else {
    const f32x4 v = *reinterpret_cast<const f32x4*>(ptr);
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
  }
}

// mask bit of the lane set -> a, else b: ONE VALU instruction for a wave-uniform 64-bit lane mask held in SGPRs
__device__ __forceinline__ unsigned lane_select(unsigned long long mask, unsigned a, unsigned b) {
  unsigned r = b;
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
#endif
  return r;
}

// FOLD: tiles that span several filter taps (Cin < BNI); their border test is per lane.  Otherwise the tap is uniform
// over the workgroup and the border test runs on the scalar unit (see the loader below).
// BVEC: also interleave the cin blocks (one vector read for the X operands too).  Fewer LDS instructions, but the
// lanes of an atomic then step TN floats apart and touch twice the cache lines - kept as an experiment.
template <int BMO, int BNI, int NSLOT = 4, bool FOLD = false, bool BVEC = false>
__global__ void __launch_bounds__(256, 2) conv_wgrad_dma_kernel(WgradArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int RA = 16, WM = 2, WN = 2;
  constexpr int WTM = BMO / WM, WTN = BNI / WN, TM = WTM / 32, TN = WTN / 32;
  constexpr int ABYTES = RA * BMO * 4, BBYTES = RA * BNI * 4, SLOTB = ABYTES + BBYTES;
  constexpr int APW = ABYTES / 1024 / 4, BPW = BBYTES / 1024 / 4;   // wave-instructions per wave per chunk
  constexpr int LPW = APW + BPW;
  static_assert(BMO % 64 == 0 && BNI % 64 == 0, "1-KiB DMA pieces are dealt to 4 waves");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = reinterpret_cast<char*>(smem);

  const int taps = p.R * p.R;
  const int tap_groups = (taps + p.fold - 1) / p.fold;
  // Workgroup order.  All taps x tiles of one pixel range read the same dY / X rows; dispatch hands consecutive
  // workgroup ids to the 8 XCDs round robin and each XCD has its own L2.  xcd_order: the workgroups of pixel range s sit
  // on XCD s % 8, consecutively in its dispatch order, so they stream those rows through that L2 together (the plain
  // order spreads each range over all XCDs and over time: every tap re-fetches the operands from HBM).
  int split, bid;
  if (p.xcd_order == 2) {
    // Layers with more tiles than an XCD holds at once (the 13 x 13 layers: 144 - 360 tiles): the unit that shares
    // operands is (pixel range s, cout tile c) - its taps x cin tiles (36 - 90 workgroups) all read the same
    // [range][BMO] dY slab, and the 9 taps of a cin tile the same X rows.  Units are dealt to the XCDs round robin and a
    // unit's workgroups are consecutive in its XCD's dispatch order, so the slab streams through ONE L2 once instead of
    // being fetched over the fabric by workgroups scattered over all eight (plain order: 2.4 MB per workgroup).
    const int Gu = tap_groups * p.ntile_ci;
    const int x = blockIdx.x % 8, q = blockIdx.x / 8;
    const int u = x + 8 * (q / Gu);
    if (u >= p.nsplit * p.ntile_co) return;      // grid padded to a multiple of 8 units
    split = u / p.ntile_co;
    bid = (u % p.ntile_co) * Gu + q % Gu;
  } else if (p.xcd_order) {
    const int G = tap_groups * p.ntile_ci * p.ntile_co;
    const int x = blockIdx.x % 8, q = blockIdx.x / 8;
    split = x + 8 * (q / G);
    bid = q % G;
    if (split >= p.nsplit) return;      // grid padded to a multiple of 8 pixel ranges
  } else {
    bid = blockIdx.x;
    split = bid % p.nsplit; bid /= p.nsplit;
  }
  const int tap = (bid % tap_groups) * p.fold; bid /= tap_groups;   // first tap of this workgroup's tile
  const int tile_ci = bid % p.ntile_ci;
  const int tile_co = bid / p.ntile_ci;
  const int co0 = tile_co * BMO, ci0 = tile_ci * BNI;
  const int pad = p.R >> 1;

  const int m_begin = split * p.chunk_m;
  const int m_end = min(p.M, m_begin + p.chunk_m);
  const int niter = (m_end - m_begin + RA - 1) / RA;
  if (niter <= 0) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the scalar pixel walker depends on it
  const int wm = wid / WN, wn = wid % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t bz = (p.batch > 1) ? (int64_t)blockIdx.y : 0;   // batched launch: problem index (scalar)
  const float* const dy_b = p.dy + bz * p.batch_dy;
  const float* const x_b = p.x + bz * p.batch_x;
  float* const dw_b = p.dw + bz * p.batch_dw;

  // ---- loader lanes: piece g (1 KiB) of a tile = bytes [g*1024, g*1024+1024) of the row-major [16][C] image ----
  // A 1-KiB piece of the X tile covers RP consecutive pixel rows (BNI = 128: 2 rows of 512 B, BNI = 64: 4 rows of
  // 256 B), i.e. LPR = 64 / RP lanes per pixel.  Without FOLD the filter tap is uniform, so "does this pixel's tap
  // fall inside the image" is a per-pixel-row question: the (x, y) of each of the wave's BPW * RP rows live in SGPRs,
  // walk 16 pixels per chunk on the scalar unit, and reach the lanes as one 64-bit lane mask -> one v_cndmask per
  // load.  (The fp32 MFMA shares the vector lanes with the VALU; the scalar unit is free.)
  constexpr int RP = 1024 / (BNI * 4), LPR = 64 / RP;
  static_assert(RP == 2 || RP == 4, "BNI is 64 or 128");
  unsigned a_voff[APW], b_off[BPW];
  int b_row[BPW], b_x[BPW], b_y[BPW], b_dx[BPW], b_dy[BPW];   // FOLD: per-lane walker
  int s_x[BPW][RP], s_y[BPW][RP];                             // !FOLD: scalar walker
  const int dy0 = tap / p.R - pad, dx0 = tap % p.R - pad;
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int byte = (wid + 4 * j) * 1024 + lane * 16;
    const int row = byte / (BMO * 4), col = (byte % (BMO * 4)) / 4;
    a_voff[j] = (co0 + col < p.Cout) ? (unsigned)((row * p.lddy + co0 + col) * 4) : SSP_OOB;
  }
#pragma unroll
  for (int j = 0; j < BPW; ++j) {
    const int byte = (wid + 4 * j) * 1024 + lane * 16;
    const int row = byte / (BNI * 4), col = (byte % (BNI * 4)) / 4;
    b_row[j] = row;
    // tile column -> (tap, cin): with FOLD the tile spans `fold` consecutive taps of all Cin channels; the tap's pixel
    // shift goes into the lane offset (the descriptor base sits at pixel (y-1, x-1) of the chunk's first row)
    const int tl = FOLD ? tap + col / p.Cin : tap;
    const int cc = FOLD ? col % p.Cin : ci0 + col;
    b_dy[j] = tl / p.R - pad;
    b_dx[j] = tl % p.R - pad;
    b_off[j] = (cc < p.Cin && tl < taps) ? (unsigned)(((row + b_dy[j] * p.W + b_dx[j] + p.W + 1) * p.ldx + cc) * 4) : SSP_OOB;
    const int m = m_begin + row;
    b_x[j] = m % p.W;
    b_y[j] = (m / p.W) % p.H;
#pragma unroll
    for (int t = 0; t < RP; ++t) {
      const int ms = m_begin + (wid + 4 * j) * RP + t;
      s_x[j][t] = ms % p.W;
      s_y[j][t] = (ms / p.W) % p.H;
    }
  }
  const unsigned oob = SSP_OOB;

  // Loader state for the NEXT chunk to stage (descriptors + lane offsets), prepared one step ahead: prep_next() is all
  // scalar work (plus one v_cndmask per X piece) and sits inside the MFMA block of the previous step, where the
  // scalar unit is otherwise idle; issue() at the top of a step is then nothing but the buffer loads.
  int ld_m = m_begin;   // first pixel of the chunk prep_next() prepares
  __amdgpu_buffer_rsrc_t rsrc_a, rsrc_b;
  unsigned b_voff[BPW];
  auto prep_next = [&]() {
    const int left = m_end - ld_m;                       // pixel rows still inside this workgroup's range
    const int64_t abase = (int64_t)ld_m * p.lddy;
    rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(dy_b + abase), 0, left > 0 ? (int)min((int64_t)left * p.lddy * 4, (int64_t)0x7fffffff) : 0, 0x00020000);
    rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(x_b + ((int64_t)ld_m - p.W - 1) * p.ldx), 0, (int)SSP_OOB, 0x00020000);   // base = pixel (y-1, x-1)
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
      if constexpr (FOLD) {
        const int yy = b_y[j] + b_dy[j], xx = b_x[j] + b_dx[j];
        const bool ok = (b_row[j] < left) && ((unsigned)yy < (unsigned)p.H) && ((unsigned)xx < (unsigned)p.W);
        b_voff[j] = ok ? b_off[j] : SSP_OOB;
        // walk this lane's pixel 16 positions ahead (W may be smaller than 16: up to two row wraps)
        int x = b_x[j] + RA, y = b_y[j];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const bool wrap = x >= p.W;
          x -= wrap ? p.W : 0;
          y += wrap ? 1 : 0;
          y = (y >= p.H) ? 0 : y;
        }
        b_x[j] = x;
        b_y[j] = y;
      } else {
        // branch-free scalar code (bitwise tests, selects) so that it stays in the MFMA block's basic block
        unsigned mlo = 0u, mhi = 0u;
#pragma unroll
        for (int t = 0; t < RP; ++t) {
          const int yy = s_y[j][t] + dy0, xx = s_x[j][t] + dx0;
          const int ok = (int)((wid + 4 * j) * RP + t < left) & (int)((unsigned)yy < (unsigned)p.H) & (int)((unsigned)xx < (unsigned)p.W);
          constexpr unsigned rowmask = (LPR == 32) ? 0xffffffffu : 0xffffu;
          const unsigned bits = (0u - (unsigned)ok) & (rowmask << ((t * LPR) & 31));
          if (t * LPR < 32) mlo |= bits; else mhi |= bits;
          int x = s_x[j][t] + RA, y = s_y[j][t];
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            const int wrap = (int)(x >= p.W);
            x -= (0 - wrap) & p.W;
            y += wrap;
            y &= 0 - (int)(y < p.H);
          }
          s_x[j][t] = x;
          s_y[j][t] = y;
        }
        const unsigned long long mask = ((unsigned long long)mhi << 32) | mlo;
        b_voff[j] = lane_select(mask, b_off[j], oob);
      }
    }
    ld_m += RA;
  };
  auto issue = [&](int slot_bytes) {
#pragma unroll
    for (int j = 0; j < APW; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(lds + slot_bytes + (wid + 4 * j) * 1024),
                                               16, a_voff[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < BPW; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void*)(lds + slot_bytes + ABYTES + (wid + 4 * j) * 1024),
                                               16, b_voff[j], 0, 0, 0);
  };

  // ---- operand fetch: k-step kk uses pixel rows 2kk + lh.  The wave's TM (TN) 32-row MFMA blocks are INTERLEAVED
  // over its WTM couts (WTN cins): block i holds couts {wm*WTM + TM*r + i}, so a lane's TM operands of one k-step are
  // TM consecutive floats of a pixel row - one ds_read_b64 / b128 with a 16-bit immediate offset (no address VALU) ----
  const unsigned fa_base = (unsigned)(lh * BMO + wm * WTM + li * TM) * 4u;
  // The cin blocks stay contiguous (block j = cins wn*WTN + 32j + li): the 32 lanes of an atomic in the epilogue then
  // cover one 128-byte line of the gradient; their operands are TN conflict-free ds_read_b32.
  const unsigned fb_base = (unsigned)ABYTES + (unsigned)(lh * BNI + wn * WTN + (BVEC ? li * TN : li)) * 4u;
  auto read_frag = [&](int slot_bytes, int kk, float (&av)[TM], float (&bv)[TN]) {
    lds_read_vec<TM>(lds + slot_bytes + fa_base + kk * 2 * BMO * 4, av);
    if constexpr (BVEC) {
      lds_read_vec<TN>(lds + slot_bytes + fb_base + kk * 2 * BNI * 4, bv);
    } else {
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bv[j] = *reinterpret_cast<const float*>(lds + slot_bytes + fb_base + (kk * 2 * BNI + j * 32) * 4);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int c = 0; c < NSLOT - 1; ++c) { prep_next(); issue(c * SLOTB); }
  prep_next();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // With 4 slots the k-step 0 operands ping-pong between two register sets (step parity): the NEXT chunk's first
  // operands are fetched at the top of a step, so no LDS latency sits between a barrier and the first MFMA of the
  // following chunk.  With 3 slots (the 256-cout tiles: 72 KB per workgroup) the next chunk is published by this step's
  // barrier and its first operands are read right after it.
  float a0[2][TM], b0[2][TN];
  read_frag(0, 0, a0[0], b0[0]);

  auto step = [&](auto slot_tag) {
    constexpr int S = decltype(slot_tag)::value;
    constexpr int S1 = (S + 1) % NSLOT, SL = (S + NSLOT - 1) % NSLOT, P = (NSLOT == 4) ? (S & 1) : 0;
    issue(SL * SLOTB);
    float av[7][TM], bv[7][TN];
#pragma unroll
    for (int kk = 1; kk < 8; ++kk) read_frag(S * SLOTB, kk, av[kk - 1], bv[kk - 1]);
    if constexpr (NSLOT == 4) read_frag(S1 * SLOTB, 0, a0[P ^ 1], b0[P ^ 1]);   // published by the previous barrier
    __builtin_amdgcn_sched_barrier(0);
    prep_next();                             // scalar work for the next step's loads, free to interleave with the MFMAs
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[P][i], b0[P][j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int kk = 1; kk < 8; ++kk)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk - 1][i], bv[kk - 1][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LPW == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else if constexpr (LPW == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if constexpr (LPW == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (NSLOT == 3) read_frag(S1 * SLOTB, 0, a0[0], b0[0]);
  };
  int it = 0;
  if constexpr (NSLOT == 4) {
    for (; it + 4 <= niter; it += 4) {
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
    }
  } else {
    for (; it + 3 <= niter; it += 3) {
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
    }
  }
  if (it < niter) { step(std::integral_constant<int, 0>{}); ++it; }
  if (it < niter) { step(std::integral_constant<int, 1>{}); ++it; }
  if constexpr (NSLOT == 4) {
    if (it < niter) { step(std::integral_constant<int, 2>{}); ++it; }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef SSP_PROBES   // timing probe (wgrad_variant 9): skip the atomic epilogue - wrong results on purpose, probe builds only
  if (p.no_store == 1 && acc[0][0][0] != 12345.678f) return;
#endif

  // Lean form (un-folded tiles whose BMO filters all exist): the tile's gradient rows are addressed through a buffer
  // descriptor on dw[co0][tap][0]; the lane's byte offset is fixed (cins beyond Cin carry an out-of-range offset, the
  // buffer unit drops them), the filter-row steps are SCALAR offsets - one buffer_atomic_add_f32 and one s_add per
  // element, no 64-bit address arithmetic, no exec-mask branch on the lanes the MFMAs of the co-resident waves need.
  const int64_t row_floats = (int64_t)taps * p.Cin;                       // floats between consecutive filters of dw
  if (!FOLD && co0 + BMO <= p.Cout && (int64_t)BMO * row_floats * 4 < (1ll << 31) && p.no_store != 2) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(dw_b + ((int64_t)co0 * taps + tap) * p.Cin), 0, (int)(BMO * row_floats * 4), 0x00020000);
    const int row4 = (int)row_floats * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = wn * WTN + (BVEC ? li * TN + j : j * 32 + li);
      const int ci = ci0 + col;
      const unsigned voff = (ci < p.Cin) ? (unsigned)(((wm * WTM + 4 * lh * TM) * (int)row_floats + ci) * 4) : SSP_OOB;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int so = i * row4;                      // filter row of element r: ((r&3) + 8*(r>>2)) * TM + i
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          asm volatile("" : "+s"(so));          // one live SGPR, one s_add per element (see conv_igemm_common.h)
          const float v = acc[i][j][r];
          if (p.store) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, so, 0);
          else (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rs, voff, so, 0);
          so += (((r & 3) == 3) ? 5 : 1) * TM * row4;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = wn * WTN + (BVEC ? li * TN + j : j * 32 + li);
      const int tl = FOLD ? tap + col / p.Cin : tap;
      const int ci = FOLD ? col % p.Cin : ci0 + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * WTM + ((r & 3) + 8 * (r >> 2) + 4 * lh) * TM + i;
        if (co < p.Cout && ci < p.Cin && tl < taps) {
          float* dst = dw_b + ((int64_t)co * taps + tl) * p.Cin + ci;
          if (p.store) *dst = acc[i][j][r];
          else atomicAdd(dst, acc[i][j][r]);
        }
      }
    }
#endif
}

__global__ void __launch_bounds__(256) wgrad_zero_kernel(float4* __restrict__ p, size_t n16) {
  const float4 z = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = z;
}

template <int BMO, int BNI, int NSLOT = 4, bool FOLD = false, bool BVEC = false>
static int launch_wgrad_dma(WgradArgs a, hipStream_t stream) {
  constexpr int RA = 16;
  a.ntile_co = ssp_cdiv(a.Cout, BMO);
  a.ntile_ci = ssp_cdiv(a.Cin, BNI);
  a.fold = FOLD ? BNI / a.Cin : 1;
  a.no_store = ssp_option(SSP_OPT_WGRAD_VARIANT) == 11 ? 2 : 0;   // 11: generic epilogue (A/B, same results)
#ifdef SSP_PROBES
  if (ssp_option(SSP_OPT_WGRAD_VARIANT) == 9) a.no_store = 1;
#endif
  const int nbatch = a.batch > 1 ? a.batch : 1;
  const int64_t tiles = (int64_t)a.ntile_co * a.ntile_ci * ssp_cdiv(a.R * a.R, a.fold) * nbatch;   // all problems of the launch
  const int lds_bytes = NSLOT * RA * (BMO + BNI) * 4;
  auto kern = conv_wgrad_dma_kernel<BMO, BNI, NSLOT, FOLD, BVEC>;
  static SspKernelCache cache;   // per instantiation, per device
  int slots = 0;
  if (int rc = ssp_kernel_prepare((const void*)kern, lds_bytes, 256, &cache, &slots, "conv_wgrad_dma")) return rc;
  // Split over pixels.  Every workgroup ends with BMO x BNI atomics onto its filter tile, and all workgroups of a tile
  // hit the same lines: on the layers with few tiles (layers 4-16: 9-72 tiles) the atomic traffic is 5-12 % of the
  // launch, so the split is the SMALLEST one that fills whole resident waves of workgroups (>= 93 % of the last wave),
  // from one wave up to five; only when no split gets there the best-filling one wins.  >= 8 chunks per workgroup.
  const int variant = ssp_option(SSP_OPT_WGRAD_VARIANT);
  const int64_t max_split = (a.M + RA * 8 - 1) / (RA * 8);
  // (order 2, below: the split also has to deal its units evenly to the 8 XCDs - a wider search range)
  // measured (profiles/r03_convbench_wgrad.txt): no gain over the plain order on layers 18 - 29 (117.6 / 124.3 / 131.3 TF against
  // 120.9 / 125.1 / 129.9): their operands (44 MB each) live in the 256 MB Infinity Cache and the kernel is not bound by
  // that traffic - kept as an experiment switch (wgrad_variant 20), off by default
  const bool order2 = variant == 20 && nbatch == 1 && !FOLD && tiles > 64 && max_split >= 16 && a.R * a.R * a.ntile_ci <= 128;
  int64_t lo = ((int64_t)slots + tiles - 1) / tiles, hi = ((order2 ? 8 : 5) * (int64_t)slots) / tiles;
  if (tiles >= (int64_t)(0.93 * slots)) lo = 1;        // the tiles alone (almost) fill a wave
  if (variant == 4) lo = (2 * (int64_t)slots + tiles - 1) / tiles;   // experiment: at least two waves (the old rule)
  if (lo < 1) lo = 1;
  if (hi < lo) hi = lo;
  if (lo > max_split) lo = max_split;
  if (hi > max_split) hi = max_split;
  // XCD-aware order (see the kernel) when the workgroups of one pixel range fit one XCD's resident set (64); the split
  // is then a multiple of 8 so that every XCD gets the same number of pixel ranges
  a.xcd_order = (variant != 10 && tiles <= 64 && max_split >= 16 && nbatch == 1) ? 1 : 0;
  // More tiles than that (the 13 x 13 layers): units of (pixel range, cout tile) dealt to the XCDs (order 2); the split is
  // chosen so that the units divide evenly over the 8 XCDs (or are many enough for the remainder not to matter)
  const int64_t step = a.xcd_order ? 8 : 1;
  if (a.xcd_order) {
    lo = (lo + 7) / 8 * 8;
    hi = hi / 8 * 8;
    if (hi < lo) hi = lo;
    if (lo > max_split) { a.xcd_order = 0; lo = max_split; hi = max_split; }
  }
  int64_t nsplit = lo;
  double best = -1.0;
  for (int64_t sp = lo; sp <= hi; sp += (a.xcd_order ? step : 1)) {
    const double waves = (double)(tiles * sp) / slots;
    double eff = waves / (double)((tiles * sp + slots - 1) / slots);
    if (order2) {      // XCDs holding one unit more than the others finish last
      const int64_t units = sp * a.ntile_co;
      eff *= ((double)units / 8.0) / (double)((units + 7) / 8);
    }
    if (eff > best + 1e-3) { best = eff; nsplit = sp; }
    if (eff >= 0.93) break;
  }
  const int forced_split = ssp_option(SSP_OPT_WGRAD_SPLIT);      // experiments: tools/conv_bench.py --opt wgrad_split=N
  if (forced_split > 0 && forced_split <= max_split && !(a.xcd_order == 1 && forced_split % 8)) nsplit = forced_split;
  int64_t chunk = (a.M + nsplit - 1) / nsplit;
  chunk = (chunk + RA - 1) / RA * RA;
  nsplit = (a.M + chunk - 1) / chunk;
  a.nsplit = (int)nsplit;
  a.chunk_m = (int)chunk;
  if (a.zero_ptr != nullptr) {
    a.store = nsplit == 1 ? 1 : 0;
    // (a float4 grid-stride store kernel, not hipMemsetAsync: the runtime's fill kernel moved the 151 - 189 MB of a 13 x 13
    // layer's transform-domain gradient at 0.8 TB/s - 0.19 ms per launch, three of them per step on the filter-gradient
    // stream, profiles/r05_timeline_launches.txt - where a plain store loop runs at the HBM write rate)
    if (!a.store) {
      const size_t n16 = a.zero_bytes / 16;
      if (n16 > 0) {
        const unsigned blocks = (unsigned)((n16 + 1023) / 1024 < 16384 ? (n16 + 1023) / 1024 : 16384);
        hipLaunchKernelGGL(wgrad_zero_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<float4*>(a.zero_ptr), n16);
        SSP_CHECK_LAUNCH("conv_wgrad_dma(zero)");
      }
      if (a.zero_bytes % 16) {
        if (hipMemsetAsync((char*)a.zero_ptr + n16 * 16, 0, a.zero_bytes % 16, stream) != hipSuccess) {
          ssp_set_error("conv_wgrad_dma: hipMemsetAsync failed");
          return SSP_ERR_HIP;
        }
      }
    }
  }
  int64_t nwg = a.xcd_order ? tiles * ((nsplit + 7) / 8 * 8) : tiles * nsplit;
  if (order2) {
    a.xcd_order = 2;
    const int64_t units = nsplit * a.ntile_co, gu = (int64_t)ssp_cdiv(a.R * a.R, a.fold) * a.ntile_ci;
    nwg = 8 * gu * ((units + 7) / 8);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(nwg / nbatch), nbatch), dim3(256), lds_bytes, stream, a);
  SSP_CHECK_LAUNCH("conv_wgrad_dma");
  return SSP_OK;
}

// Batched launches (the P planes of a Winograd filter gradient: P x few tiles, a short pixel range each): the pixel split
// the launcher would pick for a tile height (the smallest one filling >= 93 % of whole resident waves, launch_wgrad_dma's
// rule; `slots` = the resident workgroups of that instantiation: 512 for 256 x 128 tiles, 768 for 128 x 128).  The 256-cout
// tiles are the better kernel, but where they need a 3- to 5-way split (13 x 13 layers at F(4x4): 1024 rows per plane cut
// into 13-chunk pieces, each ending in a tile of atomics) and the 128-cout tiles none or two, the latter win.
static int wgrad_est_split(int64_t M, int batch, int Cout, int Cin, int bmo, int slots) {
  const int64_t tiles = (int64_t)ssp_cdiv(Cout, bmo) * ssp_cdiv(Cin, 128) * (batch > 1 ? batch : 1);
  const int64_t max_split = (M + 16 * 8 - 1) / (16 * 8);
  int64_t lo = (slots + tiles - 1) / tiles, hi = (5 * (int64_t)slots) / tiles;
  if (tiles >= (int64_t)(0.93 * slots)) lo = 1;
  if (lo < 1) lo = 1;
  if (hi < lo) hi = lo;
  if (lo > max_split) lo = max_split;
  if (hi > max_split) hi = max_split;
  int64_t best_sp = lo;
  double best = -1.0;
  for (int64_t sp = lo; sp <= hi; ++sp) {
    const double eff = ((double)(tiles * sp) / slots) / (double)((tiles * sp + slots - 1) / slots);
    if (eff > best + 1e-3) { best = eff; best_sp = sp; }
    if (eff >= 0.93) break;
  }
  return (int)best_sp;
}

// returns 1 when the shape is handled here (launched), 0 when the caller should use conv_wgrad.hip, < 0 on error
int ssp_conv_wgrad_dma_try(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int lddy,
                           int ldx, int R, hipStream_t stream, int batch, int64_t batch_dy, int64_t batch_x,
                           int64_t batch_dw, int overwrite) {
  if (Cout < 64 || (Cin < 64 && !(Cin == 32 && R == 3))) return 0;   // Cin 32: two taps fold into one 64-column tile
  if (W < 8) return 0;   // the per-lane pixel walker advances 16 pixels with at most two row wraps
  const int64_t M = (int64_t)B * H * W;
  // 32-bit lane offsets: 16 staged rows of the widest operand, and the whole dY range of a workgroup
  if ((int64_t)16 * lddy * 4 >= (1ll << 31) || (int64_t)(2 * W + 18) * ldx * 4 >= (1ll << 31)) return 0;
  WgradArgs a;
  a.dy = dy; a.x = x; a.dw = dw;
  a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.lddy = lddy; a.ldx = ldx; a.R = R; a.M = (int)M;
  a.batch = batch; a.batch_dy = batch_dy; a.batch_x = batch_x; a.batch_dw = batch_dw;
  a.store = 0;
  a.zero_ptr = nullptr;
  a.zero_bytes = 0;
  if (overwrite) {      // dw = result (not +=): the launcher stores when it does not split, zeroes first when it does
    a.zero_ptr = dw;
    a.zero_bytes = (size_t)(batch > 1 ? batch : 1) * (size_t)(batch > 1 ? batch_dw : (int64_t)Cout * R * R * Cin) * 4;
  }
  int rc;
  const int wv = ssp_option(SSP_OPT_WGRAD_VARIANT);
  if (Cin == 32) rc = (Cout >= 128) ? launch_wgrad_dma<128, 64, 4, true>(a, stream) : launch_wgrad_dma<64, 64, 4, true>(a, stream);
  // 256-cout tiles (128x64 per wave) on a 3-slot ring, two workgroups per CU: fewer LDS reads, DMA pieces and border
  // walks per MFMA than 128x128, and - what decides it inside the training step, where the data-gradient kernel runs
  // concurrently on the other stream - two fat workgroups per CU leave that kernel room (whole-step A/B on one box:
  // 50.8 ms against 52.3 ms for 128x128 tiles at three per CU, although the stand-alone launch times are equal)
  else if (Cout >= 256 && Cin >= 128 && wv != 8 && !(batch > 1 && wgrad_est_split(M, batch, Cout, Cin, 128, 768) < wgrad_est_split(M, batch, Cout, Cin, 256, 512)))
    rc = launch_wgrad_dma<256, 128, 3>(a, stream);
  else if (Cout >= 128 && Cin >= 128 && wv == 6) rc = launch_wgrad_dma<128, 128, 3, false, true>(a, stream);   // experiment
  else if (Cout >= 128 && Cin >= 128 && wv == 3) rc = launch_wgrad_dma<128, 128, 4>(a, stream);           // experiment
  // 3-slot ring: 48 KB of LDS, three workgroups per CU (measured +3..6 % over the 4-slot ring at two per CU)
  else if (Cout >= 128 && Cin >= 128) rc = launch_wgrad_dma<128, 128, 3>(a, stream);
  else if (Cout >= 128) rc = launch_wgrad_dma<128, 64>(a, stream);
  else if (Cin >= 128) rc = launch_wgrad_dma<64, 128>(a, stream);
  else rc = launch_wgrad_dma<64, 64>(a, stream);
  return rc == SSP_OK ? 1 : rc;
}
