// Implicit-GEMM convolution, LDS-direct loader variant (the workhorse for Cin % 16 == 0, Cout > 64).
//
// Same contraction, tiles, MFMA and epilogue as conv_igemm.hip; what changes is how the operands reach the LDS.
// On gfx950 the fp32 MFMA executes on the SIMD's vector lanes, so every VALU instruction in the K loop (address
// arithmetic, v_cndmask zero-fill, register->LDS staging) steals matrix-core issue time one for one (measured:
// 64 extra v_add per chunk = -11 %).  This kernel therefore moves ALL per-chunk loader work off the VALU:
//
//   * operands are fetched with `buffer_load_dwordx4 ... lds` (global -> LDS DMA, no VGPR staging, no ds_write);
//   * the per-lane byte offset (`voffset`) is fixed for a whole filter tap and already encodes the image-border test:
//     lanes whose tap falls outside the image (and rows beyond M / filters beyond Cout) carry an out-of-range offset,
//     for which the buffer unit returns zeros - the zero padding costs no instruction;
//   * the channel walk inside a tap is a scalar `soffset` increment; only a tap change (every Cin/16 chunks)
//     recomputes the lane offsets;
//   * an LDS-DMA image is lane-linear (1 KiB per wave-instruction, no row padding possible), so bank conflicts are
//     removed by an XOR swizzle of the 16-byte chunk index with (row>>2)&3, applied on the SOURCE address of the load
//     and on the fragment reads (cdna_hip_programming.md, rule 21);
//   * 4-slot LDS ring, loads run 3 chunks ahead; one `s_waitcnt vmcnt(n)` + one raw `s_barrier` per chunk; the K loop
//     is unrolled x4 so every LDS address is a register + immediate.
#include "conv_igemm_common.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define SSP_OOB 0x80000000u   // >= num_records of the descriptors below: the load returns zeros

// s_waitcnt vmcnt(N) with a compile-time N (the immediate must be a literal in the asm string)
template <int N>
__device__ __forceinline__ void ssp_wait_vmcnt() {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if constexpr (N == 25) asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
  else static_assert(N == 10, "add the literal for this count");
#endif
}

// One accumulation group of the K loop: FLN unrolled ring revolutions of NS chunks; the very first chunk is FRESH
template <int FLN, int NS, typename Step, int... Is>
__device__ __forceinline__ void ssp_sfor_groups_impl(Step& step, std::integer_sequence<int, Is...>) {
  (step(std::integral_constant<int, Is % NS>{}, std::integral_constant<bool, Is == 0>{}), ...);
}
template <int FLN, int NS, typename Step>
__device__ __forceinline__ void ssp_sfor_groups(Step& step) {
  ssp_sfor_groups_impl<FLN, NS>(step, std::make_integer_sequence<int, FLN * NS>{});
}

// PASS (0 = forward, 1 = data gradient) does not change the code: it gives the two uses distinct kernel names, so a
// profile can tell the forward launches (exclusive on the GPU) from the dgrad launches (which overlap wgrad on a
// second stream in Plan.backward).
//
// WM x WN = arrangement of the 4 waves over the tile: 2 x 2 for the square tiles, 4 x 1 for the 256 x 32 tile that
// serves GEMMs with 32 output columns (the data gradient of layer 2: 64 -> 32 channels at 208 x 208).
//
// NSLOT = LDS ring depth.  3 and 4 are the throughput forms (2-3 workgroups per CU cover each other's memory latency).
// 8 is the LATENCY form for grids too small to give a CU more than one workgroup (batch-1 inference at 672 x 672: 100-450
// tiles): every workgroup walks K in step, so each chunk is a first touch served at fabric / HBM latency (~1-2 us), and
// with one workgroup per CU nothing else hides it.  A chunk has to land only when its slot is about to be read, so the
// per-step wait leaves NSLOT-3 chunks outstanding: 7 chunks run ahead of the MFMAs instead of 3 (the 4-slot ring's
// wait - everything but the newest chunk - is the NSLOT = 4 case of the same rule).
//
// FL > 0 = CHUNKED ACCUMULATION: the fp32 MFMA adds every product into one running sum per output, a chain of K
// roundings whose error grows like sqrt(K) relative to the partial sums - and in the batched GEMMs of a Winograd plan
// those partial sums are 3-4 x the size of the final output (the inverse transform cancels them), which is where F(4x4)'s
// error came from (tools/wino_error_budget.py: rounding V and U to fp32 costs 5e-8 of the output range, the accumulation
// chain 4e-7 at K = 256 and 9e-7 at K = 1024).  With FL set the K loop runs in groups of FL ring revolutions (8 chunks =
// K 128 on the 4-slot ring, 6 = K 96 on the 3-slot ring): the first MFMA of a group starts from the inline constant 0
// instead of the accumulator, and the finished group sum is added to a second register set - two short chains (K 128,
// then K / 128 group sums) instead of one long one: 2.6 x less error on a K = 1024 F(4x4) plane, 6.6 x on a direct 3x3
// layer (same simulation), for TM * TN * 16 VALU adds per group (~1.5 % of the group's MFMA time).
template <int BM, int BN, int PASS, int NSLOT = 4, int WM = 2, int WN = 2, int FL = 0>
__global__ void __launch_bounds__(256, NSLOT >= 6 ? 1 : ((FL > 0 && BM + BN >= 256) ? 2 : ((NSLOT == 3 || BM + BN < 256) ? 3 : 2)))
conv_igemm_dma_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource builtins and gfx950 asm exist in the device pass only
  constexpr int BK = 16, NT = 256;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  constexpr int ROWB = BK * 4;                  // bytes per tile row
  constexpr int SLOTB = (BM + BN) * ROWB;       // bytes per ring slot: A rows then B rows
  constexpr int APW = BM / 64, BPW = (BN + 63) / 64;   // 1-KiB wave-instructions per wave per chunk (16 rows each)
  constexpr int LPW = APW + BPW;
  // a B tile narrower than 64 rows has fewer 1-KiB pieces than waves: the spare waves still issue their (all
  // out-of-range, zero-filled) load so that every wave's vmcnt bookkeeping is identical; it lands in a scratch KiB
  // behind the ring
  constexpr bool B_SPARE = (BN % 64) != 0;
  static_assert(BM % 64 == 0 && BN % 32 == 0, "tile rows are dealt to the 4 waves in groups of 16");

  extern __shared__ __attribute__((aligned(16))) float smem[];   // NSLOT * SLOTB bytes
  char* const lds = reinterpret_cast<char*>(smem);

  const int ntiles = p.ntile_m * p.ntile_n;
  int split, lid;
  bool partial;
  if (p.tail_ks > 0) {
    // hybrid: workgroups [0, tail_begin) = whole un-split tiles; the rest = the tail tiles x tail_ks K ranges.  The XCD
    // remap permutes each group on its own so that the short tail workgroups are dispatched last, on every XCD.
    const int ntail = ntiles - p.tail_begin;
    if ((int)blockIdx.x < p.tail_begin) {
      lid = p.xcd_remap ? ssp_xcd_remap(blockIdx.x, p.tail_begin) : (int)blockIdx.x;
      split = 0;
      partial = false;
    } else {
      const int nt = ntail * p.tail_ks;
      const int t0 = (int)blockIdx.x - p.tail_begin;
      const int t = p.xcd_remap ? ssp_xcd_remap(t0, nt) : t0;
      split = t / ntail;
      lid = p.tail_begin + (t - split * ntail);
      partial = true;
    }
  } else {
    const int nwg = ntiles * p.ksplit;
    const int lid0 = p.xcd_remap ? ssp_xcd_remap(blockIdx.x, nwg) : (int)blockIdx.x;
    split = lid0 / ntiles;
    lid = lid0 - split * ntiles;
    partial = p.ksplit > 1;
  }
  // Tile order.  Row-major (tile_n fastest): the workgroups resident on an XCD share activation rows and together stream
  // the WHOLE filter matrix once per group of tile rows - fine while it fits the XCD's 4 MB L2.  For the deep layers
  // (filters of 5-47 MB) the order is column-major inside the un-split part and inside the tail part: an XCD's resident
  // workgroups then share ONE 128-filter slab and walk it in step, and the (smaller) activation rows are what gets
  // re-read (measured HBM traffic per launch of the 13x13 layers: 1.4 GB row-major).
  int tile_n, tile_m;
  if (p.col_major) {
    const int tm1 = p.tail_begin / p.ntile_n;                  // tile rows of the un-split part (all of them: no tail)
    if (lid < p.tail_begin || p.tail_ks == 0) {
      const int rows = (p.tail_ks > 0) ? tm1 : p.ntile_m;
      tile_m = lid % rows;
      tile_n = lid / rows;
    } else {
      const int rows = p.ntile_m - tm1, t = lid - p.tail_begin;
      tile_m = tm1 + t % rows;
      tile_n = t / rows;
    }
  } else {
    tile_n = lid % p.ntile_n;
    tile_m = lid / p.ntile_n;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA destinations stay in SGPRs
  const int wm = wid / WN, wn = wid % WN;
  const int li = lane & 31, lh = lane >> 5;

  const int pad = p.R >> 1;
  const int K = p.R * p.R * p.Cin;
  const int cpt = p.Cin / BK;
  const int niter_all = p.R * p.R * cpt;
  const int it_ps = (p.tail_ks > 0) ? (partial ? p.tail_it_per_split : niter_all) : p.it_per_split;
  const int it_begin = split * it_ps;
  const int niter = min(niter_all, it_begin + it_ps) - it_begin;

  // ---- buffer descriptors: A starts (W+1) pixels before the tile so every tap offset is non-negative (a 1x1 filter has
  // no negative tap: no halo - the batched GEMMs of conv_wino.hip run this kernel with W = their row count) ----
  const int halo = (p.R == 1) ? 0 : p.W + 1;
  const int64_t bz = (p.batch > 1) ? (int64_t)blockIdx.y : 0;      // batched launch: problem index (scalar)
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.in + bz * p.batch_in + ((int64_t)m0 - halo) * p.ldin), 0, (int)SSP_OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.wt + bz * p.batch_wt + (int64_t)n0 * K), 0, (int)SSP_OOB, 0x00020000);

  // ---- loader lanes: a wave-instruction covers 16 tile rows; lane -> (row = lane>>2, physical chunk = lane&3) ----
  const int lrow = lane >> 2, lch = lane & 3;
  int a_y[APW], a_x[APW];
  unsigned a_row_off[APW], a_voff[APW], b_voff[BPW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int row = (wid + 4 * j) * 16 + lrow;
    const int m = m0 + row;
    const int chunk = lch ^ ((row >> 2) & 3);              // logical 16-byte chunk stored at physical position lch
    if (m < p.M) {
      const unsigned q = ssp_div((unsigned)m, p.divW);       // m / W
      a_x[j] = m - (int)q * p.W;
      a_y[j] = (int)(q - ssp_div(q, p.divH) * (unsigned)p.H);   // (m / W) % H
    } else {
      a_x[j] = 0;
      a_y[j] = -(1 << 20);                                 // never inside the image
    }
    a_row_off[j] = (unsigned)((row + halo) * p.ldin + chunk * 4) * 4u;
  }
#pragma unroll
  for (int j = 0; j < BPW; ++j) {
    const int row = (wid + 4 * j) * 16 + lrow;
    const int chunk = lch ^ ((row >> 2) & 3);
    b_voff[j] = (row < BN && n0 + row < p.Cout) ? (unsigned)(row * K + chunk * 4) * 4u : SSP_OOB;
  }
  // LDS byte offset (inside a slot) of this wave's B pieces; spare waves point at the scratch KiB (slot-independent)
  const bool b_spare = B_SPARE && (wid * 16 >= BN);

  // ---- K-chunk walker (scalar): tap (dy,dx), channel offset, filter column; lane offsets recomputed per tap ----
  int ld_tap = it_begin / cpt;
  int ld_c0 = (it_begin - ld_tap * cpt) * BK;
  int ld_koff = it_begin * BK;
  auto set_tap = [&]() {
    const int dy = ld_tap / p.R - pad, dx = ld_tap % p.R - pad;
    const bool tap_ok = ld_tap < p.R * p.R;
    const int shift = (dy * p.W + dx) * p.ldin * 4;
#pragma unroll
    for (int j = 0; j < APW; ++j) {
      const int yy = a_y[j] + dy, xx = a_x[j] + dx;
      const bool ok = tap_ok && ((unsigned)yy < (unsigned)p.H) && ((unsigned)xx < (unsigned)p.W);
      a_voff[j] = ok ? (unsigned)((int)a_row_off[j] + shift) : SSP_OOB;
    }
  };
  set_tap();
  auto issue_loads = [&](int slot_bytes) {
#pragma unroll
    for (int j = 0; j < APW; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(lds + slot_bytes + (wid + 4 * j) * 1024),
                                               16, a_voff[j], ld_c0 * 4, 0, 0);
#pragma unroll
    for (int j = 0; j < BPW; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void*)(lds + (b_spare ? NSLOT * SLOTB : slot_bytes + BM * ROWB + (wid + 4 * j) * 1024)),
                                               16, b_voff[j], ld_koff * 4, 0, 0);
    // advance; past the last chunk the filter column is clamped and the tap index runs out of range (all lanes OOB)
    ld_koff = min(ld_koff + BK, K - BK);
    ld_c0 += BK;
    if (ld_c0 == p.Cin) {
      ld_c0 = 0;
      ++ld_tap;
      set_tap();
    }
  };

  // ---- fragment addressing: lane (i,h) reads logical chunk 2q+h of its row, stored at chunk ^ ((row>>2)&3) ----
  unsigned fa_off[TM][2], fb_off[TN][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const unsigned ch = (unsigned)((2 * q + lh) ^ ((li >> 2) & 3)) * 16u;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa_off[i][q] = (unsigned)(wm * WTM + i * 32 + li) * ROWB + ch;
#pragma unroll
    for (int j = 0; j < TN; ++j) fb_off[j][q] = (unsigned)(BM + wn * WTN + j * 32 + li) * ROWB + ch;
  }
  auto read_frag = [&](int slot_bytes, int q, f32x4 (&fa)[TM], f32x4 (&fb)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(lds + slot_bytes + fa_off[i][q]);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(lds + slot_bytes + fb_off[j][q]);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // FRESH: the first k-step of an accumulation group takes C = 0 (an inline constant of the MFMA: no register is cleared)
  auto mma = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN], auto fresh_tag) {
    constexpr bool FRESH = decltype(fresh_tag)::value;
    constexpr f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (FRESH && e == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], zero16, 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
  };

  // ---- prologue: the first NSLOT-1 chunks in flight, wait for all, publish ----
#pragma unroll
  for (int c = 0; c < NSLOT - 1; ++c) issue_loads(c * SLOTB);
  if constexpr (NSLOT >= 6) ssp_wait_vmcnt<(NSLOT - 3) * LPW>();      // chunks 0 and 1 are in; the rest keeps flying
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // fragment registers: with 4 slots the q = 0 fragments ping-pong between two sets (step parity) so that the NEXT
  // chunk's first fragments are fetched at the top of a step, a full chunk of MFMAs before they are needed.  With 3
  // slots (3 workgroups per CU) the next chunk is only published by this step's barrier; the third wave per SIMD
  // covers that LDS latency instead.
  f32x4 fa0[2][TM], fb0[2][TN], fa1[TM], fb1[TN];
  read_frag(0, 0, fa0[0], fb0[0]);
  // Enter the K loop with nothing but these in-order LDS reads on the compiler's lgkm scoreboard: the kernel-argument
  // scalar loads of the prologue return out of order, and a loop header that may still have one pending makes the
  // compiler wait lgkmcnt(0) before the first MFMA of every unrolled iteration - i.e. for the fragment reads it has
  // just issued - instead of the counted lgkmcnt(n) the other steps get.  (vmcnt 63 / expcnt 7 = no wait on those.)
  __builtin_amdgcn_s_waitcnt(0xC07F);

  // one K chunk; S = ring slot of the chunk being multiplied (compile time: LDS addresses become immediates)
  auto step = [&](auto slot_tag, auto fresh_tag) {
    constexpr int S = decltype(slot_tag)::value;
    constexpr int S1 = (S + 1) % NSLOT, SL = (S + NSLOT - 1) % NSLOT;
    constexpr int P = (NSLOT != 3) ? (S & 1) : 0;
    issue_loads(SL * SLOTB);                 // chunk it+NSLOT-1 (slot last read one barrier ago)
    read_frag(S * SLOTB, 1, fa1, fb1);
    if constexpr (NSLOT != 3) read_frag(S1 * SLOTB, 0, fa0[P ^ 1], fb0[P ^ 1]);   // published by the previous barrier
    __builtin_amdgcn_sched_barrier(0);       // keep the LDS reads up here (the scheduler would sink them to their use)
    mma(fa0[P], fb0[P], fresh_tag);
    mma(fa1, fb1, std::false_type{});
    __builtin_amdgcn_sched_barrier(0);       // ... and the wait + barrier below the MFMAs (MFMAs are register-only, so
                                             // the scheduler is otherwise free to hoist the barrier above them)
    // the chunk issued one step ago must have landed before it is published; this step's LPW loads stay in flight
    if constexpr (NSLOT >= 6) {
      // deep ring: only chunk it+2 (read at the top of the next step) has to be in; NSLOT-3 chunks stay in flight
      ssp_wait_vmcnt<(NSLOT - 3) * LPW>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (LPW == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
    else if constexpr (LPW == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if constexpr (LPW == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (NSLOT == 3) read_frag(S1 * SLOTB, 0, fa0[0], fb0[0]);
  };
  int it = 0;
  [[maybe_unused]] f32x16 tot[TM][TN];
  if constexpr (FL > 0) {
    static_assert(NSLOT == 3 || NSLOT == 4, "chunked accumulation: throughput forms only");
    constexpr int G = FL * NSLOT;              // chunks per accumulation group
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
    for (; it + G <= niter; it += G) {
      ssp_sfor_groups<FL, NSLOT>(step);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) tot[i][j] += acc[i][j];
    }
    // the remaining chunks (fewer than a group) start from a cleared accumulator and are added below
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  if constexpr (NSLOT == 8) {
    for (; it + 8 <= niter; it += 8) {
      step(std::integral_constant<int, 0>{}, std::false_type{});
      step(std::integral_constant<int, 1>{}, std::false_type{});
      step(std::integral_constant<int, 2>{}, std::false_type{});
      step(std::integral_constant<int, 3>{}, std::false_type{});
      step(std::integral_constant<int, 4>{}, std::false_type{});
      step(std::integral_constant<int, 5>{}, std::false_type{});
      step(std::integral_constant<int, 6>{}, std::false_type{});
      step(std::integral_constant<int, 7>{}, std::false_type{});
    }
  } else if constexpr (NSLOT == 4) {
    for (; it + 4 <= niter; it += 4) {
      step(std::integral_constant<int, 0>{}, std::false_type{});
      step(std::integral_constant<int, 1>{}, std::false_type{});
      step(std::integral_constant<int, 2>{}, std::false_type{});
      step(std::integral_constant<int, 3>{}, std::false_type{});
    }
  } else {
    for (; it + 3 <= niter; it += 3) {
      step(std::integral_constant<int, 0>{}, std::false_type{});
      step(std::integral_constant<int, 1>{}, std::false_type{});
      step(std::integral_constant<int, 2>{}, std::false_type{});
    }
  }
  if (it < niter) { step(std::integral_constant<int, 0>{}, std::false_type{}); ++it; }
  if (it < niter) { step(std::integral_constant<int, 1>{}, std::false_type{}); ++it; }
  if constexpr (NSLOT == 4 || NSLOT == 8) {
    if (it < niter) { step(std::integral_constant<int, 2>{}, std::false_type{}); ++it; }
  }
  if constexpr (NSLOT == 8) {
    if (it < niter) { step(std::integral_constant<int, 3>{}, std::false_type{}); ++it; }
    if (it < niter) { step(std::integral_constant<int, 4>{}, std::false_type{}); ++it; }
    if (it < niter) { step(std::integral_constant<int, 5>{}, std::false_type{}); ++it; }
    if (it < niter) { step(std::integral_constant<int, 6>{}, std::false_type{}); ++it; }
  }

  if constexpr (FL > 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] += tot[i][j];
  }
  // retire the run-ahead DMA before the LDS is reused by the epilogue
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

#ifdef SSP_PROBES   // timing probe (igemm_variant 60): skip the epilogue - wrong results on purpose, probe builds only (make PROBES=1)
  if (p.probe == 1 && acc[0][0][0] != 12345.678f) return;
#endif
  igemm_epilogue<BM, BN, WM, WN, NT>(p, acc, smem, m0, n0, tile_m, split, tid, partial, bz * p.batch_out);
#endif
}

template <int BM, int BN, int PASS, int NSLOT = 4, int WM = 2, int WN = 2, int FL = 0>
static int launch_dma(ConvArgs& a, int tail_ks, hipStream_t stream) {
  a.ntile_m = ssp_cdiv(a.M, BM);
  a.ntile_n = ssp_cdiv(a.Cout, BN);
  const int niter_total = a.R * a.R * (a.Cin / 16);
#ifdef SSP_PROBES
  if (ssp_option(SSP_OPT_IGEMM_VARIANT) == 60) a.probe = 1;
#endif
  a.it_per_split = ssp_cdiv(niter_total, a.ksplit);
  a.ksplit = ssp_cdiv(niter_total, a.it_per_split);
  const int lds_bytes = NSLOT * (BM + BN) * 64 + ((BN % 64) ? 1024 : 0);
  auto kern = conv_igemm_dma_kernel<BM, BN, PASS, NSLOT, WM, WN, FL>;
  static SspKernelCache cache;   // per instantiation, per device
  int slots = 0;                 // workgroups resident on the whole chip at once
  if (int rc = ssp_kernel_prepare((const void*)kern, lds_bytes, 256, &cache, &slots, "conv_igemm_dma")) return rc;
  a.col_major = ((int64_t)a.R * a.R * a.Cin * a.Cout * 4 >= (4ll << 20) && a.ntile_n > 1 &&
                 ssp_option(SSP_OPT_IGEMM_VARIANT) != 80) ? 1 : 0;
  // hybrid launch: as many whole resident waves of un-split tiles as fit, the rest of the tiles split tail_ks ways
  a.tail_ks = 0;
  a.tail_begin = a.ntile_m * a.ntile_n;
  a.ws_row0 = 0;
  a.ws_rows = a.M;
  unsigned nwg = (unsigned)(a.ntile_m * a.ntile_n * a.ksplit);
  if (tail_ks > 1 && a.ksplit == 1 && a.ws != nullptr && a.Cout % 4 == 0 && niter_total / tail_ks >= 8) {
    const int64_t tiles = (int64_t)a.ntile_m * a.ntile_n;
    const int64_t full = tiles / slots;                              // whole waves
    const int tm1 = (int)((full * slots) / a.ntile_n);               // first tile row of the tail
    if (full >= 1 && tm1 < a.ntile_m) {
      a.tail_ks = tail_ks;
      a.tail_it_per_split = ssp_cdiv(niter_total, tail_ks);
      a.tail_ks = ssp_cdiv(niter_total, a.tail_it_per_split);
      a.tail_begin = tm1 * a.ntile_n;
      a.ws_row0 = tm1 * BM;
      a.ws_rows = a.M - a.ws_row0;
      nwg = (unsigned)(a.tail_begin + (a.ntile_m - tm1) * a.ntile_n * a.tail_ks);
    }
  }
  hipLaunchKernelGGL(kern, dim3(nwg, a.batch > 1 ? a.batch : 1), dim3(256), lds_bytes, stream, a);
  SSP_CHECK_LAUNCH("conv_igemm_dma");
  return SSP_OK;
}

// bm in {64, 128} with BN = 128, 128 x 64 tiles for Cout <= 64, 256 x 32 tiles for Cout <= 32.  Preconditions (checked by the caller): Cin % 16 == 0, 16-byte aligned operands,
// ldin % 4 == 0, and every byte offset of a tile (rows + halo) below 2^31.
int ssp_conv_igemm_dma_launch(ConvArgs& a, int bm, int slots, int tail_ks, int is_dgrad, hipStream_t stream) {
  // ring depth: 3 slots (48 KB, 3 workgroups per CU) for un-split 128x128 grids and the 128x64 tiles - the third wave
  // per SIMD covers the barrier / LDS-latency bubbles better than the deeper prefetch does (measured +4..6 % on layers
  // 2-8); split-K launches and 64x128 tiles keep 4 slots.  `slots` (3 / 4) from an explicit plan overrides this.
  // On return a.tail_ks > 0 tells the caller that rows >= a.ws_row0 were left as partials in the workspace.
  bool three = (bm == 128 && a.Cout > 64 && a.ksplit == 1) || a.Cout <= 64;
  if (slots == 3) three = true;
  if (slots == 4) three = false;
  const int tk = tail_ks;
  if (slots == 8 && a.Cout > 64) {      // latency form (explicit plans only): 8-slot ring, one workgroup per CU
    if (is_dgrad) return bm == 64 ? launch_dma<64, 128, 1, 8>(a, tk, stream) : launch_dma<128, 128, 1, 8>(a, tk, stream);
    return bm == 64 ? launch_dma<64, 128, 0, 8>(a, tk, stream) : launch_dma<128, 128, 0, 8>(a, tk, stream);
  }
  if (a.Cout <= 32) return is_dgrad ? launch_dma<256, 32, 1, 4, 4, 1>(a, 0, stream) : launch_dma<256, 32, 0, 4, 4, 1>(a, 0, stream);
  // Chunked accumulation (see the kernel; the "acc_chunk" option, on by default): groups of two ring revolutions - 6 chunks
  // = K 96 on the 3-slot ring, 8 chunks = K 128 on the 4-slot ring - for every K loop of at least 12 chunks (K >= 192: a
  // group and a half); shorter loops run the plain kernel (their whole chain is hardly longer than one group).  Measured
  // (profiles/r05_convbench_acc_chunk.txt, error against float64 on the network's operand statistics): direct 3x3, K =
  // 9216: rms 1.75e-7 -> 4.0e-8 of the output range, F(4x4) planes of K = 1024: 6.1e-7 -> 2.45e-7 (max 9.5e-6 -> 2.4e-6),
  // launch times within +-1.5 %.
  const int chunks = ssp_cdiv(a.R * a.R * (a.Cin / 16), a.ksplit > 1 ? a.ksplit : 1);
  const bool fl = ssp_option(SSP_OPT_ACC_CHUNK) != 0 && chunks >= 12;
  if (fl) {
    if (is_dgrad) {
      if (a.Cout <= 64) return three ? launch_dma<128, 64, 1, 3, 2, 2, 2>(a, tk, stream) : launch_dma<128, 64, 1, 4, 2, 2, 2>(a, tk, stream);
      if (bm == 64) return three ? launch_dma<64, 128, 1, 3, 2, 2, 2>(a, tk, stream) : launch_dma<64, 128, 1, 4, 2, 2, 2>(a, tk, stream);
      return three ? launch_dma<128, 128, 1, 3, 2, 2, 2>(a, tk, stream) : launch_dma<128, 128, 1, 4, 2, 2, 2>(a, tk, stream);
    }
    if (a.Cout <= 64) return three ? launch_dma<128, 64, 0, 3, 2, 2, 2>(a, tk, stream) : launch_dma<128, 64, 0, 4, 2, 2, 2>(a, tk, stream);
    if (bm == 64) return three ? launch_dma<64, 128, 0, 3, 2, 2, 2>(a, tk, stream) : launch_dma<64, 128, 0, 4, 2, 2, 2>(a, tk, stream);
    return three ? launch_dma<128, 128, 0, 3, 2, 2, 2>(a, tk, stream) : launch_dma<128, 128, 0, 4, 2, 2, 2>(a, tk, stream);
  }
  if (is_dgrad) {
    if (a.Cout <= 64) return three ? launch_dma<128, 64, 1, 3>(a, tk, stream) : launch_dma<128, 64, 1>(a, tk, stream);
    if (bm == 64) return three ? launch_dma<64, 128, 1, 3>(a, tk, stream) : launch_dma<64, 128, 1>(a, tk, stream);
    return three ? launch_dma<128, 128, 1, 3>(a, tk, stream) : launch_dma<128, 128, 1>(a, tk, stream);
  }
  if (a.Cout <= 64) return three ? launch_dma<128, 64, 0, 3>(a, tk, stream) : launch_dma<128, 64, 0>(a, tk, stream);
  if (bm == 64) return three ? launch_dma<64, 128, 0, 3>(a, tk, stream) : launch_dma<64, 128, 0>(a, tk, stream);
  return three ? launch_dma<128, 128, 0, 3>(a, tk, stream) : launch_dma<128, 128, 0>(a, tk, stream);
}
