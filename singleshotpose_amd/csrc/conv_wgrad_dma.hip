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
  int fold;   // filter taps per cin tile: 1, or BNI / Cin when Cin < BNI (thin layers: two taps of 32 cins share a tile)
};

#define SSP_OOB 0x80000000u

template <int BMO, int BNI, int NSLOT = 4>
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
  int bid = blockIdx.x;
  const int split = bid % p.nsplit; bid /= p.nsplit;
  const int tap = (bid % tap_groups) * p.fold; bid /= tap_groups;   // first tap of this workgroup's tile
  const int tile_ci = bid % p.ntile_ci;
  const int tile_co = bid / p.ntile_ci;
  const int co0 = tile_co * BMO, ci0 = tile_ci * BNI;
  const int pad = p.R >> 1;

  const int m_begin = split * p.chunk_m;
  const int m_end = min(p.M, m_begin + p.chunk_m);
  const int niter = (m_end - m_begin + RA - 1) / RA;
  if (niter <= 0) return;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int li = lane & 31, lh = lane >> 5;

  // ---- loader lanes: piece g (1 KiB) of a tile = bytes [g*1024, g*1024+1024) of the row-major [16][C] image ----
  unsigned a_voff[APW], b_off[BPW];
  int b_row[BPW], b_x[BPW], b_y[BPW], b_dx[BPW], b_dy[BPW];
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
    // tile column -> (tap, cin): with fold > 1 the tile spans `fold` consecutive taps of all Cin channels; the tap's
    // pixel shift goes into the lane offset (relative to the pixel-row base of the descriptor)
    const int tl = (p.fold > 1) ? tap + col / p.Cin : tap;
    const int cc = (p.fold > 1) ? col % p.Cin : ci0 + col;
    b_dy[j] = tl / p.R - pad;
    b_dx[j] = tl % p.R - pad;
    b_off[j] = (cc < p.Cin && tl < taps) ? (unsigned)(((row + b_dy[j] * p.W + b_dx[j] + p.W + 1) * p.ldx + cc) * 4) : SSP_OOB;
    const int m = m_begin + row;
    b_x[j] = m % p.W;
    b_y[j] = (m / p.W) % p.H;
  }

  int ld_m = m_begin;   // first pixel of the chunk the loader stages next
  auto issue_loads = [&](int slot_bytes) {
    const int left = m_end - ld_m;                       // pixel rows still inside this workgroup's range
    const int64_t abase = (int64_t)ld_m * p.lddy;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.dy + abase), 0, left > 0 ? (int)min((int64_t)left * p.lddy * 4, (int64_t)0x7fffffff) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + ((int64_t)ld_m - p.W - 1) * p.ldx), 0, (int)SSP_OOB, 0x00020000);   // base = pixel (y-1, x-1)
#pragma unroll
    for (int j = 0; j < APW; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(lds + slot_bytes + (wid + 4 * j) * 1024),
                                               16, a_voff[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
      const int yy = b_y[j] + b_dy[j], xx = b_x[j] + b_dx[j];
      const bool ok = (b_row[j] < left) && ((unsigned)yy < (unsigned)p.H) && ((unsigned)xx < (unsigned)p.W);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void*)(lds + slot_bytes + ABYTES + (wid + 4 * j) * 1024),
                                               16, ok ? b_off[j] : SSP_OOB, 0, 0, 0);
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
    }
    ld_m += RA;
  };

  // ---- operand fetch: k-step kk uses pixel rows 2kk + lh; A[i=cout][k], B[k][j=cin]; conflict-free ds_read_b32 ----
  const unsigned fa_base = (unsigned)(lh * BMO + wm * WTM + li) * 4u;
  const unsigned fb_base = (unsigned)ABYTES + (unsigned)(lh * BNI + wn * WTN + li) * 4u;
  auto read_frag = [&](int slot_bytes, int kk, float (&av)[TM], float (&bv)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
      av[i] = *reinterpret_cast<const float*>(lds + slot_bytes + fa_base + (kk * 2 * BMO + i * 32) * 4);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      bv[j] = *reinterpret_cast<const float*>(lds + slot_bytes + fb_base + (kk * 2 * BNI + j * 32) * 4);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int c = 0; c < NSLOT - 1; ++c) issue_loads(c * SLOTB);
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
    issue_loads(SL * SLOTB);
    float av[7][TM], bv[7][TN];
#pragma unroll
    for (int kk = 1; kk < 8; ++kk) read_frag(S * SLOTB, kk, av[kk - 1], bv[kk - 1]);
    if constexpr (NSLOT == 4) read_frag(S1 * SLOTB, 0, a0[P ^ 1], b0[P ^ 1]);   // published by the previous barrier
    __builtin_amdgcn_sched_barrier(0);
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

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = wn * WTN + j * 32 + li;
      const int tl = (p.fold > 1) ? tap + col / p.Cin : tap;
      const int ci = (p.fold > 1) ? col % p.Cin : ci0 + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < p.Cout && ci < p.Cin && tl < taps) atomicAdd(p.dw + ((int64_t)co * taps + tl) * p.Cin + ci, acc[i][j][r]);
      }
    }
#endif
}

template <int BMO, int BNI, int NSLOT = 4>
static int launch_wgrad_dma(WgradArgs a, hipStream_t stream) {
  constexpr int RA = 16;
  a.ntile_co = ssp_cdiv(a.Cout, BMO);
  a.ntile_ci = ssp_cdiv(a.Cin, BNI);
  a.fold = (a.Cin < BNI && BNI % a.Cin == 0 && a.R > 1) ? BNI / a.Cin : 1;
  const int64_t tiles = (int64_t)a.ntile_co * a.ntile_ci * ssp_cdiv(a.R * a.R, a.fold);
  const int lds_bytes = NSLOT * RA * (BMO + BNI) * 4;
  auto kern = conv_wgrad_dma_kernel<BMO, BNI, NSLOT>;
  static int configured = 0;
  static int slots = 0;
  if (lds_bytes > configured) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
      ssp_set_error("conv_wgrad_dma: cannot reserve %d bytes of LDS", lds_bytes);
      return SSP_ERR_HIP;
    }
    configured = lds_bytes;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 256, lds_bytes) != hipSuccess || per_cu < 1)
      per_cu = 2;
    slots = per_cu * 256;
  }
  // whole resident waves of workgroups, 2..5 of them, >= 8 chunks per workgroup (as conv_wgrad.hip)
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
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * nsplit)), dim3(256), lds_bytes, stream, a);
  SSP_CHECK_LAUNCH("conv_wgrad_dma");
  return SSP_OK;
}

// returns 1 when the shape is handled here (launched), 0 when the caller should use conv_wgrad.hip, < 0 on error
int ssp_conv_wgrad_dma_try(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int lddy,
                           int ldx, int R, hipStream_t stream) {
  if (Cout < 64 || (Cin < 64 && !(Cin == 32 && R == 3))) return 0;   // Cin 32: two taps fold into one 64-column tile
  if (W < 8) return 0;   // the per-lane pixel walker advances 16 pixels with at most two row wraps
  const int64_t M = (int64_t)B * H * W;
  // 32-bit lane offsets: 16 staged rows of the widest operand, and the whole dY range of a workgroup
  if ((int64_t)16 * lddy * 4 >= (1ll << 31) || (int64_t)(2 * W + 18) * ldx * 4 >= (1ll << 31)) return 0;
  WgradArgs a;
  a.dy = dy; a.x = x; a.dw = dw;
  a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.lddy = lddy; a.ldx = ldx; a.R = R; a.M = (int)M;
  int rc;
  // 256-cout tiles (128x64 per wave): 25-50 % fewer LDS reads, DMA pieces and border walks per MFMA than 128x128
  if (Cout >= 256 && Cin >= 128 && ssp_option(SSP_OPT_WGRAD_VARIANT) != 3) rc = launch_wgrad_dma<256, 128, 3>(a, stream);
  else if (Cout >= 128 && Cin >= 128) rc = launch_wgrad_dma<128, 128>(a, stream);
  else if (Cout >= 128) rc = launch_wgrad_dma<128, 64>(a, stream);
  else if (Cin >= 128) rc = launch_wgrad_dma<64, 128>(a, stream);
  else rc = launch_wgrad_dma<64, 64>(a, stream);
  return rc == SSP_OK ? 1 : rc;
}
