// Winograd F(2x2, 3x3) with the transform domain kept ON THE CHIP (plan codes 7000000 + v, conv_wino.h).
//
// conv_wino.hip evaluates a Winograd plan as three launches around two HBM round trips: V = B^T d B is written and read
// (4 floats per input float at n = 2), M = U (.) V is written and read (4 floats per output float).  On the wide maps of
// the network - 64 -> 128 channels at 104 x 104, 32 -> 64 at 208 x 208 - those round trips cost more than the 16/36
// multiplies save (F(2x2) through HBM is no faster than the direct kernel there, and F(4x4), which is,
// carries 2.5x the rounding error that BatchNorm amplifies ~200x on the way to the head).  This kernel keeps both on the
// chip:
//
//   * a workgroup = 4 waves; a wave owns a PATCH of 4 x 8 tiles (8 x 16 output pixels) and 32 output channels, and holds
//     the accumulators of ALL 16 transform positions of that block: 16 x (32 x 32 MFMA tile) = 256 registers (the
//     accumulation half of the unified 512-register file; one wave per SIMD);
//   * the raw input patch (10 x 18 pixels x 16 channels per stage) comes global -> LDS by `buffer_load ... lds` DMA, the
//     zero padding is the buffer unit's out-of-range answer; each lane reads the 4 x 4 window of ITS tile (ds_read_b128:
//     4 channels) and transforms it in registers - the 16 results ARE the lane's A operands of the 16 planes' MFMAs
//     (lane = (tile, k half) is the A layout of v_mfma_f32_32x32x2_f32): V never exists in memory;
//   * the transformed filters U [16][Cout][Cin] (wino_filter_kernel<2>, rebuilt from the parameters every step) stream
//     through a second LDS ring, shared by the 4 waves (they work on the same 32 output channels);
//   * after the last channel chunk the 16 accumulator planes are folded to the 2 x 2 outputs in registers (A^T M A) and the
//     epilogue does what the igemm epilogue does: bias / eval-mode affine + leaky / accumulate / BatchNorm statistics
//     (counted format, one group per workgroup block) / fused BatchNorm-backward sums.  M never exists in memory.
//   * PERSISTENT: one workgroup per CU walks its (patch block, channel block) items as ONE software pipeline - the DMA of
//     an item's first chunk runs under the previous item's last chunk and epilogue - because with 512 registers per lane
//     there is no second workgroup on the CU to hide a prologue behind.
//
// LDS (all 160 KiB): raw ring 2 x 4 x 11520 B | U ring 2 x 32768 B | 6 KiB of epilogue scratch.  The raw patch of a wave is
// private to it (no barrier needed for it, only the wave's own vmcnt); U is shared: one s_barrier per 16-channel stage.
//
// Same arithmetic as the F(2x2) plans of conv_wino.hip (WinoMat<2>: constants 0, +-1, 1/2), fp32 throughout; the K loop is
// Cin long (not 9 Cin), so no chunked accumulation is needed.
#include <utility>

#include "conv_igemm_common.h"
#include "conv_wino.h"

#define WF_OOB 0x80000000u

namespace {
constexpr int WF_RAW_WAVE = 180 * 64;            // bytes of one wave's raw patch per stage: 10 x 18 pixels x 16 channels
constexpr int WF_RAW_STAGE = 4 * WF_RAW_WAVE;    // 46080
constexpr int WF_U_STAGE = 16 * 32 * 64;         // 32768: 16 planes x 32 output channels x 16 input channels
constexpr int WF_U_BASE = 2 * WF_RAW_STAGE;      // 92160
constexpr int WF_SCRATCH = WF_U_BASE + 2 * WF_U_STAGE;   // 157696
constexpr int WF_LDS_BYTES = WF_SCRATCH + 6144;  // 163840 = the CU's whole LDS
constexpr int WF_RAW_INSTR = 12;                 // 1-KiB DMA pieces per wave per stage for the raw patch (the last: 16 lanes)
constexpr int WF_U_INSTR = 8;                    // ... and for the wave's share of the U slab

struct WinoFusedArgs {
  const float* in;
  const float* U;       // [16][Cout][Cin]
  float* out;
  const float* bias;
  const float* escale;
  float act_slope;
  float* stats;         // [ntb][Cout][2] (mean, M2) | [ntb] pixel counts, or nullptr
  int B, H, W, Cin, Cout, ldin, ldout, accumulate;
  int npx, npy, npatch; // wave patches per image row / column, and in all
  int ntb, ncb, nitems; // patch blocks (4 patches), 32-channel blocks, items = ntb * ncb
  SspFastDiv div_npx, div_npy, div_ncb;
  const float* bn_raw;  // fused BatchNorm-backward sums (ConvArgs::bn_*)
  const float* bn_scale;
  const float* bn_shift;
  const float* bn_mean;
  const float* bn_invstd;
  float* bn_partial;
  int bn_nslot, bn_ld;
  float bn_slope;
  int probe;            // SSP_PROBES builds only (timing probes, WRONG results): 1 no stores, 2 no epilogue, 4 no input transform, 8 no DMA after the first stage, 16 no statistics
};
}  // namespace

// V = B^T d B of a 4 x 4 window, in place (4 channels per element).  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1].
__device__ __forceinline__ void wf_input_transform_(f32x4 (&d)[4][4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 a0 = d[0][j] - d[2][j], a1 = d[1][j] + d[2][j], a2 = d[2][j] - d[1][j], a3 = d[1][j] - d[3][j];
    d[0][j] = a0; d[1][j] = a1; d[2][j] = a2; d[3][j] = a3;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 a0 = d[i][0] - d[i][2], a1 = d[i][1] + d[i][2], a2 = d[i][2] - d[i][1], a3 = d[i][1] - d[i][3];
    d[i][0] = a0; d[i][1] = a1; d[i][2] = a2; d[i][3] = a3;
  }
}

// FLAGS (compile time, so that the epilogue is straight-line code): 1 = per-channel scale / bias / leaky slope (eval-mode block
// or a biased conv), 2 = accumulate into the output, 4 = fused BatchNorm-backward sums.
template <int FLAGS>
__global__ void __launch_bounds__(256, 1) wino2_fused_kernel(WinoFusedArgs p) {
#ifdef SSP_PROBES
  auto wf_input_transform = [&](f32x4 (&d)[4][4]) { if (!(p.probe & 4)) wf_input_transform_(d); };
#else
  auto wf_input_transform = [&](f32x4 (&d)[4][4]) { wf_input_transform_(d); };
#endif
  constexpr bool AFFINE = (FLAGS & 1) != 0, ACCUM = (FLAGS & 2) != 0, BNB = (FLAGS & 4) != 0;
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;

  // ---- item walk: iteration k of workgroup w takes item k * G + (w % 8) * (G / 8) + w / 8 - the G / 8 workgroups of an XCD
  // work on consecutive items, i.e. on the channel blocks of the same patch blocks: the raw patch is fetched into that
  // XCD's L2 once ----
  const int G = (int)gridDim.x;
  const int item_base = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
  int item = item_base;
  if (item >= p.nitems) return;

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)WF_OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void*)p.U, 0, (int)WF_OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)WF_OOB, 0x00020000);

  // ---- raw-patch DMA slots: piece j covers lane slots j * 64 + lane of the wave's 720 (pixel entry, 16-byte chunk) pairs.
  // Entry e = ry * 10 + parity * 5 + xh holds patch pixel (ry, rx = 2 xh + parity): the even and the odd columns of a row
  // are stored apart, so that the 4 tiles of a tile row read 4 CONSECUTIVE entries for any window column; the chunk index is
  // XOR-ed with (ry >> 1) & 3, which differs between the tile rows one ds_read_b128 lane group covers: conflict-free window
  // reads.  (Decomposed again for every item - constant divisions - rather than held in 12 registers.)
  // ---- U-slab DMA offsets (constant): piece gi = wid + 4 j covers rows (xi * 32 + n), 4 chunks each, swizzled like the
  // filter tiles of conv_igemm_dma.hip ----
  unsigned uvoff[WF_U_INSTR];
#pragma unroll
  for (int j = 0; j < WF_U_INSTR; ++j) {
    const int g = (wid + 4 * j) * 64 + lane;
    const int row = g >> 2, pch = g & 3;
    const int xi = row >> 5, n = row & 31;
    const int lch = pch ^ ((row >> 2) & 3);
    uvoff[j] = (unsigned)(((xi * p.Cout + n) * p.Cin + lch * 4) * 4);
  }
  // ---- fragment addresses ----
  // window of tile t = li (tx = t & 3, ty = t >> 2), row i, channel half q: entry (2 ty + i) * 10 + tx (+ column immediate)
  const int tx = li & 3, ty = li >> 2;
  unsigned awin[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      awin[i][q] = (unsigned)(wid * WF_RAW_WAVE + ((2 * ty + i) * 10 + tx) * 64 + (((2 * q + lh) ^ ((ty + (i >> 1)) & 3)) * 16));
  unsigned bfr[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) bfr[q] = (unsigned)(WF_U_BASE + li * 64 + (((2 * q + lh) ^ ((li >> 2) & 3)) * 16));

  // ---- per-item state (wave-uniform scalars) ----
  struct Item { int valid, b, y0, x0, n0, tb; };
  auto decode = [&](int it) {
    Item r;
    const unsigned tb = ssp_div((unsigned)it, p.div_ncb);
    const int cb = it - (int)tb * p.ncb;
    const unsigned P = tb * 4u + (unsigned)wid;
    const unsigned t2 = ssp_div(P, p.div_npx);
    const int pxi = (int)(P - t2 * (unsigned)p.npx);
    const unsigned b = ssp_div(t2, p.div_npy);
    const int pyi = (int)(t2 - b * (unsigned)p.npy);
    r.valid = ((int)P < p.npatch) ? 1 : 0;
    r.b = (int)b; r.y0 = pyi * 16; r.x0 = pxi * 8; r.n0 = cb * 32; r.tb = (int)tb;
    return r;
  };
  unsigned rvoff[WF_RAW_INSTR];
  auto raw_offsets = [&](const Item& it) {
#pragma unroll
    for (int j = 0; j < WF_RAW_INSTR; ++j) {
      const int g = j * 64 + lane;
      const int e = g >> 2, pch = g & 3;
      const int ry = e / 10, rem = e - ry * 10;
      const int par = rem >= 5 ? 1 : 0, xh = rem - 5 * par;
      const int rx = 2 * xh + par;
      const int lch = pch ^ ((ry >> 1) & 3);
      const int y = it.y0 - 1 + ry, x = it.x0 - 1 + rx;
      const bool ok = it.valid && ((unsigned)y < (unsigned)p.H) && ((unsigned)x < (unsigned)p.W);
      rvoff[j] = ok ? (unsigned)((((it.b * p.H + y) * p.W + x) * p.ldin + lch * 4) * 4) : WF_OOB;
    }
  };
  // DMA of one 16-channel stage into ring slot `buf`: the wave's raw patch first, its share of the U slab after it (so
  // that vmcnt(WF_U_INSTR) means "my raw patch has landed")
  bool first_issue = true;
  auto issue_stage = [&](int buf, int c0, int n0) {
#ifdef SSP_PROBES
    if ((p.probe & 8) && !first_issue) return;
    first_issue = false;
#endif
    char* rbase = lds + buf * WF_RAW_STAGE + wid * WF_RAW_WAVE;
#pragma unroll
    for (int j = 0; j < WF_RAW_INSTR - 1; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(rbase + j * 1024), 16, rvoff[j], c0 * 4, 0, 0);
    if (lane < 16)      // the 12th piece: entries 176 .. 179 only
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(rbase + (WF_RAW_INSTR - 1) * 1024), 16,
                                               rvoff[WF_RAW_INSTR - 1], c0 * 4, 0, 0);
    char* ubase = lds + WF_U_BASE + buf * WF_U_STAGE;
    const int so = (n0 * p.Cin + c0) * 4;
#pragma unroll
    for (int j = 0; j < WF_U_INSTR; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (__attribute__((address_space(3))) void*)(ubase + (wid + 4 * j) * 1024), 16, uvoff[j], so, 0, 0);
  };

  f32x16 acc[16];
  f32x4 va[4][4], vb[4][4];      // transformed windows: the set being multiplied and the set being prepared

  auto load_windows = [&](f32x4 (&d)[4][4], auto buf_tag, auto q_tag) {
    constexpr int S = decltype(buf_tag)::value, Q = decltype(q_tag)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        d[i][j] = *reinterpret_cast<const f32x4*>(lds + awin[i][Q] + (S * WF_RAW_STAGE + ((j & 1) * 5 + (j >> 1)) * 64));
  };
  // the 64 MFMAs of one (stage, channel half): planes in pairs so that consecutive MFMAs never share an accumulator; the
  // filter fragments of pair k + 1 are fetched in front of pair k's MFMAs and the scheduling barrier keeps the compiler from
  // pulling more of them forward (it fetched 18 fragments ahead and spilled the DMA offsets)
  auto mma = [&](const f32x4 (&d)[4][4], auto buf_tag, auto q_tag, auto fresh_tag) {
    constexpr int S = decltype(buf_tag)::value, Q = decltype(q_tag)::value;
    constexpr bool FRESH = decltype(fresh_tag)::value;
    constexpr f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x4 b0 = *reinterpret_cast<const f32x4*>(lds + bfr[Q] + (S * WF_U_STAGE));
    f32x4 b1 = *reinterpret_cast<const f32x4*>(lds + bfr[Q] + (S * WF_U_STAGE + 2048));
#pragma unroll
    for (int x2 = 0; x2 < 16; x2 += 2) {
      f32x4 n0 = b0, n1 = b1;
      if (x2 + 2 < 16) {
        n0 = *reinterpret_cast<const f32x4*>(lds + bfr[Q] + (S * WF_U_STAGE + (x2 + 2) * 2048));
        n1 = *reinterpret_cast<const f32x4*>(lds + bfr[Q] + (S * WF_U_STAGE + (x2 + 3) * 2048));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (FRESH && e == 0) {
          acc[x2] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[x2 >> 2][x2 & 3][e], b0[e], zero16, 0, 0, 0);
          acc[x2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[(x2 + 1) >> 2][(x2 + 1) & 3][e], b1[e], zero16, 0, 0, 0);
        } else {
          acc[x2] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[x2 >> 2][x2 & 3][e], b0[e], acc[x2], 0, 0, 0);
          acc[x2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[(x2 + 1) >> 2][(x2 + 1) & 3][e], b1[e], acc[x2 + 1], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      b0 = n0; b1 = n1;
    }
  };
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;

  // ---- epilogue of one item ----
  // Register r of a 32 x 32 accumulator tile: tile row t = (r & 3) + 8 (r >> 2) + 4 lh, i.e. tx = r & 3, ty = 2 (r >> 2) + lh;
  // column = li = output channel n0 + li.  Output pixel of Y[pp][qq] of register r: (y0 + 2 ty + pp, x0 + 2 tx + qq).
  // The 16 planes are folded register by register (Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]) so that only 16 + 4 values are
  // live at a time: the accumulators fill the accumulation half of the register file, everything else shares the other half
  // with the next item's transformed windows.
  float* const scratch = reinterpret_cast<float*>(lds + WF_SCRATCH);
  auto epilogue = [&](const Item& it) {
    const int n = it.n0 + li;
    const int ld4 = p.ldout * 4;
    const bool interior = it.valid && (it.y0 + 16 <= p.H) && (it.x0 + 8 <= p.W);
    const unsigned vbase = (unsigned)((((it.b * p.H + it.y0 + 2 * lh) * p.W + it.x0) * p.ldout + n) * 4);
    float bias = 0.f, esc = 1.f;
    if constexpr (AFFINE) {
      bias = p.bias != nullptr ? p.bias[n] : 0.f;
      esc = p.escale != nullptr ? p.escale[n] : 1.f;
    }
    bool want_stats = p.stats != nullptr;
#ifdef SSP_PROBES
    if (p.probe & 16) want_stats = false;
#endif
    __amdgpu_buffer_rsrc_t rs_x = rs_out;
    float b_sc = 0.f, b_sh = 0.f, b_mu = 0.f, b_is = 0.f, s1 = 0.f, s2 = 0.f;
    unsigned vbase_x = 0;
    int ldx4 = 0;
    if constexpr (BNB) {
      rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.bn_raw, 0, (int)WF_OOB, 0x00020000);
      b_sc = p.bn_scale[n]; b_sh = p.bn_shift[n]; b_mu = p.bn_mean[n]; b_is = p.bn_invstd[n];
      vbase_x = (unsigned)((((it.b * p.H + it.y0 + 2 * lh) * p.W + it.x0) * p.bn_ld + n) * 4);
      ldx4 = p.bn_ld * 4;
    }
    // statistics of the raw values in ONE pass, shifted by the lane's first value (a sample of the channel: the sums of
    // (x - K) and (x - K)^2 over 64 values do not cancel): mean = K + s / n, M2 = ss - s^2 / n
    float cnt = 0.f, sh_s = 0.f, sh_ss = 0.f, shift_k = 0.f;
    // Y = A^T M A, one OUTPUT row pp at a time (2 + 2 vectors live, not 4 + 2): c0 / c1 = a row of the transform domain
    // folded over its columns, added into the output row with A^T's signs
    auto half = [&](auto pp_tag) {
      constexpr int pp = decltype(pp_tag)::value;
      f32x16 Ya, Yb;      // Y[pp][0], Y[pp][1]
      // (a row's four planes are pinned in the accumulation registers by an empty volatile asm right before they are read,
      // and the asm takes the running sums as operands so that it cannot be scheduled ahead of the previous row's adds:
      // without it the compiler reads all 256 accumulators at the top of the epilogue and spills half the register file)
      auto row = [&](auto i_tag, f32x16& c0, f32x16& c1) {
        constexpr int i = decltype(i_tag)::value;
        asm volatile("" : "+a"(acc[4 * i]), "+a"(acc[4 * i + 1]), "+a"(acc[4 * i + 2]), "+a"(acc[4 * i + 3]), "+v"(Ya), "+v"(Yb));
        c0 = acc[4 * i] + acc[4 * i + 1] + acc[4 * i + 2];
        c1 = acc[4 * i + 1] - acc[4 * i + 2] - acc[4 * i + 3];
      };
      f32x16 c0, c1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { Ya[r] = 0.f; Yb[r] = 0.f; }
      if constexpr (pp == 0) {
        row(std::integral_constant<int, 0>{}, c0, c1);
        Ya = c0; Yb = c1;
        row(std::integral_constant<int, 1>{}, c0, c1);
        Ya += c0; Yb += c1;
        row(std::integral_constant<int, 2>{}, c0, c1);
        Ya += c0; Yb += c1;
      } else {
        row(std::integral_constant<int, 1>{}, c0, c1);
        Ya = c0; Yb = c1;
        row(std::integral_constant<int, 2>{}, c0, c1);
        Ya -= c0; Yb -= c1;
        row(std::integral_constant<int, 3>{}, c0, c1);
        Ya -= c0; Yb -= c1;
      }
      asm volatile("" : "+v"(Ya), "+v"(Yb));
      if constexpr (pp == 0) shift_k = Ya[0];
      int row_o = pp * p.W * ld4, row_x = pp * p.W * ldx4;      // scalar byte offsets of pixel row 4 rq + pp
      auto group = [&](auto rq_tag) {
        constexpr int rq = decltype(rq_tag)::value;
        asm volatile("" : "+s"(row_o), "+s"(row_x));     // opaque: one group's offsets live at a time (no hoisting of all 64)
        // validity of this lane's pixels of the group: row y0 + 4 rq + 2 lh + pp, columns x0 + 2 rr + qq = x0 + o
        unsigned okmask = 0xffu;                       // bit o = 2 rr + qq
        if (!interior) {
          okmask = 0u;
          const int y = it.y0 + 4 * rq + 2 * lh + pp;
#pragma unroll
          for (int o = 0; o < 8; ++o)
            if (it.valid && y < p.H && it.x0 + o < p.W) okmask |= 1u << o;
        }
        float xr[8], prev[8];
        if constexpr (BNB) {
#pragma unroll
          for (int o = 0; o < 8; ++o)
            xr[o] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, ((okmask >> o) & 1u) ? vbase_x : WF_OOB,
                                                                                   row_x + o * ldx4, 0));
        }
        if constexpr (ACCUM) {
#pragma unroll
          for (int o = 0; o < 8; ++o)
            prev[o] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_out, ((okmask >> o) & 1u) ? vbase : WF_OOB,
                                                                                     row_o + o * ld4, 0));
        }
#pragma unroll
        for (int o = 0; o < 8; ++o) {                  // o = 2 rr + qq: eight consecutive pixels of the row
          const float raw = (o & 1) ? Yb[rq * 4 + (o >> 1)] : Ya[rq * 4 + (o >> 1)];
          float v = raw;
          if constexpr (AFFINE) {
            v = v * esc + bias;
            v = v > 0.f ? v : v * p.act_slope;
          }
          if constexpr (ACCUM) v += prev[o];
          const bool ok = (okmask >> o) & 1u;
#ifdef SSP_PROBES
          if (!(p.probe & 1) || v == 12345.678f)
#endif
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, ok ? vbase : WF_OOB, row_o + o * ld4, 0);
          const float okf = ok ? 1.f : 0.f;            // branch-free: masked pixels add zeros
          if constexpr (!BNB && !ACCUM) {              // (accumulating / data-gradient launches produce no statistics)
            const float dd = (raw - shift_k) * okf;
            cnt += okf;
            sh_s += dd;
            sh_ss += dd * dd;
          }
          if constexpr (BNB) {
            const float yb = xr[o] * b_sc + b_sh;
            const float dyv = (yb > 0.f ? v : v * p.bn_slope) * okf;
            s1 += dyv;
            s2 += dyv * ((xr[o] - b_mu) * b_is);
          }
        }
        row_o += 4 * p.W * ld4;
        row_x += 4 * p.W * ldx4;
      };
      group(std::integral_constant<int, 0>{});
      group(std::integral_constant<int, 1>{});
      group(std::integral_constant<int, 2>{});
      group(std::integral_constant<int, 3>{});
    };
    half(std::integral_constant<int, 0>{});
    half(std::integral_constant<int, 1>{});
    if constexpr (BNB) {
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (lh == 0) { scratch[(wid * 32 + li) * 2 + 0] = s1; scratch[(wid * 32 + li) * 2 + 1] = s2; }
      __syncthreads();
      if (tid < 32) {
        float a = 0.f, bsum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += scratch[(w * 32 + tid) * 2 + 0]; bsum += scratch[(w * 32 + tid) * 2 + 1]; }
        const int ch = it.n0 + tid;
        if (p.ntb > p.bn_nslot) {
          float* dst = p.bn_partial + ((int64_t)(it.tb % p.bn_nslot) * p.Cout + ch) * 2;
          atomicAdd(dst, a);
          atomicAdd(dst + 1, bsum);
        } else {
          float* dst = p.bn_partial + ((int64_t)it.tb * p.Cout + ch) * 2;
          dst[0] = a;
          dst[1] = bsum;
        }
      }
    }
    if constexpr (!BNB && !ACCUM) {
      if (want_stats) {
        // per-lane (count, mean, M2), Chan-combined: lane halves -> waves (through LDS) -> one triple per channel and block
        float mean = cnt > 0.f ? shift_k + sh_s / cnt : 0.f;
        float m2 = cnt > 0.f ? fmaxf(sh_ss - sh_s * sh_s / cnt, 0.f) : 0.f;
        const float ocnt = __shfl_xor(cnt, 32), omean = __shfl_xor(mean, 32), om2 = __shfl_xor(m2, 32);
        chan_combine(cnt, mean, m2, ocnt, omean, om2);
        if (lh == 0) {
          scratch[512 + (wid * 32 + li) * 3 + 0] = cnt;
          scratch[512 + (wid * 32 + li) * 3 + 1] = mean;
          scratch[512 + (wid * 32 + li) * 3 + 2] = m2;
        }
        __syncthreads();
        if (tid < 32) {
          float c_ = scratch[512 + tid * 3 + 0], mu = scratch[512 + tid * 3 + 1], ss = scratch[512 + tid * 3 + 2];
#pragma unroll
          for (int w = 1; w < 4; ++w)
            chan_combine(c_, mu, ss, scratch[512 + (w * 32 + tid) * 3 + 0], scratch[512 + (w * 32 + tid) * 3 + 1],
                         scratch[512 + (w * 32 + tid) * 3 + 2]);
          float* st = p.stats + ((int64_t)it.tb * p.Cout + it.n0 + tid) * 2;
          st[0] = mu;
          st[1] = ss;
          if (tid == 0 && it.n0 == 0) p.stats[(int64_t)p.ntb * p.Cout * 2 + it.tb] = c_;      // the block's pixel count
        }
      }
    }
  };

  // ---- the pipeline ----
  const int nsp = p.Cin >> 5;        // stage PAIRS per item (Cin % 32 == 0)
  Item cur = decode(item);
  raw_offsets(cur);
  issue_stage(0, 0, cur.n0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  load_windows(va, T0{}, T0{});
  wf_input_transform(va);
  bool stage1_issued = false;        // the DMA of this item's second stage was queued before the previous item's epilogue
  for (;;) {
    Item nxt = cur;
    bool has_next = false;
    for (int sp = 0; sp < nsp; ++sp) {
      // ---- stage 2 sp (ring slot 0); the DMA of stage 2 sp + 1 goes to slot 1 ----
      if (!(sp == 0 && stage1_issued)) issue_stage(1, (2 * sp + 1) * 16, cur.n0);
      load_windows(vb, T0{}, T1{});
      if (sp == 0) mma(va, T0{}, T0{}, std::true_type{});
      else mma(va, T0{}, T0{}, std::false_type{});
      wf_input_transform(vb);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // my raw patch of slot 1 has landed (WF_U_INSTR newer pieces may fly)
      load_windows(va, T1{}, T0{});
      mma(vb, T0{}, T1{}, std::false_type{});
      wf_input_transform(va);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // ---- stage 2 sp + 1 (slot 1); next: this item's stage 2 sp + 2, or the next item's first stage, into slot 0 ----
      if (sp + 1 < nsp) {
        issue_stage(0, (2 * sp + 2) * 16, cur.n0);
      } else {
        const int nitem = item + G;
        has_next = nitem < p.nitems;
        if (has_next) {
          nxt = decode(nitem);
          raw_offsets(nxt);
          issue_stage(0, 0, nxt.n0);
        }
      }
      load_windows(vb, T1{}, T1{});
      mma(va, T1{}, T0{}, std::false_type{});
      wf_input_transform(vb);
      // (the first windows of the NEXT ITEM are fetched after the epilogue instead: the epilogue then has the whole register
      // file to itself - with them live it spilled)
      const bool same_item = sp + 1 < nsp;
      if (same_item) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        load_windows(va, T0{}, T0{});
      }
      mma(vb, T1{}, T1{}, std::false_type{});
      if (same_item) wf_input_transform(va);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    // the next item's second stage goes out before this item's epilogue (slot 1 is free since the barrier above; rvoff
    // already holds the next item's offsets): the epilogue's VALU / store work runs under it
    stage1_issued = has_next;
    if (has_next) issue_stage(1, 16, nxt.n0);
#ifdef SSP_PROBES
    if (!(p.probe & 2) || acc[0][0] == 12345.678f)
#endif
    epilogue(cur);
    if (!has_next) break;
    item += G;
    cur = nxt;
    load_windows(va, T0{}, T0{});      // slot 0 holds the new item's first stage (landed before the barrier above)
    wf_input_transform(va);
  }
#endif
}

// ---- host side ----
static void wf_geometry(int B, int H, int W, int Cout, int& npx, int& npy, int& npatch, int& ntb, int& ncb) {
  const int th = (H + 1) / 2, tw = (W + 1) / 2;
  npx = (tw + 3) / 4;
  npy = (th + 7) / 8;
  npatch = B * npy * npx;
  ntb = (npatch + 3) / 4;
  ncb = Cout / 32;
}
int ssp_wino_fused_stat_groups(int B, int H, int W, int Cout) {
  int npx, npy, npatch, ntb, ncb;
  wf_geometry(B, H, W, Cout, npx, npy, npatch, ntb, ncb);
  return ntb;
}
bool ssp_wino_fused_fits(int B, int H, int W, int Cin, int Cout, int R) {
  return R == 3 && Cin % 32 == 0 && Cout % 32 == 0 && Cin >= 32 && (int64_t)B * H * W < (1ll << 28);
}

template <int FLAGS>
static int wf_launch(const WinoFusedArgs& p, hipStream_t stream) {
  static SspKernelCache cache;      // per instantiation, per device
  int slots = 0;
  auto kern = wino2_fused_kernel<FLAGS>;
  if (int rc = ssp_kernel_prepare((const void*)kern, WF_LDS_BYTES, 256, &cache, &slots, "wino2_fused")) return rc;
  int grid = slots < 8 ? 8 : (slots / 8) * 8;
  const int want = ((p.nitems + 7) / 8) * 8;
  if (grid > want) grid = want;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), WF_LDS_BYTES, stream, p);
  SSP_CHECK_LAUNCH("wino2_fused");
  return SSP_OK;
}

int ssp_wino_fused_launch(const ConvArgs& a, int B, int H, int W, int prof_kind, hipStream_t stream) {
  SSP_CHECK_ARG(ssp_wino_fused_fits(B, H, W, a.Cin, a.Cout, a.R),
                "conv (on-chip Winograd plan): needs a 3x3 filter, Cin %% 32 == 0 and Cout %% 32 == 0");
  SSP_CHECK_ARG(a.ldin % 4 == 0 && (((uintptr_t)a.in) & 15) == 0 && (((uintptr_t)a.wt) & 15) == 0,
                "conv (on-chip Winograd plan): 16-byte aligned operands, ldin %% 4 == 0");
  SSP_CHECK_ARG((int64_t)B * H * W * a.ldin * 4 < (1ll << 31) && (int64_t)B * H * W * a.ldout * 4 < (1ll << 31) &&
                    (int64_t)16 * a.Cout * a.Cin * 4 < (1ll << 31) &&
                    (a.bn_partial == nullptr || (int64_t)B * H * W * a.bn_ld * 4 < (1ll << 31)),
                "conv (on-chip Winograd plan): operands beyond the 2 GiB buffer range");
  WinoFusedArgs p;
  p.in = a.in; p.U = a.wt; p.out = a.out; p.bias = a.bias; p.escale = a.escale; p.act_slope = a.act_slope; p.stats = a.stats;
  p.B = B; p.H = H; p.W = W; p.Cin = a.Cin; p.Cout = a.Cout; p.ldin = a.ldin; p.ldout = a.ldout; p.accumulate = a.accumulate;
  wf_geometry(B, H, W, a.Cout, p.npx, p.npy, p.npatch, p.ntb, p.ncb);
  p.nitems = p.ntb * p.ncb;
  p.div_npx = ssp_fastdiv((unsigned)p.npx); p.div_npy = ssp_fastdiv((unsigned)p.npy); p.div_ncb = ssp_fastdiv((unsigned)p.ncb);
  p.bn_raw = a.bn_raw; p.bn_scale = a.bn_scale; p.bn_shift = a.bn_shift; p.bn_mean = a.bn_mean; p.bn_invstd = a.bn_invstd;
  p.bn_partial = a.bn_partial; p.bn_nslot = a.bn_nslot; p.bn_ld = a.bn_ld; p.bn_slope = a.bn_slope;
  p.probe = 0;
#ifdef SSP_PROBES
  p.probe = ssp_option(SSP_OPT_WINO_VARIANT) >> 8;
#endif
  const int flags = ((a.bias != nullptr || a.escale != nullptr || a.act_slope != 1.f) ? 1 : 0) | (a.accumulate ? 2 : 0) |
                    (a.bn_partial != nullptr ? 4 : 0);
  SSP_CHECK_ARG(flags <= 4, "conv (on-chip Winograd plan): the fused BatchNorm-backward sums need a plain data-gradient launch");
  SSP_CHECK_ARG(a.stats == nullptr || (flags & 6) == 0, "conv (on-chip Winograd plan): statistics only from a non-accumulating launch");
  SspProfScope prof(prof_kind, stream, 2.0 * (double)a.M * a.Cout * 9.0 * a.Cin);      // algorithmic (direct) FLOPs
  switch (flags) {
    case 1: return wf_launch<1>(p, stream);
    case 2: return wf_launch<2>(p, stream);
    case 3: return wf_launch<3>(p, stream);
    case 4: return wf_launch<4>(p, stream);
    default: return wf_launch<0>(p, stream);
  }
}
