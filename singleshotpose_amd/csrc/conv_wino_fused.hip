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
//     4 channels) and transforms it in registers - the 16 results ARE the lane's B operands of the 16 planes' MFMAs
//     (lane = (tile, k half) is the operand layout of v_mfma_f32_32x32x2_f32): V never exists in memory;
//   * the transformed filters U [16][Cout][Cin] (wino_filter_kernel<2>, rebuilt from the parameters every step) stream
//     through a second LDS ring, shared by the 4 waves (they work on the same 32 output channels); they are the A operand,
//     so an accumulator tile is [channel][tile]: a lane holds 4 CONSECUTIVE channels of one tile in 4 consecutive registers;
//   * after the last channel chunk the 16 accumulator planes are folded to the 2 x 2 outputs in registers (A^T M A) and the
//     epilogue does what the igemm epilogue does: bias / eval-mode affine + leaky / accumulate / BatchNorm statistics
//     (counted format, one group per workgroup block) / fused BatchNorm-backward sums.  M never exists in memory.  Outputs
//     leave as 16 `buffer_store_dwordx4` per lane;
//   * PERSISTENT: one workgroup per CU walks its (patch block, channel block) items as ONE software pipeline: with 512
//     registers per lane there is no second workgroup on the CU to hide a prologue or a store burst behind.  The stores of
//     an item drain under the next item's MFMAs: the vmcnt waits of the next item's first stage leave them outstanding
//     (vmcnt counts loads and stores in issue order on gfx9: "all but the N youngest");
//   * inside a stage the memory instructions (LDS window / fragment reads, the next stage's DMA pieces) are dealt one or
//     two at a time between PAIRS of MFMAs (a 32x32x2 fp32 MFMA occupies the pipe 64 cycles: one memory instruction issues
//     in its shadow) - as a block of 20 DMA pieces in front of the MFMAs they cost 15 % of the launch.
//
// LDS (all 160 KiB): raw ring 2 x 4 x 12288 B | U ring 2 x 32768 B.  A wave's raw patch is 180 entries of 64 B = 11520 B,
// fetched by 12 DMA pieces of 1 KiB: the idle lanes of the 12th write zeros into the 768 B behind it, and those 768 B of
// ring slot 0 - quiet while an epilogue runs - are the epilogue's cross-wave scratch.  The raw patch of a wave is private
// to it (no barrier needed for it, only the wave's own vmcnt); U is shared: one s_barrier per 16-channel stage.
//
// Same arithmetic as the F(2x2) plans of conv_wino.hip (WinoMat<2>: constants 0, +-1, 1/2), fp32 throughout; the K loop is
// Cin long (not 9 Cin), so no chunked accumulation is needed.
#include <utility>

#include "conv_igemm_common.h"
#include "conv_wino.h"

#define WF_OOB 0x80000000u
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

namespace {
constexpr int WF_RAW_USED = 180 * 64;            // bytes of one wave's raw patch per stage: 10 x 18 pixels x 16 channels
constexpr int WF_RAW_WAVE = 12 * 1024;           // ... and what its 12 DMA pieces cover (the last 768 B: zeros / scratch)
constexpr int WF_RAW_STAGE = 4 * WF_RAW_WAVE;    // 49152
constexpr int WF_U_STAGE = 16 * 32 * 64;         // 32768: 16 planes x 32 output channels x 16 input channels
constexpr int WF_U_BASE = 2 * WF_RAW_STAGE;      // 98304
constexpr int WF_LDS_BYTES = WF_U_BASE + 2 * WF_U_STAGE;   // 163840 = the CU's whole LDS
constexpr int WF_RAW_INSTR = 12;                 // 1-KiB DMA pieces per wave per stage for the raw patch
constexpr int WF_U_INSTR = 8;                    // ... and for the wave's share of the U slab
constexpr int WF_NDMA = WF_RAW_INSTR + WF_U_INSTR;

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
  int probe;            // SSP_PROBES builds only (timing probes, WRONG results): 1 no stores, 2 no epilogue, 4 no input
                        // transform, 8 no DMA after the first stage, 16 no statistics
};

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <typename F, int... Is>
__device__ __forceinline__ void wf_sfor_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void wf_sfor(F&& f) {
  wf_sfor_impl(f, std::make_integer_sequence<int, N>{});
}
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

// Sum over the 32 lanes of a wave half of 16 per-lane values, as a reduce-scatter: after the four halving steps a lane holds
// the sum over 16 lanes of value r = 8 b4 + 4 b3 + 2 b2 + b1 (b_k = bit k of li), the last step adds its neighbour's half:
// 16 shuffles instead of 80.
__device__ __forceinline__ float wf_reduce16(const float (&v)[16], int li) {
  float w8[8], w4[4], w2[2];
  const bool b4 = (li & 16) != 0, b3 = (li & 8) != 0, b2 = (li & 4) != 0, b1 = (li & 2) != 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float keep = b4 ? v[j + 8] : v[j], send = b4 ? v[j] : v[j + 8];
    w8[j] = keep + __shfl_xor(send, 16);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float keep = b3 ? w8[j + 4] : w8[j], send = b3 ? w8[j] : w8[j + 4];
    w4[j] = keep + __shfl_xor(send, 8);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float keep = b2 ? w4[j + 2] : w4[j], send = b2 ? w4[j] : w4[j + 2];
    w2[j] = keep + __shfl_xor(send, 4);
  }
  const float keep = b1 ? w2[1] : w2[0], send = b1 ? w2[0] : w2[1];
  const float w1 = keep + __shfl_xor(send, 2);
  return w1 + __shfl_xor(w1, 1);
}
// the element r = 8 b4 + 4 b3 + 2 b2 + b1 of a per-lane array (the one wf_reduce16 leaves in the lane)
__device__ __forceinline__ float wf_select16(const float (&v)[16], int li) {
  float w8[8], w4[4], w2[2];
#pragma unroll
  for (int j = 0; j < 8; ++j) w8[j] = (li & 16) ? v[j + 8] : v[j];
#pragma unroll
  for (int j = 0; j < 4; ++j) w4[j] = (li & 8) ? w8[j + 4] : w8[j];
#pragma unroll
  for (int j = 0; j < 2; ++j) w2[j] = (li & 4) ? w4[j + 2] : w4[j];
  return (li & 2) ? w2[1] : w2[0];
}

// FLAGS (compile time, so that the epilogue is straight-line code): 1 = per-channel scale / bias / leaky slope (eval-mode block
// or a biased conv), 2 = accumulate into the output, 4 = fused BatchNorm-backward sums.
template <int FLAGS>
__global__ void __launch_bounds__(256, 1) wino2_fused_kernel(WinoFusedArgs p) {
  constexpr bool AFFINE = (FLAGS & 1) != 0, ACCUM = (FLAGS & 2) != 0, BNB = (FLAGS & 4) != 0;
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef SSP_PROBES
  auto wf_input_transform = [&](f32x4 (&d)[4][4]) { if (!(p.probe & 4)) wf_input_transform_(d); };
#else
  auto wf_input_transform = [&](f32x4 (&d)[4][4]) { wf_input_transform_(d); };
#endif
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;

  // ---- item walk: iteration k of workgroup w takes item k * G + (w % 8) * (G / 8) + w / 8 - the G / 8 workgroups of an XCD
  // work on consecutive items, i.e. on the channel blocks of the same patch blocks: the raw patch is fetched into that
  // XCD's L2 once ----
  const int G = (int)gridDim.x;
  int item = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
  if (item >= p.nitems) return;
#ifdef SSP_PROBES
  if (p.probe & 32) {      // stagger the workgroups of an XCD over ~one item time (store-burst experiment)
    const int k = (((int)blockIdx.x >> 3) & 7) * ((p.probe >> 8) & 15);
    for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)WF_OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void*)p.U, 0, (int)WF_OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)WF_OOB, 0x00020000);

  // ---- U-slab DMA offsets (constant): piece gi = wid + 4 j covers rows (xi * 32 + n), 4 chunks each, swizzled like the
  // filter tiles of conv_igemm_dma.hip ----
  unsigned uvoff0;      // piece gi covers plane gi >> 1, channels (gi & 1) * 16 + (lane >> 2): piece j is piece 0 + j * 2 planes
  {
    const int g = wid * 64 + lane;
    const int row = g >> 2, pch = g & 3;
    const int xi = row >> 5, n = row & 31;
    const int lch = pch ^ ((row >> 2) & 3);
    uvoff0 = (unsigned)(((xi * p.Cout + n) * p.Cin + lch * 4) * 4);
  }
  const int ustep = 2 * p.Cout * p.Cin * 4;      // bytes between the rows of pieces j and j + 1 (two planes)
  // ---- fragment addresses ----
  // Raw patch of a wave: entry e = ry * 10 + parity * 5 + xh holds patch pixel (ry, rx = 2 xh + parity) - the even and the
  // odd columns of a row are stored apart, so that the 4 tiles of a tile row read 4 CONSECUTIVE entries for any window column;
  // the 16-byte chunk index is XOR-ed with (ry >> 1) & 3, which differs between the tile rows one ds_read_b128 lane group
  // covers: conflict-free window reads.  Window of tile t = li (tx = t & 3, ty = t >> 2), row i, channel half q: entry
  // (2 ty + i) * 10 + tx, + ((j & 1) * 5 + (j >> 1)) entries for column j (an immediate).
  const int tx = li & 3, ty = li >> 2;
  unsigned awin[2][2];      // window rows 2 ih and 2 ih + 1 share their swizzle term: the second is the first + 640 bytes
#pragma unroll
  for (int ih = 0; ih < 2; ++ih)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      awin[ih][q] = (unsigned)(wid * WF_RAW_WAVE + ((2 * ty + 2 * ih) * 10 + tx) * 64 + (((2 * q + lh) ^ ((ty + ih) & 3)) * 16));
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
  // raw-patch DMA offsets of an item: piece j covers lane slots j * 64 + lane of the wave's (entry, chunk) pairs; slots past
  // entry 179 (lanes 16 .. 63 of the 12th piece) and pixels outside the image read as zeros (out-of-range offset)
  unsigned rvoff[WF_RAW_INSTR];
  auto raw_offsets = [&](const Item& it, bool live) {
    // (the slot decomposition below is item-invariant; the empty asm keeps the compiler from hoisting its 36 values out of the
    // item loop, where they only get spilled and reloaded one by one - each reload behind a full vmcnt wait)
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));
#pragma unroll
    for (int j = 0; j < WF_RAW_INSTR; ++j) {
      const int g = j * 64 + lane_;
      const int e = g >> 2, pch = g & 3;
      const int ry = e / 10, rem = e - ry * 10;
      const int par = rem >= 5 ? 1 : 0, xh = rem - 5 * par;
      const int rx = 2 * xh + par;
      const int lch = pch ^ ((ry >> 1) & 3);
      const int y = it.y0 - 1 + ry, x = it.x0 - 1 + rx;
      const bool ok = live && it.valid && e < 180 && ((unsigned)y < (unsigned)p.H) && ((unsigned)x < (unsigned)p.W);
      rvoff[j] = ok ? (unsigned)((((it.b * p.H + y) * p.W + x) * p.ldin + lch * 4) * 4) : WF_OOB;
    }
  };
  // one DMA piece of a 16-channel stage into ring slot S: pieces 0 .. 11 = the wave's raw patch, 12 .. 19 = its share of the
  // U slab (in that order, so that vmcnt(WF_U_INSTR) means "my raw patch has landed")
  bool first_issue = true;
  int dma_c0 = 0, dma_uso = 0;       // channel offset of the stage being fetched (bytes), scalar offset into U
  auto dma_piece = [&](auto s_tag, auto j_tag) {
    constexpr int S = decltype(s_tag)::value, J = decltype(j_tag)::value;
#ifdef SSP_PROBES
    if ((p.probe & 8) && !first_issue) return;
#endif
    if constexpr (J < WF_RAW_INSTR)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(lds + S * WF_RAW_STAGE + wid * WF_RAW_WAVE + J * 1024),
                                               16, rvoff[J], dma_c0, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (__attribute__((address_space(3))) void*)(lds + WF_U_BASE + S * WF_U_STAGE + (wid + 4 * (J - WF_RAW_INSTR)) * 1024),
                                               16, uvoff0, dma_uso + (J - WF_RAW_INSTR) * ustep, 0, 0);
  };
  auto set_stage = [&](int c0, int n0) { dma_c0 = c0 * 4; dma_uso = (n0 * p.Cin + c0) * 4; };
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using TT = std::true_type;
  using TF = std::false_type;
  auto issue_block = [&](auto s_tag) {      // all 20 pieces back to back (prologue, and in front of an epilogue)
    wf_sfor<WF_NDMA>([&](auto J) { dma_piece(s_tag, J); });
    first_issue = false;
  };

  f32x16 acc[16];
  f32x4 va[4][4], vb[4][4];      // transformed windows: the set being multiplied and the set being prepared

  // ---- one (stage S, channel half Q): 64 MFMAs in 32 groups of two (the planes in pairs, so that consecutive MFMAs never
  // share an accumulator), with this half's memory instructions dealt between the groups:
  //   * the filter fragments of plane pair k + 1, in front of pair k's first group;
  //   * WIN: the 16 window reads of the NEXT half (slot WS, half WQ) into `nxt`, one per group;
  //   * DMA: the 20 pieces of the stage being fetched into slot S ^ 1, in front of groups 1, 2, 4, 5, 7, 8 ... 29.
  // A scheduling barrier behind every group keeps the compiler from regrouping them (pulled forward, the fragment reads alone
  // spilled the register file).
  auto half_stage = [&](const f32x4 (&cur)[4][4], f32x4 (&nxt)[4][4], auto s_tag, auto q_tag, auto fresh_tag, auto dma_tag,
                        auto win_tag, auto ws_tag, auto wq_tag, bool vm24 = false) {
    constexpr int S = decltype(s_tag)::value, Q = decltype(q_tag)::value;
    constexpr bool FRESH = decltype(fresh_tag)::value, DMA = decltype(dma_tag)::value, WIN = decltype(win_tag)::value;
    constexpr int WS = decltype(ws_tag)::value, WQ = decltype(wq_tag)::value;
    constexpr f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x4 bb[2][2];
    bb[0][0] = *reinterpret_cast<const f32x4*>(lds + bfr[Q] + (S * WF_U_STAGE));
    bb[0][1] = *reinterpret_cast<const f32x4*>(lds + bfr[Q] + (S * WF_U_STAGE + 2048));
    wf_sfor<32>([&](auto G_) {
      constexpr int g = decltype(G_)::value;
      constexpr int pr = g >> 2, e = g & 3, x2 = 2 * pr;
      if constexpr (e == 0 && pr < 7) {
        bb[(pr + 1) & 1][0] = *reinterpret_cast<const f32x4*>(lds + bfr[Q] + (S * WF_U_STAGE + (x2 + 2) * 2048));
        bb[(pr + 1) & 1][1] = *reinterpret_cast<const f32x4*>(lds + bfr[Q] + (S * WF_U_STAGE + (x2 + 3) * 2048));
      }
      // windows of the SAME slot (first half) are read in groups 0 .. 15; windows of the OTHER slot (second half) in groups
      // 16 .. 31, behind a vmcnt wait in front of group 16: the wave's raw patch of that slot was fetched during the first
      // half - the later it is needed, the less of its HBM latency (~2 us under load) shows
      constexpr int WG0 = (WS == S) ? 0 : 16;
      if constexpr (WIN && WS != S && g == 16) {
        if (vm24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
      if constexpr (WIN && g >= WG0 && g < WG0 + 16) {
        constexpr int i = (g - WG0) >> 2, j = (g - WG0) & 3;
        nxt[i][j] = *reinterpret_cast<const f32x4*>(lds + awin[i >> 1][WQ] + (WS * WF_RAW_STAGE + (i & 1) * 640 + ((j & 1) * 5 + (j >> 1)) * 64));
      }
      // DMA pieces: the raw patch first, one piece in front of each of groups 1 .. 12, the U slab in front of groups 13, 15 .. 27
      if constexpr (DMA && g >= 1 && g <= WF_RAW_INSTR) dma_piece(std::integral_constant<int, S ^ 1>{}, std::integral_constant<int, g - 1>{});
      if constexpr (DMA && g > WF_RAW_INSTR && g <= WF_RAW_INSTR + 2 * WF_U_INSTR - 1 && ((g - WF_RAW_INSTR) & 1))
        dma_piece(std::integral_constant<int, S ^ 1>{}, std::integral_constant<int, WF_RAW_INSTR + (g - WF_RAW_INSTR - 1) / 2>{});
      // A = filter fragment (rows = channels), B = transformed window (columns = tiles)
      if constexpr (FRESH && e == 0) {
        acc[x2] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb[pr & 1][0][e], cur[x2 >> 2][x2 & 3][e], zero16, 0, 0, 0);
        acc[x2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb[pr & 1][1][e], cur[(x2 + 1) >> 2][(x2 + 1) & 3][e], zero16, 0, 0, 0);
      } else {
        acc[x2] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb[pr & 1][0][e], cur[x2 >> 2][x2 & 3][e], acc[x2], 0, 0, 0);
        acc[x2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb[pr & 1][1][e], cur[(x2 + 1) >> 2][(x2 + 1) & 3][e], acc[x2 + 1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (DMA) first_issue = false;
  };

  // ---- epilogue of one item ----
  // An accumulator tile is [channel][tile]: lane (li, lh) holds tile li (tx = li & 3, ty = li >> 2) and, in register r, channel
  // n0 + 8 (r >> 2) + 4 lh + (r & 3): registers 4 rq .. 4 rq + 3 are four consecutive channels - one 16-byte store.  Output
  // pixel of Y[pp][qq]: (y0 + 2 ty + pp, x0 + 2 tx + qq).
  float* const scratch = reinterpret_cast<float*>(lds + WF_RAW_USED);      // + wave * (WF_RAW_WAVE / 4): 192 floats per wave
  auto epilogue = [&](const Item& it) {
    const int ld4 = p.ldout * 4;
    const bool interior = it.valid && (it.y0 + 16 <= p.H) && (it.x0 + 8 <= p.W);
    const int py = it.y0 + 2 * ty, px = it.x0 + 2 * tx;
    const unsigned vbase = (unsigned)((((it.b * p.H + py) * p.W + px) * p.ldout + it.n0 + 4 * lh) * 4);
    // validity of the lane's four pixels (bit pp * 2 + qq)
    unsigned okmask = 0xfu;
    if (!interior) {
      okmask = 0u;
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (it.valid && py + (o >> 1) < p.H && px + (o & 1) < p.W) okmask |= 1u << o;
    }
    bool want_stats = p.stats != nullptr;
#ifdef SSP_PROBES
    if (p.probe & 16) want_stats = false;
#endif
    __amdgpu_buffer_rsrc_t rs_x = rs_out;
    unsigned vbase_x = 0;
    int ldx4 = 0;
    if constexpr (BNB) {
      rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.bn_raw, 0, (int)WF_OOB, 0x00020000);
      vbase_x = (unsigned)((((it.b * p.H + py) * p.W + px) * p.bn_ld + it.n0 + 4 * lh) * 4);
      ldx4 = p.bn_ld * 4;
    }
    // Per-channel sums over the lane's pixels: statistics of the raw values, shifted by a sample of the channel (the value
    // of the half's first tile: the sums of (x - K) and (x - K)^2 do not cancel; mean = K + s / n, M2 = ss - s^2 / n) - or
    // the two BatchNorm-backward sums.
    float sa[16], sb[16], kk[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa[r] = 0.f; sb[r] = 0.f; kk[r] = 0.f; }
    // Y = A^T M A, one OUTPUT row pp at a time (2 + 2 vectors live, not 4 + 2): c0 / c1 = a row of the transform domain
    // folded over its columns, added into the output row with A^T's signs
    auto half = [&](auto pp_tag) {
      constexpr int pp = decltype(pp_tag)::value;
      f32x16 Ya, Yb;      // Y[pp][0], Y[pp][1]
      // One PLANE at a time: Y[pp][qq] = sum_i sum_j AT[pp][i] AT[qq][j] M[i][j], AT = [1 1 1 0; 0 1 -1 -1].  Each plane is
      // pinned in the accumulation registers by an empty volatile asm right before it is read, and the asm takes the running
      // sums as operands so that it cannot be scheduled ahead of the previous plane's adds: without it the compiler reads all
      // 256 accumulators at the top of the epilogue and spills half the register file.
#pragma unroll
      for (int r = 0; r < 16; ++r) { Ya[r] = 0.f; Yb[r] = 0.f; }
      wf_sfor<16>([&](auto XI) {
        constexpr int xi = decltype(XI)::value, i = xi >> 2, j = xi & 3;
        constexpr int wi = pp == 0 ? (i < 3 ? 1 : 0) : (i == 0 ? 0 : (i == 1 ? 1 : -1));      // AT[pp][i]
        constexpr int wa = j < 3 ? 1 : 0, wb = j == 0 ? 0 : (j == 1 ? 1 : -1);                // AT[0][j], AT[1][j]
        if constexpr (wi != 0) {
          asm volatile("" : "+a"(acc[xi]), "+v"(Ya), "+v"(Yb));
          if constexpr (wi * wa == 1) Ya += acc[xi];
          if constexpr (wi * wa == -1) Ya -= acc[xi];
          if constexpr (wi * wb == 1) Yb += acc[xi];
          if constexpr (wi * wb == -1) Yb -= acc[xi];
        }
      });
      asm volatile("" : "+v"(Ya), "+v"(Yb));
      if constexpr (pp == 0 && !BNB && !ACCUM) {
        // the statistics' shift: the value of the half's first tile (lane 0 / lane 32), the same for every lane of the half
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float k0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (float)Ya[r]), 0));
          const float k1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (float)Ya[r]), 32));
          kk[r] = lh ? k1 : k0;
        }
      }
      const bool okb0 = (okmask >> (pp * 2)) & 1u, okb1 = (okmask >> (pp * 2 + 1)) & 1u;
      const float ok0 = okb0 ? 1.f : 0.f, ok1 = okb1 ? 1.f : 0.f;
      unsigned v0 = okb0 ? vbase : WF_OOB, v1 = okb1 ? vbase : WF_OOB;
#ifdef SSP_PROBES
      if (p.probe & 64) { v0 = (unsigned)(((int)blockIdx.x * 256 + tid) * 16); v1 = v0; }      // all stores of a lane to ONE 16-byte slot (L2-resident)
#endif
      const unsigned x0_ = okb0 ? vbase_x : WF_OOB, x1_ = okb1 ? vbase_x : WF_OOB;
      const int so0 = pp * p.W * ld4, so1 = so0 + ld4;                 // scalar byte offsets of pixels (pp, 0), (pp, 1)
      const int sx0 = pp * p.W * ldx4, sx1 = sx0 + ldx4;
      wf_sfor<4>([&](auto RQ) {
        constexpr int rq = decltype(RQ)::value;
        f32x4 ya = {Ya[4 * rq], Ya[4 * rq + 1], Ya[4 * rq + 2], Ya[4 * rq + 3]};
        f32x4 yb = {Yb[4 * rq], Yb[4 * rq + 1], Yb[4 * rq + 2], Yb[4 * rq + 3]};
        if constexpr (!BNB && !ACCUM) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float da = (ya[k] - kk[4 * rq + k]) * ok0, db = (yb[k] - kk[4 * rq + k]) * ok1;
            sa[4 * rq + k] += da + db;
            sb[4 * rq + k] += da * da + db * db;
          }
        }
        if constexpr (AFFINE) {
          const int n = it.n0 + 8 * rq + 4 * lh;
          f32x4 b4 = {0.f, 0.f, 0.f, 0.f}, e4 = {1.f, 1.f, 1.f, 1.f};
          if (p.bias != nullptr) b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
          if (p.escale != nullptr) e4 = *reinterpret_cast<const f32x4*>(p.escale + n);
          ya = ya * e4 + b4;
          yb = yb * e4 + b4;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            ya[k] = ya[k] > 0.f ? ya[k] : ya[k] * p.act_slope;
            yb[k] = yb[k] > 0.f ? yb[k] : yb[k] * p.act_slope;
          }
        }
        if constexpr (ACCUM) {
          ya += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_out, v0, so0 + rq * 32, 0));
          yb += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_out, v1, so1 + rq * 32, 0));
        }
#ifdef SSP_PROBES
        if (!(p.probe & 1) || ya[0] == 12345.678f)
#endif
        {
#ifdef SSP_PROBES
          if (p.probe & 64) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ya), rs_out, v0, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, yb), rs_out, v1, 0, 0);
          } else
#endif
#ifdef SSP_PROBES
          if (p.probe & 128) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ya), rs_out, v0, so0 + rq * 32, 2);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, yb), rs_out, v1, so1 + rq * 32, 2);
          } else if (p.probe & 256) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ya), rs_out, v0, so0 + rq * 32, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, yb), rs_out, v1, so1 + rq * 32, 16);
          } else if (p.probe & 512) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ya), rs_out, v0, so0 + rq * 32, 17);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, yb), rs_out, v1, so1 + rq * 32, 17);
          } else
#endif
          {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ya), rs_out, v0, so0 + rq * 32, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, yb), rs_out, v1, so1 + rq * 32, 0);
          }
        }
        if constexpr (BNB) {
          const int n = it.n0 + 8 * rq + 4 * lh;
          const f32x4 xa = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, x0_, sx0 + rq * 32, 0));
          const f32x4 xb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, x1_, sx1 + rq * 32, 0));
          const f32x4 sc = *reinterpret_cast<const f32x4*>(p.bn_scale + n), sh = *reinterpret_cast<const f32x4*>(p.bn_shift + n);
          const f32x4 mu = *reinterpret_cast<const f32x4*>(p.bn_mean + n), is = *reinterpret_cast<const f32x4*>(p.bn_invstd + n);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float ga = ((xa[k] * sc[k] + sh[k]) > 0.f ? ya[k] : ya[k] * p.bn_slope) * ok0;
            const float gb = ((xb[k] * sc[k] + sh[k]) > 0.f ? yb[k] : yb[k] * p.bn_slope) * ok1;
            sa[4 * rq + k] += ga + gb;
            sb[4 * rq + k] += ga * ((xa[k] - mu[k]) * is[k]) + gb * ((xb[k] - mu[k]) * is[k]);
          }
        }
      });
    };
    half(std::integral_constant<int, 0>{});
    half(std::integral_constant<int, 1>{});
    // ---- per-channel sums over the wave's 32 tiles, then over the 4 waves through the scratch ----
    float* const my = scratch + wid * (WF_RAW_WAVE / 4);
    if constexpr (BNB) {
      const float t1 = wf_reduce16(sa, li), t2 = wf_reduce16(sb, li);
      const int r = ((li >> 4) & 1) * 8 + ((li >> 3) & 1) * 4 + ((li >> 2) & 1) * 2 + ((li >> 1) & 1);
      const int cl = 8 * (r >> 2) + 4 * lh + (r & 3);      // channel of the block this lane ended up with
      if (!(li & 1)) { my[cl * 2 + 0] = t1; my[cl * 2 + 1] = t2; }
      __syncthreads();
      if (tid < 32) {
        float a = 0.f, bsum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += scratch[w * (WF_RAW_WAVE / 4) + tid * 2 + 0]; bsum += scratch[w * (WF_RAW_WAVE / 4) + tid * 2 + 1]; }
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
        const float s = wf_reduce16(sa, li), ss = wf_reduce16(sb, li), k = wf_select16(kk, li);
        float cnt = (float)__builtin_popcount(okmask);      // valid pixels of the lane's tile, summed over the half's 32 tiles
        cnt += __shfl_xor(cnt, 16); cnt += __shfl_xor(cnt, 8); cnt += __shfl_xor(cnt, 4); cnt += __shfl_xor(cnt, 2); cnt += __shfl_xor(cnt, 1);
        const float mean = cnt > 0.f ? k + s / cnt : 0.f;
        const float m2 = cnt > 0.f ? fmaxf(ss - s * s / cnt, 0.f) : 0.f;
        const int r = ((li >> 4) & 1) * 8 + ((li >> 3) & 1) * 4 + ((li >> 2) & 1) * 2 + ((li >> 1) & 1);
        const int cl = 8 * (r >> 2) + 4 * lh + (r & 3);
        if (!(li & 1)) { my[cl * 3 + 0] = cnt; my[cl * 3 + 1] = mean; my[cl * 3 + 2] = m2; }
        __syncthreads();
        if (tid < 32) {
          float c_ = scratch[tid * 3 + 0], mu = scratch[tid * 3 + 1], sq = scratch[tid * 3 + 2];
#pragma unroll
          for (int w = 1; w < 4; ++w)
            chan_combine(c_, mu, sq, scratch[w * (WF_RAW_WAVE / 4) + tid * 3 + 0], scratch[w * (WF_RAW_WAVE / 4) + tid * 3 + 1],
                         scratch[w * (WF_RAW_WAVE / 4) + tid * 3 + 2]);
          float* st = p.stats + ((int64_t)it.tb * p.Cout + it.n0 + tid) * 2;
          st[0] = mu;
          st[1] = sq;
          if (tid == 0 && it.n0 == 0) p.stats[(int64_t)p.ntb * p.Cout * 2 + it.tb] = c_;      // the block's pixel count
        }
      }
    }
  };

  // ---- the pipeline ----
  // Ring slot of stage k of an item = k & 1 (Cin % 32 == 0: an item is a whole number of stage PAIRS), and every stage pair
  // runs the same four half-stages - one straight-line body, so that the register allocation of the hot loop has no merge
  // points to patch up with copies and spills (an earlier form with seven variants of the half-stage spilled 300 registers):
  //   slot 0, half 0: DMA of the next stage -> slot 1 between the MFMAs; window reads (slot 0, half 1)
  //   slot 0, half 1: [my raw patch of slot 1 has landed: vmcnt] window reads (slot 1, half 0)          | barrier
  //   slot 1, half 0: DMA of the stage after -> slot 0 (the NEXT ITEM's first stage behind an item's last); windows (1, 1)
  //   slot 1, half 1: [vmcnt] window reads (slot 0, half 0) - the same addresses for either successor     | barrier
  // Only the first half-stage of an item differs: its MFMAs start from zero, and (behind an epilogue) its DMA went out early.
  const int nsp = p.Cin >> 5;
  Item cur = decode(item);
  raw_offsets(cur, true);
  set_stage(0, cur.n0);
  issue_block(T0{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  wf_sfor<16>([&](auto K) {
    constexpr int i = decltype(K)::value >> 2, j = decltype(K)::value & 3;
    va[i][j] = *reinterpret_cast<const f32x4*>(lds + awin[i >> 1][0] + ((i & 1) * 640 + ((j & 1) * 5 + (j >> 1)) * 64));
  });
  wf_input_transform(va);
  bool after_epilogue = false;
  for (;;) {
    Item nxt = cur;
    bool has_next = false;
    for (int sp = 0; sp < nsp; ++sp) {
      const bool early = sp == 0 && after_epilogue;      // stage 1 went out in front of the previous item's epilogue
      if (!early) set_stage((2 * sp + 1) * 16, cur.n0);
      if (early) half_stage(va, vb, T0{}, T0{}, TT{}, TF{}, TT{}, T0{}, T1{});
      else if (sp == 0) half_stage(va, vb, T0{}, T0{}, TT{}, TT{}, TT{}, T0{}, T1{});
      else half_stage(va, vb, T0{}, T0{}, TF{}, TT{}, TT{}, T0{}, T1{});
      wf_input_transform(vb);
      // (inside: a wait for my raw patch of slot 1 - all but the WF_U_INSTR youngest pieces; behind an epilogue its 16 stores
      // are younger still)
      half_stage(vb, va, T0{}, T1{}, TF{}, TF{}, TT{}, T1{}, T0{}, early);
      wf_input_transform(va);
      if (early) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");      // the stores may fly on
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (sp + 1 < nsp) {
        set_stage((2 * sp + 2) * 16, cur.n0);
      } else {
        const int nitem = item + G;
        has_next = nitem < p.nitems;
        if (has_next) nxt = decode(nitem);
        raw_offsets(nxt, has_next);        // no next item: every offset out of range (the pieces fetch zeros)
        set_stage(0, nxt.n0);
      }
      half_stage(va, vb, T1{}, T0{}, TF{}, TT{}, TT{}, T1{}, T1{});
      wf_input_transform(vb);
      half_stage(vb, va, T1{}, T1{}, TF{}, TF{}, TT{}, T0{}, T0{});
      wf_input_transform(va);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    // The next item's SECOND stage goes out in front of this item's epilogue (slot 1 is free since the barrier above; rvoff
    // already holds the next item's offsets): the CU's vector-memory pipe is in order, and behind the epilogue's 64 KiB of
    // stores (~3 us to drain) the pieces would land too late for the next item's second half-stage.  This way the stores
    // drain under the next item's first stage, which waits for nothing younger than its own pieces.
    after_epilogue = has_next;
    if (has_next) {
      set_stage(16, nxt.n0);
      issue_block(T1{});
    }
#ifdef SSP_PROBES
    if (!(p.probe & 2) || acc[0][0] == 12345.678f)
#endif
    epilogue(cur);
    if (!has_next) break;
    item += G;
    cur = nxt;
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
  SSP_CHECK_ARG(a.ldin % 4 == 0 && a.ldout % 4 == 0 && (((uintptr_t)a.in) & 15) == 0 && (((uintptr_t)a.wt) & 15) == 0 &&
                    (((uintptr_t)a.out) & 15) == 0,
                "conv (on-chip Winograd plan): 16-byte aligned operands, ldin %% 4 == 0, ldout %% 4 == 0");
  SSP_CHECK_ARG((int64_t)B * H * W * a.ldin * 4 < (1ll << 31) && (int64_t)B * H * W * a.ldout * 4 < (1ll << 31) &&
                    (int64_t)16 * a.Cout * a.Cin * 4 < (1ll << 31) &&
                    (a.bn_partial == nullptr || (int64_t)B * H * W * a.bn_ld * 4 < (1ll << 31)),
                "conv (on-chip Winograd plan): operands beyond the 2 GiB buffer range");
  if (a.bn_partial != nullptr)
    SSP_CHECK_ARG(a.bn_ld % 4 == 0 && (((uintptr_t)a.bn_raw) & 15) == 0,
                  "conv (on-chip Winograd plan): bn_raw must be 16-byte aligned, bn_ld %% 4 == 0");
  if (a.bias != nullptr) SSP_CHECK_ARG((((uintptr_t)a.bias) & 15) == 0, "conv (on-chip Winograd plan): bias must be 16-byte aligned");
  if (a.escale != nullptr) SSP_CHECK_ARG((((uintptr_t)a.escale) & 15) == 0, "conv (on-chip Winograd plan): scale must be 16-byte aligned");
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
  SspProfScope prof(prof_kind == SSP_PROF_CONV_DGRAD ? SSP_PROF_ONCHIP_DGRAD : SSP_PROF_ONCHIP_FWD, stream,
                    2.0 * (double)a.M * a.Cout * 9.0 * a.Cin);      // algorithmic (direct) FLOPs
  switch (flags) {
    case 1: return wf_launch<1>(p, stream);
    case 2: return wf_launch<2>(p, stream);
    case 3: return wf_launch<3>(p, stream);
    case 4: return wf_launch<4>(p, stream);
    default: return wf_launch<0>(p, stream);
  }
}
