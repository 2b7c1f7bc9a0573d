// ConvArgs + epilogue shared by conv_igemm.hip (register-staged loader) and conv_igemm_dma.hip (LDS-direct loader).
#pragma once
#include <type_traits>

#include "ssp_common.h"

struct ConvArgs {
  const float* in;
  const float* wt;
  float* out;
  const float* bias;  // [Cout] or nullptr (added in the epilogue; the linear head conv, or the BN shift in eval mode)
  const float* escale; // [Cout] or nullptr: per-channel factor applied before the bias (eval-mode BatchNorm folded in)
  float act_slope;    // leaky-ReLU slope applied last (1 = no activation)
  float* stats;       // [ntile_m][Cout][2] = per-M-tile (mean, M2) of the raw output, or nullptr
  int H, W, Cin, Cout, ldin, ldout, R, M;
  int accumulate;     // out += result (second consumer of a routed activation in dgrad)
  int ntile_m, ntile_n;
  int xcd_remap;
  float* ws;           // split-K partial tiles [split][ws_rows][Cout], row m stored at m - ws_row0
  int ksplit, it_per_split;
  // hybrid launch (LDS-direct kernel): tiles [0, tail_begin) run un-split and finish in their epilogue - whole resident
  // waves of workgroups - and the remaining tiles, which would leave the last wave mostly idle, split their K loop
  // tail_ks ways (partials to the workspace, summed by splitk_reduce_kernel over rows >= ws_row0).  tail_ks = 0: off.
  int tail_begin, tail_ks, tail_it_per_split;
  int ws_row0, ws_rows;
  int col_major;       // tile order inside a launch: tile_m fastest (workgroups resident on one XCD share a filter slab)
  SspFastDiv divW, divH;   // m -> (x, y) of a tile row without run-time integer division
  int probe;           // 2: generic (predicated) epilogue everywhere (A/B switch, same results); 1: SSP_PROBES builds only
  // Data-gradient launches only: BatchNorm-backward reductions of the block that PRODUCED the activation whose gradient
  // this launch writes (out = g = dL/d leaky(BN(raw))).  With bn_partial set, the finishing pass (tile epilogue or
  // splitk_reduce_kernel) also reads that block's raw conv output at the tile's positions and leaves, per M tile and
  // channel, (sum dy, sum dy * xhat) with dy = g * leaky'(scale*raw+shift), xhat = (raw-mean)*invstd - what
  // bn_act_bwd_reduce_kernel would otherwise re-read both g and raw for (darknet.py:157,162 autograd).
  const float* bn_raw;
  const float* bn_scale;
  const float* bn_shift;
  const float* bn_mean;
  const float* bn_invstd;
  float* bn_partial;   // [min(ntile_m, bn_nslot)][Cout][2] or nullptr
  int bn_nslot;        // launches with more M tiles than rows fold tile t into row t % bn_nslot with fp32 atomics (the rows
                       // must be zero on entry); fewer tiles: one plain store per (tile, channel), deterministic
  int bn_ld;
  float bn_slope;
  // Batched GEMM (the 16 transform positions of a Winograd layer, conv_wino.hip): gridDim.y problems that share the shape;
  // problem b reads in + b * batch_in, wt + b * batch_wt and writes out + b * batch_out (floats).  batch = 0: one problem.
  // Plain stores only (no statistics, bias, split-K or fused reductions in a batched launch).
  int batch;
  int64_t batch_in, batch_wt, batch_out;
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


// Shared epilogue of the implicit-GEMM kernels.  C/D map of the 32x32 MFMA: col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  `smem` must be free (all LDS traffic of the K loop retired).
template <int BM, int BN, int WM, int WN, int NT>
__device__ __forceinline__ void igemm_epilogue(const ConvArgs& p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* smem,
                                               int m0, int n0, int tile_m, int split, int tid, bool partial,
                                               int64_t out_off = 0) {
  float* const outp = p.out + out_off;      // batched launches: this problem's output plane
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int li = lane & 31, lh = lane >> 5;
  if (partial) {
    // split-K: raw partial tile to the workspace; bias / accumulate / BN statistics happen in splitk_reduce_kernel
    float* wsp = p.ws + ((int64_t)split * p.ws_rows - p.ws_row0) * p.Cout;
#if defined(__HIP_DEVICE_COMPILE__)
    if (m0 + BM <= p.M && (int64_t)BM * p.Cout * 4 < (1ll << 31) && p.probe != 2) {
      // lean form (see body_fast below): buffer stores, fixed lane offset, scalar row steps
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(wsp + (int64_t)m0 * p.Cout), 0,
                                                                          BM * p.Cout * 4, 0x00020000);
      const int ld4 = p.Cout * 4;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 32 + li;
        const unsigned voff = (n < p.Cout) ? (unsigned)(((wm * WTM + 4 * lh) * p.Cout + n) * 4) : 0x80000000u;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          int so = i * 32 * ld4;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            asm volatile("" : "+s"(so));
            // (copy to a scalar first: __builtin_bit_cast applied directly to an ext-vector ELEMENT picked element 0 for
            // every r with this compiler - the first version of this path stored acc[i][j][0] sixteen times)
            const float v = acc[i][j][r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, so, 0);
            so += ((r & 3) == 3) ? 5 * ld4 : ld4;
          }
        }
      }
      return;
    }
#endif
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WTN + j * 32 + li;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < p.M && n < p.Cout) wsp[(int64_t)m * p.Cout + n] = acc[i][j][r];
        }
    }
    return;
  }
  auto body = [&](auto accum_tag, auto bnb_tag) {
    constexpr bool ACCUM = decltype(accum_tag)::value;
    constexpr bool BNB = decltype(bnb_tag)::value;     // fused BatchNorm-backward reductions (data-gradient launches)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WTN + j * 32 + li;
      const bool n_ok = n < p.Cout;
      const float bias = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
      const float esc = (p.escale != nullptr && n_ok) ? p.escale[n] : 1.f;
      float cnt = 0.f, sum = 0.f;
      float b_sc = 0.f, b_sh = 0.f, b_mu = 0.f, b_is = 0.f, s1 = 0.f, s2 = 0.f;
      if constexpr (BNB) {
        if (n_ok) { b_sc = p.bn_scale[n]; b_sh = p.bn_shift[n]; b_mu = p.bn_mean[n]; b_is = p.bn_invstd[n]; }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float xr[16];
        if constexpr (BNB) {
          // the producing block's raw conv output at this lane's 16 rows: issued together, ahead of the stores
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            xr[r] = (m < p.M && n_ok) ? p.bn_raw[(int64_t)m * p.bn_ld + n] : 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          int m = m0 + row;
          float v = acc[i][j][r] * esc + bias;
          v = v > 0.f ? v : v * p.act_slope;
          if (m < p.M && n_ok) {
            float* o = outp + (int64_t)m * p.ldout + n;
            if constexpr (ACCUM) v += *o;
            *o = v;
            cnt += 1.f;
            sum += acc[i][j][r];
            if constexpr (BNB) {
              const float y = xr[r] * b_sc + b_sh;
              const float dyv = y > 0.f ? v : v * p.bn_slope;
              s1 += dyv;
              s2 += dyv * ((xr[r] - b_mu) * b_is);
            }
          }
        }
      }
      if constexpr (BNB) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        const int col = wn * WTN + j * 32 + li;
        if (lh == 0) {
          smem[(wm * BN + col) * 2 + 0] = s1;
          smem[(wm * BN + col) * 2 + 1] = s2;
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
        float* red = smem;  // [WM][BN][3]
        int col = wn * WTN + j * 32 + li;
        if (lh == 0) {
          red[(wm * BN + col) * 3 + 0] = cnt;
          red[(wm * BN + col) * 3 + 1] = mean;
          red[(wm * BN + col) * 3 + 2] = m2;
        }
      }
    }
  };
  // ---- lean form for tiles whose BM rows all exist (every tile but the last tile row) ----
  // The fp32 MFMA runs on the SIMD's vector lanes: every VALU instruction of the epilogue is matrix time lost, and the
  // generic form above spends ~20 of them per stored element (64-bit address arithmetic and an exec-mask branch for the
  // m < M && n < Cout test): ~5 k VALU cycles per wave against 37 k - 74 k cycles of MFMA in the K = 288 / 576 layers.
  // Here a tile is addressed through a buffer descriptor on its first row: the lane's byte offset is fixed (columns
  // beyond Cout carry an out-of-range offset and are dropped by the buffer unit), the row steps are SCALAR offsets
  // (SALU), so a stored element costs one buffer_store_dword plus - only when there is an affine / activation /
  // accumulate / BatchNorm-backward term - the arithmetic itself.
  auto body_fast = [&](auto accum_tag, auto bnb_tag, auto ident_tag) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool ACCUM = decltype(accum_tag)::value;
    constexpr bool BNB = decltype(bnb_tag)::value;
    constexpr bool IDENT = decltype(ident_tag)::value;     // no bias, no per-channel scale, no activation
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(outp + (int64_t)m0 * p.ldout), 0, BM * p.ldout * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x = rs_o;
    if constexpr (BNB)
      rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bn_raw + (int64_t)m0 * p.bn_ld), 0, BM * p.bn_ld * 4, 0x00020000);
    const int ld4 = p.ldout * 4, ldx4 = p.bn_ld * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WTN + j * 32 + li;
      const bool n_ok = n < p.Cout;
      const unsigned voff = n_ok ? (unsigned)(((wm * WTM + 4 * lh) * p.ldout + n) * 4) : 0x80000000u;
      float bias = 0.f, esc = 1.f;
      if constexpr (!IDENT) {
        bias = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
        esc = (p.escale != nullptr && n_ok) ? p.escale[n] : 1.f;
      }
      float b_sc = 0.f, b_sh = 0.f, b_mu = 0.f, b_is = 0.f, s1 = 0.f, s2 = 0.f;
      unsigned voffx = 0x80000000u;
      if constexpr (BNB) {
        if (n_ok) {
          b_sc = p.bn_scale[n]; b_sh = p.bn_shift[n]; b_mu = p.bn_mean[n]; b_is = p.bn_invstd[n];
          voffx = (unsigned)(((wm * WTM + 4 * lh) * p.bn_ld + n) * 4);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        // Row offsets walk in a running SCALAR (rows r&3 + 8*(r>>2): steps of 1,1,1,5 rows).  The empty asm makes the
        // value opaque, so the compiler keeps ONE live SGPR and an s_add per element instead of hoisting all 16 * TM *
        // TN products in front of the four code paths (which spilled ~300 SGPRs into VGPR lanes: v_readlane per store).
        float xr[16], prev[16];
        if constexpr (BNB) {
          int so = i * 32 * ldx4;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            asm volatile("" : "+s"(so));
            xr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, voffx, so, 0));
            so += ((r & 3) == 3) ? 5 * ldx4 : ldx4;
          }
        }
        if constexpr (ACCUM) {
          int so = i * 32 * ld4;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            asm volatile("" : "+s"(so));
            prev[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_o, voff, so, 0));
            so += ((r & 3) == 3) ? 5 * ld4 : ld4;
          }
        }
        int so = i * 32 * ld4;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r];
          if constexpr (!IDENT) {
            v = v * esc + bias;
            v = v > 0.f ? v : v * p.act_slope;
          }
          if constexpr (ACCUM) v += prev[r];
          asm volatile("" : "+s"(so));
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_o, voff, so, 0);
          so += ((r & 3) == 3) ? 5 * ld4 : ld4;
          if constexpr (BNB) {
            const float y = xr[r] * b_sc + b_sh;
            const float dyv = y > 0.f ? v : v * p.bn_slope;
            s1 += dyv;
            s2 += dyv * ((xr[r] - b_mu) * b_is);
          }
        }
      }
      if constexpr (BNB) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        const int col = wn * WTN + j * 32 + li;
        if (lh == 0) {
          smem[(wm * BN + col) * 2 + 0] = s1;
          smem[(wm * BN + col) * 2 + 1] = s2;
        }
      }
      if (p.stats != nullptr) {
        // every one of this lane's 16 * TM rows exists: count is a constant
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
        float cnt = n_ok ? (float)(16 * TM) : 0.f;
        float mean = n_ok ? sum * (1.f / (float)(16 * TM)) : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float d = acc[i][j][r] - mean;
            m2 += d * d;
          }
        if (!n_ok) m2 = 0.f;
        float ocnt = __shfl_xor(cnt, 32), omean = __shfl_xor(mean, 32), om2 = __shfl_xor(m2, 32);
        chan_combine(cnt, mean, m2, ocnt, omean, om2);
        const int col = wn * WTN + j * 32 + li;
        if (lh == 0) {
          smem[(wm * BN + col) * 3 + 0] = cnt;
          smem[(wm * BN + col) * 3 + 1] = mean;
          smem[(wm * BN + col) * 3 + 2] = m2;
        }
      }
    }
#endif
  };
  const bool interior = (m0 + BM <= p.M) && ((int64_t)BM * max(p.ldout, p.bn_ld) * 4 < (1ll << 31)) && p.probe != 2;
  const bool ident = p.bias == nullptr && p.escale == nullptr && p.act_slope == 1.f;
  if (interior) {
    if (p.bn_partial != nullptr) body_fast(std::false_type{}, std::true_type{}, std::true_type{});   // dgrad: plain output
    else if (p.accumulate && ident) body_fast(std::true_type{}, std::false_type{}, std::true_type{});
    else if (p.accumulate) body_fast(std::true_type{}, std::false_type{}, std::false_type{});
    else if (ident) body_fast(std::false_type{}, std::false_type{}, std::true_type{});
    else body_fast(std::false_type{}, std::false_type{}, std::false_type{});
  } else if (p.bn_partial != nullptr) body(std::false_type{}, std::true_type{});       // never with accumulate (checked on the host)
  else if (p.accumulate) body(std::true_type{}, std::false_type{});
  else body(std::false_type{}, std::false_type{});
  if (p.bn_partial != nullptr) {
    __syncthreads();
    for (int col = tid; col < BN; col += NT) {
      const int n = n0 + col;
      if (n < p.Cout) {
        float a = smem[col * 2 + 0], b = smem[col * 2 + 1];
#pragma unroll
        for (int w = 1; w < WM; ++w) { a += smem[(w * BN + col) * 2 + 0]; b += smem[(w * BN + col) * 2 + 1]; }
        if (p.ntile_m > p.bn_nslot) {
          float* dst = p.bn_partial + ((int64_t)(tile_m % p.bn_nslot) * p.Cout + n) * 2;
          atomicAdd(dst, a);
          atomicAdd(dst + 1, b);
        } else {
          float* dst = p.bn_partial + ((int64_t)tile_m * p.Cout + n) * 2;
          dst[0] = a;
          dst[1] = b;
        }
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
        float* st = p.stats + ((int64_t)tile_m * p.Cout + n) * 2;
        st[0] = mean;
        st[1] = m2;
      }
    }
  }
}
