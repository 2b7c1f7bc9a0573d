// Filter gradient of a 3x3 layer in the Winograd F(2x2, 3x3) domain with BOTH transforms done on the chip
// (ssp_conv_wgrad_wino_t with tile = SSP_WINO_WGRAD_FUSED; the forward / data-gradient counterpart is conv_wino_fused.hip).
//
//   dU_xi [Cout][Cin] = sum over the tiles t of  dM_xi[t][Cout] (x) V_xi[t][Cin],   dM = A dY A^T,  V = B^T d B,  xi = 0 .. 15
//   dw [Cout][3][3][Cin] += G^T dU G                                                  (wino_wgrad_finish_kernel<2>)
//
// conv_wino.hip runs this as four launches around 2 x 4 floats per pixel and channel of HBM traffic (V and dM written and
// read): on the wide maps - 32 -> 64 channels at 208 x 208, 64 -> 128 at 104 x 104 - the transforms cost more than the GEMM.
// Here the contraction index is the TILE: a lane of the A operand holds (output channel li, tile 2 s + lh), a lane of the B
// operand (input channel li, tile 2 s + lh) - so a lane simply LOADS the 2 x 2 output-gradient pixels / the 4 x 4 input window
// of its tile at its channel (32 lanes = 128 contiguous bytes per pixel), transforms them in registers, and the 16 results are
// its operands of the 16 planes' MFMAs.  No LDS at all.  A wave owns a 32 x 32 (Cout x Cin) block of all 16 planes (256
// accumulator registers) and a CHUNK of tile rows; the launch is waves = blocks x chunks, and every wave ends with fp32
// atomic adds of its 16 tiles into dU (zeroed in front of the launch).
//
// Measured (profiles/r06_pmc_conv.txt): the matrix pipe is busy 45 % of the launch - a wave is alone on its SIMD and the
// compiler runs each k-step as three blocks (transforms, 16 MFMAs, 20 loads + scalar address work), so nothing overlaps the
// MFMAs; dealing the loads between the MFMAs by hand (scheduling barriers) made the register allocator spill the
// accumulators (15 ms); splitting the 16 planes over TWO waves per SIMD (128 accumulator registers each, the other wave's
// loads under this wave's MFMAs) was correct and 10 % SLOWER (777 / 755 us against 702 / 681 us on layers 2 / 4): both
// halves load the same 20 values, and 40 dword loads per SIMD and k-step is what bounds it - the per-lane dword loads
// (lane = channel) are the limit of this form, not the overlap.  Still 1.3 - 1.6 x faster than the direct kernel on the
// 208 x 208 / 104 x 104 layers.
//
// Loads run three k-steps (six tiles) ahead of the MFMAs in a ring of three register sets - the vmcnt counter allows 63
// operations in flight, a k-step is 12 (20 at a row's first).  Zero padding and ragged edges: out-of-range buffer offsets (rows of the image in the
// per-row lane offsets, columns only on the first / last k-step of a row).
#include <utility>

#include "conv_wino.h"

#define WG_OOB 0x80000000u

namespace {
struct WinoWgradFusedArgs {
  const float* dy;
  const float* x;
  float* dU;            // [16][Cout][Cin], zero on entry
  int B, H, W, Cin, Cout, lddy, ldx;
  int th, ksr;          // tile rows per image, k-steps (tile pairs) per tile row
  int nrows, rpc;       // tile rows in all (B * th), rows per chunk
  int nblk, nib;        // 32 x 32 blocks (Cout / 32 * Cin / 32), blocks along Cin
  int nunits;           // chunks * nblk
  SspFastDiv div_nblk, div_nib, div_th;
};

template <typename F, int... Is>
__device__ __forceinline__ void wg_sfor_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void wg_sfor(F&& f) {
  wg_sfor_impl(f, std::make_integer_sequence<int, N>{});
}
}  // namespace

__global__ void __launch_bounds__(256, 1) wino2_wgrad_fused_kernel(WinoWgradFusedArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int unit = (int)blockIdx.x * 4 + wid;
  if (unit >= p.nunits) return;                     // (no barriers in this kernel: waves are independent)
  const unsigned chunk = ssp_div((unsigned)unit, p.div_nblk);
  const int blk = unit - (int)chunk * p.nblk;
  const unsigned cb = ssp_div((unsigned)blk, p.div_nib);
  const int n0 = (int)cb * 32, c0 = (blk - (int)cb * p.nib) * 32;
  const int row0 = (int)chunk * p.rpc;
  const int row1 = min(p.nrows, row0 + p.rpc);
  if (row0 >= row1) return;

  // x is addressed from one row and one pixel BEFORE its first pixel, so that every window offset is non-negative; positions
  // outside the image carry an out-of-range lane offset instead (the buffer unit returns zeros)
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - (int64_t)(p.W + 1) * p.ldx), 0, (int)WG_OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)WG_OOB, 0x00020000);
  const int ldx4 = p.ldx * 4, ldy4 = p.lddy * 4;
  // k-step s of a tile row multiplies tiles s (lanes lh = 0) and s + ksr (lanes lh = 1): a lane walks CONSECUTIVE tiles, whose
  // windows share two of their four columns - only the two new columns are loaded (12 loads per k-step instead of 20: the
  // per-lane dword loads are what bounds this kernel) and their row-transformed values are carried over in registers
  const unsigned lane_x = (unsigned)((lh * 2 * p.ksr * p.ldx + c0 + li) * 4);
  const unsigned lane_y = (unsigned)((lh * 2 * p.ksr * p.lddy + n0 + li) * 4);
  const int tw = (p.W + 1) >> 1;

  f32x16 acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

  // ---- load cursor: (tile row, k-step) of the NEXT set to fetch, with the row's scalars and per-row lane offsets ----
  int lrow = row0, ls = 0;
  int l_sox = 0, l_soy = 0;                 // scalar byte offsets of the row's first window pixel / first output pixel
  unsigned vx[4], vy[2];                    // lane offsets of window row i / output row pp: out of range when the row is outside
  auto set_row = [&]() {
    const unsigned b = ssp_div((unsigned)lrow, p.div_th);
    const int ty = lrow - (int)b * p.th;
    // window row i is image row 2 ty - 1 + i; from the shifted base that is row 2 ty + i, pixel column 4 s + j
    l_sox = ((int)b * p.H + 2 * ty) * p.W * ldx4;
    l_soy = ((int)b * p.H + 2 * ty) * p.W * ldy4;
#pragma unroll
    for (int i = 0; i < 4; ++i) vx[i] = ((unsigned)(2 * ty - 1 + i) < (unsigned)p.H) ? lane_x : WG_OOB;
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) vy[pp] = (2 * ty + pp < p.H) ? lane_y : WG_OOB;
  };
  float rd[3][4], rx[3][16];                // raw sets in flight: the 2 x 2 output-gradient pixels, the window columns (all four at a
                                            // row's first k-step, else the two new ones: j = 2, 3)
  auto issue = [&](auto k_tag) {
    constexpr int K = decltype(k_tag)::value;
    if (lrow >= row1) return;
    // columns may fall outside the image at a row's first k-step (lanes lh = 0: column -1) and at its last ones (lanes lh = 1)
    const bool edge = ls == 0 || ls + p.ksr >= tw - 1;
    const int sx = l_sox + ls * 2 * ldx4, sy = l_soy + ls * 2 * ldy4;
    const int xcol = 2 * (ls + lh * p.ksr);            // first output column of this lane's tile
    if (ls == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const unsigned v = ((unsigned)(xcol - 1 + j) < (unsigned)p.W) ? vx[i] : WG_OOB;
          rx[K][i * 4 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, v, sx + (i * p.W + j) * ldx4, 0));
        }
    }
    if (edge) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 2; j < 4; ++j) {
          const unsigned v = ((unsigned)(xcol - 1 + j) < (unsigned)p.W) ? vx[i] : WG_OOB;
          rx[K][i * 4 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, v, sx + (i * p.W + j) * ldx4, 0));
        }
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const unsigned v = (xcol + qq < p.W) ? vy[pp] : WG_OOB;
          rd[K][pp * 2 + qq] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_y, v, sy + (pp * p.W + qq) * ldy4, 0));
        }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 2; j < 4; ++j)
          rx[K][i * 4 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, vx[i], sx + (i * p.W + j) * ldx4, 0));
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
          rd[K][pp * 2 + qq] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_y, vy[pp], sy + (pp * p.W + qq) * ldy4, 0));
    }
    if (++ls == p.ksr) {
      ls = 0;
      if (++lrow < row1) set_row();
    }
  };
  // ---- one k-step: transforms in registers, 16 MFMAs (A = dM: rows = output channels, B = V: columns = input channels) ----
  int cs = 0;                               // k-step of the set being consumed (0 = a row's first: all four columns are fresh)
  float tc[4][2];                           // row-transformed window columns 2, 3 of the previous k-step = columns 0, 1 of this one
  auto consume = [&](auto k_tag) {
    constexpr int K = decltype(k_tag)::value;
    // dM = A dY A^T, A = [1 0; 1 1; 1 -1; 0 -1]
    const float d00 = rd[K][0], d01 = rd[K][1], d10 = rd[K][2], d11 = rd[K][3];
    const float s0 = d00 + d10, m0 = d00 - d10, s1 = d01 + d11, m1 = d01 - d11;
    float dm[16];
    dm[0] = d00;  dm[1] = d00 + d01;  dm[2] = d00 - d01;  dm[3] = -d01;
    dm[4] = s0;   dm[5] = s0 + s1;    dm[6] = s0 - s1;    dm[7] = -s1;
    dm[8] = m0;   dm[9] = m0 + m1;    dm[10] = m0 - m1;   dm[11] = -m1;
    dm[12] = -d10; dm[13] = -d10 - d11; dm[14] = d11 - d10; dm[15] = d11;
    // V = B^T d B, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: the row step per window column (columns 0, 1 carried over
    // from the previous tile unless this is the row's first), then the column step
    float v[16];
    auto rowstep = [&](int j) {
      const float a0 = rx[K][j] - rx[K][8 + j], a1 = rx[K][4 + j] + rx[K][8 + j], a2 = rx[K][8 + j] - rx[K][4 + j],
                  a3 = rx[K][4 + j] - rx[K][12 + j];
      v[j] = a0; v[4 + j] = a1; v[8 + j] = a2; v[12 + j] = a3;
    };
    if (cs == 0) {
      rowstep(0);
      rowstep(1);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[4 * i] = tc[i][0]; v[4 * i + 1] = tc[i][1]; }
    }
    rowstep(2);
    rowstep(3);
#pragma unroll
    for (int i = 0; i < 4; ++i) { tc[i][0] = v[4 * i + 2]; tc[i][1] = v[4 * i + 3]; }
    if (++cs == p.ksr) cs = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a0 = v[4 * i] - v[4 * i + 2], a1 = v[4 * i + 1] + v[4 * i + 2], a2 = v[4 * i + 2] - v[4 * i + 1],
                  a3 = v[4 * i + 1] - v[4 * i + 3];
      v[4 * i] = a0; v[4 * i + 1] = a1; v[4 * i + 2] = a2; v[4 * i + 3] = a3;
    }
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(dm[xi], v[xi], acc[xi], 0, 0, 0);
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;

#pragma unroll
  for (int i = 0; i < 4; ++i) { tc[i][0] = 0.f; tc[i][1] = 0.f; }
  const int nsteps = (row1 - row0) * p.ksr;
  set_row();
  issue(K0{});
  issue(K1{});
  issue(K2{});
  int step = 0;
  for (; step + 3 <= nsteps; step += 3) {
    consume(K0{}); issue(K0{});
    consume(K1{}); issue(K1{});
    consume(K2{}); issue(K2{});
  }
  if (step < nsteps) { consume(K0{}); ++step; }
  if (step < nsteps) { consume(K1{}); ++step; }

  // ---- flush: accumulator tile [output channel row][input channel li] of every plane, added into dU ----
  // register r of a 32 x 32 tile: row (r & 3) + 8 (r >> 2) + 4 lh
  float* const base = p.dU + ((int64_t)(n0 + 4 * lh) * p.Cin + c0 + li);
  const int64_t plane = (int64_t)p.Cout * p.Cin;
  wg_sfor<16>([&](auto XI) {
    constexpr int xi = decltype(XI)::value;
    asm volatile("" : "+a"(acc[xi]));      // one plane at a time out of the accumulation registers (see conv_wino_fused.hip)
#pragma unroll
    for (int r = 0; r < 16; ++r) atomicAdd(base + xi * plane + (int64_t)((r & 3) + 8 * (r >> 2)) * p.Cin, acc[xi][r]);
  });
#endif
}

__global__ void __launch_bounds__(256) wino_wgrad_fused_zero_kernel(float4* __restrict__ p, size_t n16) {
  const float4 z = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = z;
}

int64_t ssp_wino_wgrad_fused_ws_floats(int Cin, int Cout) { return (int64_t)16 * Cin * Cout; }
bool ssp_wino_wgrad_fused_fits(int B, int H, int W, int Cin, int Cout) {
  return Cin % 32 == 0 && Cout % 32 == 0 && Cin >= 32 && Cout >= 32 && (int64_t)B * H * W < (1ll << 28);
}

int ssp_wino_wgrad_fused_launch(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int lddy,
                                int ldx, float* ws, int64_t ws_floats, hipStream_t stream) {
  SSP_CHECK_ARG(ssp_wino_wgrad_fused_fits(B, H, W, Cin, Cout), "wgrad (on-chip Winograd): needs Cin %% 32 == 0 and Cout %% 32 == 0");
  SSP_CHECK_ARG(x != nullptr && dy != nullptr && dw != nullptr && ldx >= Cin && lddy >= Cout,
                "wgrad (on-chip Winograd): needs the layer input (no shared transformed input), ldx >= Cin, lddy >= Cout");
  SSP_CHECK_ARG(ws != nullptr && ws_floats >= ssp_wino_wgrad_fused_ws_floats(Cin, Cout) && (((uintptr_t)ws) & 15) == 0 &&
                    (((uintptr_t)dw) & 15) == 0,
                "wgrad (on-chip Winograd): needs an aligned workspace of %lld floats", (long long)ssp_wino_wgrad_fused_ws_floats(Cin, Cout));
  SSP_CHECK_ARG(((int64_t)B * H * W + W + 1) * ldx * 4 < (1ll << 31) && (int64_t)B * H * W * lddy * 4 < (1ll << 31),
                "wgrad (on-chip Winograd): operands beyond the 2 GiB buffer range");
  WinoWgradFusedArgs p;
  p.dy = dy; p.x = x; p.dU = ws;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.lddy = lddy; p.ldx = ldx;
  p.th = (H + 1) / 2;
  const int tw = (W + 1) / 2;
  p.ksr = (tw + 1) / 2;
  p.nrows = B * p.th;
  p.nib = Cin / 32;
  p.nblk = (Cout / 32) * p.nib;
  // chunks of whole tile rows: as many waves as the chip holds (one per SIMD: 1024)
  int slots = 1024;
  {
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0)
      slots = ncu * 4;
  }
  int nchunk = slots / p.nblk;        // ONE resident round of waves: fewer, longer chunks = fewer atomic flushes
  if (nchunk > p.nrows) nchunk = p.nrows;
  if (nchunk < 1) nchunk = 1;
  p.rpc = (p.nrows + nchunk - 1) / nchunk;
  nchunk = (p.nrows + p.rpc - 1) / p.rpc;
  p.nunits = nchunk * p.nblk;
  p.div_nblk = ssp_fastdiv((unsigned)p.nblk); p.div_nib = ssp_fastdiv((unsigned)p.nib); p.div_th = ssp_fastdiv((unsigned)p.th);
  SspProfScope prof(SSP_PROF_ONCHIP_WGRAD, stream, 2.0 * (double)B * H * W * Cout * 9.0 * Cin);      // algorithmic (direct) FLOPs
  const size_t n16 = (size_t)ssp_wino_wgrad_fused_ws_floats(Cin, Cout) / 4;
  const unsigned zb = (unsigned)((n16 + 1023) / 1024 < 4096 ? (n16 + 1023) / 1024 : 4096);
  hipLaunchKernelGGL(wino_wgrad_fused_zero_kernel, dim3(zb), dim3(256), 0, stream, reinterpret_cast<float4*>(ws), n16);
  SSP_CHECK_LAUNCH("wino_wgrad_fused(zero)");
  hipLaunchKernelGGL(wino2_wgrad_fused_kernel, dim3((unsigned)((p.nunits + 3) / 4)), dim3(256), 0, stream, p);
  SSP_CHECK_LAUNCH("wino2_wgrad_fused");
  return ssp_wino_wgrad_finish_launch(ws, dw, Cout, Cin, 2, stream);
}
