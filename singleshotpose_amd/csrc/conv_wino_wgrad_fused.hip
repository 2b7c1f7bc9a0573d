// Filter gradient of a 3x3 layer in the Winograd F(2x2, 3x3) domain with BOTH transforms done on the chip
// (ssp_conv_wgrad_wino_t with tile = SSP_WINO_WGRAD_FUSED; the forward / data-gradient counterpart is conv_wino_fused.hip).
//
//   dU_xi [Cout][Cin] = sum over the tiles t of  dM_xi[t][Cout] (x) V_xi[t][Cin],   dM = A dY A^T,  V = B^T d B,  xi = 0 .. 15
//   dw [Cout][3][3][Cin] += G^T dU G                                                  (in the flush of every wave)
//
// conv_wino.hip runs this as four launches around 2 x 4 floats per pixel and channel of HBM traffic (V and dM written and
// read): on the wide maps - 32 -> 64 channels at 208 x 208, 64 -> 128 at 104 x 104 - the transforms cost more than the GEMM.
// Here the contraction index is the TILE: a lane of the A operand holds (output channel li, tile u + lh ksr), a lane of the B
// operand (input channel li, same tile) - so a lane simply LOADS the 2 x 2 output-gradient pixels / the new columns of the
// 4 x 4 input window of its tile at its channel (32 lanes = 128 contiguous bytes per pixel), transforms them in registers, and
// the 16 results are its operands of the 16 planes' MFMAs.  No LDS at all.  A wave owns a 32 x 32 (Cout x Cin) block of all
// 16 planes (256 accumulator registers) and a CHUNK of tile rows; the launch is waves = blocks x chunks (one per SIMD), and
// every wave ends by back-transforming its own sums (G^T dU G is linear) and adding nine taps per channel pair into dw.
//
// What bounds it, measured (probes of this file under SSP_PROBES): of 605 us on layer 4, 350 are the
// MFMAs (702 steps x 16 x 64 cycles at the ~2.05 GHz the chip sustains), ~100 the transforms / load issue / cursor (the fp32
// MFMA runs on the vector lanes: VALU work adds to it instead of hiding under it), 29 the flush and ~45 were the zeroing and
// finishing launches of the first version (dU in memory: gone).  Before the loads were taken out of the compiler's hands
// (wg_load) the same loop was latency-bound: 680-700 us, 45 % MFMA-busy.
// Tried and dropped: dealing the loads between the MFMAs by scheduling barriers (the allocator spilled the accumulators:
// 15 ms); the 16 planes split over TWO waves per SIMD (correct, 10 % slower: both halves load the same values); ring depths
// 4 and 5 (the same time as 3: with exact waits the loads are not the bound); a branch-free cursor (descriptor selects: the
// compiler turned them into waterfall loops and moved the accumulators between iterations).
//
// Zero padding and ragged edges: out-of-range buffer offsets (rows of the image in the per-row lane offsets, columns on the
// primer / last steps of a row).
#include <utility>

#include "conv_wino.h"

#define WG_OOB 0x80000000u

namespace {
struct WinoWgradFusedArgs {
  const float* dy;
  const float* x;
  float* dw;            // [Cout][3][3][Cin], accumulated into
  int B, H, W, Cin, Cout, lddy, ldx;
  int th, ksr;          // tile rows per image, k-steps (tile pairs) per tile row
  int nrows, rpc;       // tile rows in all (B * th), rows per chunk
  int nblk, nib;        // 32 x 32 blocks (Cout / 32 * Cin / 32), blocks along Cin
  int nunits;           // chunks * nblk
  SspFastDiv div_nblk, div_nib, div_th;
  int probe;            // SSP_PROBES builds only (timing probes, WRONG results): 4 no transforms, 8 no loads after the prologue, 16 no MFMAs, 32 no flush
};

typedef unsigned int wg_u32x4 __attribute__((ext_vector_type(4)));

// raw buffer descriptor (2 GiB of records: out-of-range lane offsets read zeros) as four scalar registers
__device__ __forceinline__ wg_u32x4 wg_rsrc(const void* ptr) {
  const unsigned long long a = (unsigned long long)ptr;
  wg_u32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  r.z = WG_OOB;
  r.w = 0x00020000u;
  return r;
}
// A per-lane dword load the compiler does NOT track: its vmcnt bookkeeping is exact only in straight-line code - with the
// loads of a k-step behind the row / edge branches of the cursor it waited vmcnt(0) in front of every k-step, i.e. for the
// loads of the NEXT step issued a moment before (45 % MFMA-busy, latency-bound).  The kernel counts itself: every step is
// exactly 12 of these, three steps are in flight, wg_landed waits for all but the youngest 24.
__device__ __forceinline__ float wg_load(wg_u32x4 rs, unsigned voff, int soff) {
  float r;
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(rs), "s"(soff));
  return r;
}

template <typename F, int... Is>
__device__ __forceinline__ void wg_sfor_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void wg_sfor(F&& f) {
  wg_sfor_impl(f, std::make_integer_sequence<int, N>{});
}
}  // namespace

// NSET = k-steps of loads in flight (12 loads each; the vmcnt counter holds 63)
template <int NSET>
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
  const wg_u32x4 rs_x = wg_rsrc(p.x - (int64_t)(p.W + 1) * p.ldx);
  const wg_u32x4 rs_y = wg_rsrc(p.dy);
  const int ldx4 = p.ldx * 4, ldy4 = p.lddy * 4;
  // k-step u of a tile row multiplies tiles u (lanes lh = 0) and u + ksr (lanes lh = 1): a lane walks CONSECUTIVE tiles, whose
  // windows share two of their four columns - only the two new columns are loaded and their row-transformed values are
  // carried over in registers
  const unsigned lane_x = (unsigned)((lh * 2 * p.ksr * p.ldx + c0 + li) * 4);
  const unsigned lane_y = (unsigned)((lh * 2 * p.ksr * p.lddy + n0 + li) * 4);
  const int tw = (p.W + 1) >> 1;

  f32x16 acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

  // ---- load cursor: (tile row, step) of the NEXT set to fetch, with the row's scalars and per-row lane offsets ----
  // A tile row is ksr + 1 steps: step u = -1 is the PRIMER - it fetches window columns 2, 3 of the tile in front of the lane's
  // first one (= columns 0, 1 of its first tile; out-of-range output-gradient offsets, so its 16 products are zeros) - and
  // steps u = 0 .. ksr - 1 are the k-steps.  EVERY step is exactly 12 loads on every path (wg_load).
#ifdef SSP_PROBES
  bool primed = false;
#endif
  int lrow = row0, lu = -1;
  int l_sox = 0, l_soy = 0;                 // scalar byte offsets of the row's first window pixel / first output pixel
  unsigned vx[4], vy[2];                    // lane offsets of window row i / output row pp: out of range when the row is outside
  auto set_row = [&]() {
    const bool live = lrow < row1;          // (issues run up to three steps past the chunk's end: everything out of range)
    const unsigned b = ssp_div((unsigned)lrow, p.div_th);
    const int ty = lrow - (int)b * p.th;
    // window row i is image row 2 ty - 1 + i; from the shifted base that is row 2 ty + i, pixel column 2 t + j
    l_sox = live ? ((int)b * p.H + 2 * ty) * p.W * ldx4 : 2 * ldx4;
    l_soy = live ? ((int)b * p.H + 2 * ty) * p.W * ldy4 : 2 * ldy4;
#pragma unroll
    for (int i = 0; i < 4; ++i) vx[i] = (live && (unsigned)(2 * ty - 1 + i) < (unsigned)p.H) ? lane_x : WG_OOB;
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) vy[pp] = (live && 2 * ty + pp < p.H) ? lane_y : WG_OOB;
  };
  float rd[NSET][4], rx[NSET][8];     // raw sets in flight: the 2 x 2 output-gradient pixels, window columns 2, 3 (rows 0 .. 3)
  auto issue = [&](auto k_tag) {
    constexpr int K = decltype(k_tag)::value;
#ifdef SSP_PROBES
    if ((p.probe & 8) && primed) return;
#endif
    // columns may fall outside the image at the primer (lanes lh = 0: column -1) and at a row's last k-steps (lanes lh = 1);
    // the primer has no output-gradient pixels at all
    const bool edge = lu < 0 || lu + p.ksr >= tw - 1;
    // (primer: the scalar offsets step back two pixels from the row's first; j >= 2 keeps the x offsets non-negative)
    // (... and the primer's output-gradient loads, all out of range, must not carry a negative scalar offset)
    const int sx = l_sox + lu * 2 * ldx4, sy = l_soy + max(lu, 0) * 2 * ldy4;
    if (edge) {
      const int t2 = 2 * (lu + lh * p.ksr);              // first output column of this lane's tile
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 2; j < 4; ++j) {
          const unsigned v = ((unsigned)(t2 - 1 + j) < (unsigned)p.W) ? vx[i] : WG_OOB;
          rx[K][i * 2 + j - 2] = wg_load(rs_x, v, sx + (i * p.W + j) * ldx4);
        }
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const unsigned v = (lu >= 0 && t2 + qq < p.W) ? vy[pp] : WG_OOB;
          rd[K][pp * 2 + qq] = wg_load(rs_y, v, sy + (pp * p.W + qq) * ldy4);
        }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 2; j < 4; ++j) rx[K][i * 2 + j - 2] = wg_load(rs_x, vx[i], sx + (i * p.W + j) * ldx4);
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) rd[K][pp * 2 + qq] = wg_load(rs_y, vy[pp], sy + (pp * p.W + qq) * ldy4);
    }
    if (++lu == p.ksr) {
      lu = -1;
      ++lrow;
      set_row();
    }
  };
  // ---- one step: transforms in registers, 16 MFMAs (A = dM: rows = output channels, B = V: columns = input channels) ----
  float tc[4][2];                           // row-transformed window columns 2, 3 of the previous step = columns 0, 1 of this one
  auto consume = [&](auto k_tag) {
    constexpr int K = decltype(k_tag)::value;
    // three sets (36 loads) are in flight: the oldest - this one - has landed when all but the youngest 24 have.  The empty
    // statement ties the set's registers to the wait: nothing that reads them is scheduled in front of it
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSET - 1) * 12));
    asm volatile("" : "+v"(rd[K][0]), "+v"(rd[K][1]), "+v"(rd[K][2]), "+v"(rd[K][3]), "+v"(rx[K][0]), "+v"(rx[K][1]), "+v"(rx[K][2]),
                      "+v"(rx[K][3]), "+v"(rx[K][4]), "+v"(rx[K][5]), "+v"(rx[K][6]), "+v"(rx[K][7]));
    // dM = A dY A^T, A = [1 0; 1 1; 1 -1; 0 -1]
    const float d00 = rd[K][0], d01 = rd[K][1], d10 = rd[K][2], d11 = rd[K][3];
    const float s0 = d00 + d10, m0 = d00 - d10, s1 = d01 + d11, m1 = d01 - d11;
    float dm[16];
    dm[0] = d00;  dm[1] = d00 + d01;  dm[2] = d00 - d01;  dm[3] = -d01;
    dm[4] = s0;   dm[5] = s0 + s1;    dm[6] = s0 - s1;    dm[7] = -s1;
    dm[8] = m0;   dm[9] = m0 + m1;    dm[10] = m0 - m1;   dm[11] = -m1;
    dm[12] = -d10; dm[13] = -d10 - d11; dm[14] = d11 - d10; dm[15] = d11;
    // V = B^T d B, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: the row step of the two new window columns (columns 0, 1 are
    // the previous step's 2, 3 - at a primer whatever the last row left: its products are zeros), then the column step
    float v[16];
#pragma unroll
    for (int j = 2; j < 4; ++j) {
      const float r0 = rx[K][j - 2], r1 = rx[K][j], r2 = rx[K][2 + j], r3 = rx[K][4 + j];
      v[j] = r0 - r2; v[4 + j] = r1 + r2; v[8 + j] = r2 - r1; v[12 + j] = r1 - r3;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[4 * i] = tc[i][0]; v[4 * i + 1] = tc[i][1];
      tc[i][0] = v[4 * i + 2]; tc[i][1] = v[4 * i + 3];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a0 = v[4 * i] - v[4 * i + 2], a1 = v[4 * i + 1] + v[4 * i + 2], a2 = v[4 * i + 2] - v[4 * i + 1],
                  a3 = v[4 * i + 1] - v[4 * i + 3];
      v[4 * i] = a0; v[4 * i + 1] = a1; v[4 * i + 2] = a2; v[4 * i + 3] = a3;
    }
#ifdef SSP_PROBES
    if (p.probe & 4) {
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) { dm[xi] = rd[K][xi & 3]; v[xi] = rx[K][xi & 7]; }
    }
    if (p.probe & 16) {
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) asm volatile("" ::"v"(dm[xi]), "v"(v[xi]));
      return;
    }
#endif
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(dm[xi], v[xi], acc[xi], 0, 0, 0);
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) { tc[i][0] = 0.f; tc[i][1] = 0.f; }
  const int nsteps = (row1 - row0) * (p.ksr + 1);
  set_row();
  wg_sfor<NSET>([&](auto K) { issue(K); });
#ifdef SSP_PROBES
  primed = true;
#endif
  int step = 0;
  for (; step + NSET <= nsteps; step += NSET)
    wg_sfor<NSET>([&](auto K) {
      consume(K);
      issue(K);
    });
  wg_sfor<NSET - 1>([&](auto K) {
    if (step < nsteps) {
      consume(K);
      issue(K);
      ++step;
    }
  });
  // The sets fetched past the chunk's end (out of range: zeros) must land before their registers hold anything else: the
  // wait names all of them, so none is handed to the flush's address arithmetic in front of it
#define WG_DRAIN(K)                                                                                                           \
  asm volatile("s_waitcnt vmcnt(0)"                                                                                          \
               : "+v"(rd[K][0]), "+v"(rd[K][1]), "+v"(rd[K][2]), "+v"(rd[K][3]), "+v"(rx[K][0]), "+v"(rx[K][1]), "+v"(rx[K][2]), \
                 "+v"(rx[K][3]), "+v"(rx[K][4]), "+v"(rx[K][5]), "+v"(rx[K][6]), "+v"(rx[K][7]))
  WG_DRAIN(0);
  WG_DRAIN(1);
  WG_DRAIN(2);
  if constexpr (NSET > 3) WG_DRAIN(3);
  if constexpr (NSET > 4) WG_DRAIN(4);
#undef WG_DRAIN

  // ---- flush: dw[cout][a][b][cin] += (G^T dU G)[a][b] of this wave's 32 x 32 block, straight out of the registers ----
  // The back-transform is linear, so every wave applies it to its OWN partial sums and adds nine taps into dw (fp32 atomics;
  // dw is accumulated into, like every filter gradient) - no dU in memory, no zeroing launch and no finishing launch (those
  // two were 8-workgroup, latency-bound launches: 40 - 70 us of a 600 us filter gradient).
  // register r of a 32 x 32 tile: row (r & 3) + 8 (r >> 2) + 4 lh;  G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]
#ifdef SSP_PROBES
  if (p.probe & 32) return;
#endif
  // (buffer atomics: one lane offset, every other address term is a scalar offset - 144 flat addresses would take 288 registers)
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dw + (int64_t)n0 * 9 * p.Cin + c0), 0, 32 * 9 * p.Cin * 4, 0x00020000);
  const int lane_w = (4 * lh * 9 * p.Cin + li) * 4;
  wg_sfor<16>([&](auto R) {
    constexpr int r = decltype(R)::value;
    float u[16], h[4][3];
    // (the empty statement redefines the accumulators: the 16 reads of this register row cannot move in front of it, so the
    // rows are read one at a time - hoisted, the 256 reads took 168 more registers, and with 488 instead of 320 no wave of the
    // other stream's kernels fits next to this one on a SIMD: the step got SLOWER while this kernel got faster)
    asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]),
                      "+a"(acc[8]), "+a"(acc[9]), "+a"(acc[10]), "+a"(acc[11]), "+a"(acc[12]), "+a"(acc[13]), "+a"(acc[14]), "+a"(acc[15]));
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) u[xi] = acc[xi][r];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float sm = 0.5f * (u[4 * i + 1] + u[4 * i + 2]), df = 0.5f * (u[4 * i + 1] - u[4 * i + 2]);
      h[i][0] = u[4 * i] + sm; h[i][1] = df; h[i][2] = sm + u[4 * i + 3];
    }
    const int so = ((r & 3) + 8 * (r >> 2)) * 9 * p.Cin * 4;
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      const float sm = 0.5f * (h[1][bb] + h[2][bb]), df = 0.5f * (h[1][bb] - h[2][bb]);
      __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(h[0][bb] + sm, rs_w, lane_w, so + (0 * 3 + bb) * p.Cin * 4, 0);
      __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(df, rs_w, lane_w, so + (1 * 3 + bb) * p.Cin * 4, 0);
      __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(sm + h[3][bb], rs_w, lane_w, so + (2 * 3 + bb) * p.Cin * 4, 0);
    }
    __builtin_amdgcn_sched_barrier(0);     // one register row at a time (not 256 accumulator reads up front)
  });
#endif
}

int64_t ssp_wino_wgrad_fused_ws_floats(int Cin, int Cout) { return 0; }      // nothing leaves the chip but dw
bool ssp_wino_wgrad_fused_fits(int B, int H, int W, int Cin, int Cout) {
  return Cin % 32 == 0 && Cout % 32 == 0 && Cin >= 32 && Cout >= 32 && (int64_t)B * H * W < (1ll << 28);
}

int ssp_wino_wgrad_fused_launch(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int lddy,
                                int ldx, float* ws, int64_t ws_floats, hipStream_t stream) {
  SSP_CHECK_ARG(ssp_wino_wgrad_fused_fits(B, H, W, Cin, Cout), "wgrad (on-chip Winograd): needs Cin %% 32 == 0 and Cout %% 32 == 0");
  SSP_CHECK_ARG(x != nullptr && dy != nullptr && dw != nullptr && ldx >= Cin && lddy >= Cout,
                "wgrad (on-chip Winograd): needs the layer input (no shared transformed input), ldx >= Cin, lddy >= Cout");
  SSP_CHECK_ARG((((uintptr_t)dw) & 15) == 0, "wgrad (on-chip Winograd): dw must be 16-byte aligned");
  (void)ws; (void)ws_floats;          // (no workspace: the back-transformed partial sums go straight into dw)
  SSP_CHECK_ARG(((int64_t)B * H * W + W + 1) * ldx * 4 < (1ll << 31) && (int64_t)B * H * W * lddy * 4 < (1ll << 31),
                "wgrad (on-chip Winograd): operands beyond the 2 GiB buffer range");
  WinoWgradFusedArgs p;
  p.dy = dy; p.x = x; p.dw = dw;
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
  p.probe = 0;
#ifdef SSP_PROBES
  p.probe = ssp_option(SSP_OPT_WINO_VARIANT) >> 8;
#endif
  // (ring depths 3 / 4 / 5 time the same - 607 / 598 / 596 us on layer 2: with exact waits the loads are not what bounds it)
  hipLaunchKernelGGL(wino2_wgrad_fused_kernel<3>, dim3((unsigned)((p.nunits + 3) / 4)), dim3(256), 0, stream, p);
  SSP_CHECK_LAUNCH("wino2_wgrad_fused");
  return SSP_OK;
}
