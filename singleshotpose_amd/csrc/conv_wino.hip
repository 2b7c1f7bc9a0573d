// Winograd F(n x n, 3x3) transforms, n = 2 and n = 4, for the deep 3x3 layers.
//
// The fp32 MFMA is the binding resource of the training step (conv_igemm_dma.hip runs it 88 - 93 % busy), and on gfx950 it
// has no faster fp32 form - so the remaining lever on those layers is arithmetic: a 3x3 stride-1 convolution evaluated on
// n x n output tiles needs (n+2)^2 multiplies per tile and channel pair instead of 9 n^2 (Lavin & Gray, "Fast Algorithms
// for Convolutional Neural Networks"): 16 instead of 36 for n = 2, 36 instead of 144 for n = 4.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per n x n output tile, summed over input channels
//
// becomes, with the channel sum pulled inside, P = (n+2)^2 independent GEMMs (one per position xi of the transform domain):
//   M_xi [T][Cout] = V_xi [T][Cin] * U_xi [Cout][Cin]^T,     T = B * ceil(H/n) * ceil(W/n) tiles
// which run as ONE batched launch of the LDS-direct implicit-GEMM kernel (R = 1, gridDim.y = P).  Around it (all HBM-bound):
//   * wino_filter_kernel   U = G g G^T   from the [rows][tap][K] filter layout both passes already use (forward: the
//     channels-last parameter itself; data gradient: the flipped / transposed operand ssp_repack_dgrad_packed builds);
//   * wino_input_kernel    V = B^T d B   one thread per (tile, 2 or 4 channels), zero padding = the tile's out-of-image taps;
//   * wino_output_kernel   the finishing pass: one thread per (tile, 4 channels) reads the tile's P values ONCE, forms the
//     n x n outputs A^T M A and does everything the split-K finishing pass does (bias, eval-mode affine + leaky, accumulate,
//     per-tile-group BatchNorm statistics, fused BatchNorm-backward reductions);
//   * wino_outgrad_kernel  dM = A dY A^T and wino_wgrad_finish_kernel  dw += G^T dU G  for the filter gradient, which is the
//     same P-fold batched launch of the LDS-direct filter-gradient kernel over the tiles.
//
// n = 2 uses the points (0, 1, -1): constants 1, -1, 1/2 only; result within ~1e-6 of the direct kernel.
// n = 4 uses the points (0, 1, -1, 1/2, -2) - NOT the textbook (0, +-1, +-2): on this network's operand statistics the
// mixed set halves the error (fp32 simulation against float64, K = 512: rms 5.0e-7 of the output's range against 2.3e-7 for
// the direct fp32 sum and 7.0e-7 for the textbook points; max 3.3e-6 / 1.6e-6 / 1.0e-5; the filter gradient comes out at the
// direct kernel's own error).  oracle/wino_ref.py holds the same matrices and is pinned to F.conv2d on the CPU.
// Algorithmic FLOPs stay 2 * M * Cout * 9 * Cin (what the profiler books); the MFMA executes P / (9 n^2) of them, times the
// tile padding of the map (13 x 13: 49/42.25 for n = 2, 256/169 for n = 4).
#include <utility>

#include "conv_wino.h"

// ---- compile-time loops: every matrix coefficient below is a constant the compiler folds (zeros vanish, +-1 become adds) ----
template <typename F, int... Is>
__device__ __forceinline__ void ssp_sfor_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void ssp_sfor(F&& f) {
  ssp_sfor_impl(f, std::make_integer_sequence<int, N>{});
}

// acc += c * x with a compile-time c
template <typename V>
__device__ __forceinline__ void wino_mad(V& acc, const float c, const V& x) {
  if (c == 1.f) acc += x;
  else if (c == -1.f) acc -= x;
  else if (c != 0.f) acc += x * c;
}

template <int N> struct WinoMat;
template <> struct WinoMat<2> {
  static constexpr int A = 4;
  static constexpr float BT[4][4] = {{1.f, 0.f, -1.f, 0.f}, {0.f, 1.f, 1.f, 0.f}, {0.f, -1.f, 1.f, 0.f}, {0.f, 1.f, 0.f, -1.f}};
  static constexpr float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  static constexpr float AT[2][4] = {{1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, -1.f}};
};
// Cook-Toom with the points (0, 1, -1, 1/2, -2) and infinity (oracle/wino_ref.py derives the same matrices from the points)
template <> struct WinoMat<4> {
  static constexpr int A = 6;
  static constexpr float BT[6][6] = {{1.f, -1.5f, -2.f, 1.5f, 1.f, 0.f},
                                     {0.f, -1.f, 0.5f, 2.5f, 1.f, 0.f},
                                     {0.f, 1.f, -2.5f, 0.5f, 1.f, 0.f},
                                     {0.f, -2.f, -1.f, 2.f, 1.f, 0.f},
                                     {0.f, 0.5f, -1.f, -0.5f, 1.f, 0.f},
                                     {0.f, 1.f, -1.5f, -2.f, 1.5f, 1.f}};
  static constexpr float G[6][3] = {{1.f, 0.f, 0.f},
                                    {1.f / 3.f, 1.f / 3.f, 1.f / 3.f},
                                    {-1.f / 3.f, 1.f / 3.f, -1.f / 3.f},
                                    {-16.f / 15.f, -8.f / 15.f, -4.f / 15.f},
                                    {1.f / 15.f, -2.f / 15.f, 4.f / 15.f},
                                    {0.f, 0.f, 1.f}};
  static constexpr float AT[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f},
                                     {0.f, 1.f, -1.f, 0.5f, -2.f, 0.f},
                                     {0.f, 1.f, 1.f, 0.25f, 4.f, 0.f},
                                     {0.f, 1.f, -1.f, 0.125f, -8.f, 1.f}};
};

// Tiling of the maps.  Plain (G = 1): tile t = (b * th + ty) * tw + tx covers output pixels (n ty .. n ty + n - 1,
// n tx .. n tx + n - 1) of image b, th = ceil(H / n).  A map whose edge is one pixel more than a multiple of n wastes most of
// its last tile row and column: 13 x 13 at n = 4 takes 4 x 4 tiles = 256 computed pixels for 169 (1.51 x), and the six
// 13 x 13 layers are half of the network's FLOPs.  MOSAIC (G = 2): four consecutive images are laid out 2 x 2 with ONE zero
// row / column between them - exactly the zero padding a 3 x 3 filter sees at an image border - and the (2H + 1) x (2W + 1)
// mosaic is tiled as one map: 27 x 27 -> 7 x 7 tiles for FOUR images, 49 instead of 64 (12.25 per image: 1.16 x).  Only the
// gather of the input / output-gradient transforms and the scatter of the finishing pass know about it (a window or tile that
// straddles two images reads each pixel from its own image, the gap row reads as zero and its outputs are dropped); the
// GEMMs just see fewer tile rows.  Chosen whenever it needs fewer tiles (a pure function of the launch shape: every query
// and launcher agrees): H = W = 13 (416 x 416 input) and 21 (valid.py's 672 x 672) at n = 4, 9 / 17 / 25 of the multi-scale
// schedule; never for maps that tile exactly or batches where the phantom images of the last mosaic outweigh the gain.
struct WinoGeom {
  int H, W, th, tw;
  int64_t T;
  SspFastDiv div_tw, div_th;
  int G, B;      // G = 1: plain, 2: 2 x 2 mosaic of images; B = images (the last mosaic may hold phantom ones)
};
static WinoGeom wino_geom(int B, int H, int W, int tile) {
  WinoGeom g;
  g.H = H; g.W = W; g.B = B; g.G = 1;
  g.th = (H + tile - 1) / tile; g.tw = (W + tile - 1) / tile;
  g.T = (int64_t)B * g.th * g.tw;
  const int mth = (2 * H + 1 + tile - 1) / tile, mtw = (2 * W + 1 + tile - 1) / tile;
  const int64_t mT = (int64_t)((B + 3) / 4) * mth * mtw;
  if (mT < g.T) { g.G = 2; g.th = mth; g.tw = mtw; g.T = mT; }
  g.div_tw = ssp_fastdiv((unsigned)g.tw); g.div_th = ssp_fastdiv((unsigned)g.th);
  return g;
}
// mosaic position (Y, X) of mosaic (or image, G = 1) mb -> image b and pixel (y, x); false: padding, gap or phantom image
__device__ __forceinline__ bool wino_map(const WinoGeom& g, int mb, int Y, int X, int& b, int& y, int& x) {
  if (g.G == 1) {
    b = mb; y = Y; x = X;
    return ((unsigned)Y < (unsigned)g.H) && ((unsigned)X < (unsigned)g.W);
  }
  const int iy = Y > g.H ? 1 : 0, ix = X > g.W ? 1 : 0;
  y = Y - iy * (g.H + 1);
  x = X - ix * (g.W + 1);
  b = mb * 4 + iy * 2 + ix;
  return (Y >= 0) && (X >= 0) && (y < g.H) && (x < g.W) && (b < g.B);
}

// ---- V = B^T d B: thread = (tile, CV channels) -------------------------------------------------------------------------
// NT: non-temporal stores for the planes (written once here, read once by the GEMM launch)
template <int N, int CV, bool NT = false>
__global__ void __launch_bounds__(256) wino_input_kernel(const float* __restrict__ in, int ldin, float* __restrict__ V, int C,
                                                         WinoGeom g, SspFastDiv div_cg) {
  constexpr int A = N + 2;
  using M_ = WinoMat<N>;
  typedef float vec __attribute__((ext_vector_type(CV)));
  // XCD-aware order: consecutive workgroups go to the 8 XCDs round robin, each with its own L2 - remapped so that an XCD
  // walks a contiguous run of tiles and the halo rows / columns neighbouring tiles share (20 of a tile's 36 taps at n = 4)
  // come out of its L2 instead of being fetched again over the fabric (PMC: 171 MB fetched per launch for 107 MB of input)
  const int64_t gid = (int64_t)ssp_xcd_remap((int)blockIdx.x, (int)gridDim.x) * 256 + threadIdx.x;
  const int cgn = C / CV;
  if (gid >= g.T * cgn) return;
  const unsigned t = ssp_div((unsigned)gid, div_cg);        // T * C / CV < 2^31 is checked by the launcher
  const int c = (int)((unsigned)gid - t * (unsigned)cgn) * CV;
  const unsigned q = ssp_div(t, g.div_tw);                  // mb * th + ty
  const int tx = (int)(t - q * (unsigned)g.tw);
  const unsigned mb = ssp_div(q, g.div_th);                 // image (plain) or mosaic of four
  const int ty = (int)(q - mb * (unsigned)g.th);
  const int y0 = N * ty - 1, x0 = N * tx - 1;
  const float* base = in + c;
  vec d[A][A];
#pragma unroll
  for (int i = 0; i < A; ++i) {
#pragma unroll
    for (int j = 0; j < A; ++j) {
      int b, y, x;
      const bool ok = wino_map(g, (int)mb, y0 + i, x0 + j, b, y, x);
      vec z;
#pragma unroll
      for (int k = 0; k < CV; ++k) z[k] = 0.f;
      d[i][j] = ok ? *reinterpret_cast<const vec*>(base + (((int64_t)b * g.H + y) * g.W + x) * ldin) : z;
    }
  }
  // r = B^T d (over the rows), then V = r B (the same combination over the columns)
  vec r[A][A];
  ssp_sfor<A>([&](auto I) {
    constexpr int i = decltype(I)::value;
    ssp_sfor<A>([&](auto J) {
      constexpr int j = decltype(J)::value;
      vec a;
#pragma unroll
      for (int k = 0; k < CV; ++k) a[k] = 0.f;
      ssp_sfor<A>([&](auto K) {
        constexpr int k = decltype(K)::value;
        { constexpr float cc_ = M_::BT[i][k]; wino_mad(a, cc_, d[k][j]); }
      });
      r[i][j] = a;
    });
  });
  float* dst = V + (int64_t)t * C + c;
  const int64_t plane = g.T * C;
  ssp_sfor<A>([&](auto I) {
    constexpr int i = decltype(I)::value;
    ssp_sfor<A>([&](auto J) {
      constexpr int j = decltype(J)::value;
      vec a;
#pragma unroll
      for (int k = 0; k < CV; ++k) a[k] = 0.f;
      ssp_sfor<A>([&](auto K) {
        constexpr int k = decltype(K)::value;
        { constexpr float cc_ = M_::BT[j][k]; wino_mad(a, cc_, r[i][k]); }
      });
      if constexpr (NT) __builtin_nontemporal_store(a, reinterpret_cast<vec*>(dst + (int64_t)(i * A + j) * plane));
      else *reinterpret_cast<vec*>(dst + (int64_t)(i * A + j) * plane) = a;
    });
  });
}

// ---- U[xi][row][k] = (G g G^T)[xi] of the 3x3 filter g[tap] = w[row][tap][k]; thread = (row, 4 k's) ---------------------
template <int N>
__global__ void __launch_bounds__(256) wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int rows, int K) {
  constexpr int A = N + 2;
  using M_ = WinoMat<N>;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int k4n = K >> 2;
  if (gid >= (int64_t)rows * k4n) return;
  const int row = (int)(gid / k4n);
  const int k = (int)(gid - (int64_t)row * k4n) * 4;
  const float* src = w + ((int64_t)row * 9) * K + k;
  f32x4 g[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) g[i][j] = *reinterpret_cast<const f32x4*>(src + (int64_t)(i * 3 + j) * K);
  f32x4 h[A][3];      // G g
  ssp_sfor<A>([&](auto I) {
    constexpr int i = decltype(I)::value;
    ssp_sfor<3>([&](auto J) {
      constexpr int j = decltype(J)::value;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      ssp_sfor<3>([&](auto Kk) {
        constexpr int kk = decltype(Kk)::value;
        { constexpr float cc_ = M_::G[i][kk]; wino_mad(a, cc_, g[kk][j]); }
      });
      h[i][j] = a;
    });
  });
  float* dst = U + (int64_t)row * K + k;
  const int64_t plane = (int64_t)rows * K;
  ssp_sfor<A>([&](auto I) {
    constexpr int i = decltype(I)::value;
    ssp_sfor<A>([&](auto J) {
      constexpr int j = decltype(J)::value;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      ssp_sfor<3>([&](auto Kk) {
        constexpr int kk = decltype(Kk)::value;
        { constexpr float cc_ = M_::G[j][kk]; wino_mad(a, cc_, h[i][kk]); }
      });
      *reinterpret_cast<f32x4*>(dst + (int64_t)(i * A + j) * plane) = a;
    });
  });
}

// ---- finishing pass of a forward / data-gradient launch ----------------------------------------------------------------
// Y = A^T M A per tile, then what the split-K finishing pass does.  Grid = (groups of SSP_WINO_TG tiles, 64-channel slabs);
// thread = (tile of the group, 4 channels): the tile's P plane values are read exactly once (the pixel-wise gather this
// kernel replaces read 9 of 16 planes per output pixel: 2.25 x the bytes, all of them fabric traffic on the deep layers).
// BatchNorm statistics: per tile GROUP and channel (mean, M2) of the valid pixels, plus the group's pixel count (the maps'
// odd edges make it vary) behind the pairs: stats [groups][Cout][2] | counts [groups] - bn_fwd_finalize's counted format.
// CS = channel quads per workgroup (16: a 64-channel slab, 16 tiles side by side; 64: a 256-channel slab, 4 tiles side by
// side, the group's 16 tiles in 4 passes): a tile row of the slab is one contiguous piece of a plane - 256 bytes or 1 KiB.
template <int N, int CS>
__global__ void __launch_bounds__(256) wino_output_kernel(WinoOutArgs p, WinoGeom g) {
  constexpr int A = N + 2;
  constexpr int TP = 256 / CS;                 // tiles per pass
  constexpr int NP = SSP_WINO_TG / TP;         // passes over the group
  using M_ = WinoMat<N>;
  const int tid = threadIdx.x, gl = tid % CS, tl = tid / CS;
  const int c = blockIdx.y * (CS * 4) + gl * 4;
  const bool cok = c < p.Cout;                 // Cout % 4 == 0 on this path (checked by the launcher)
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 b4 = z4, e4 = {1.f, 1.f, 1.f, 1.f};
  if (cok && p.bias != nullptr) b4 = *reinterpret_cast<const f32x4*>(p.bias + c);
  if (cok && p.escale != nullptr) e4 = *reinterpret_cast<const f32x4*>(p.escale + c);
  const bool bnb = p.bn_partial != nullptr;
  const bool want_stats = p.stats != nullptr;
  f32x4 bsc = z4, bsh = z4, bmu = z4, bis = z4, s1 = z4, s2 = z4;
  if (bnb && cok) {
    bsc = *reinterpret_cast<const f32x4*>(p.bn_scale + c); bsh = *reinterpret_cast<const f32x4*>(p.bn_shift + c);
    bmu = *reinterpret_cast<const f32x4*>(p.bn_mean + c); bis = *reinterpret_cast<const f32x4*>(p.bn_invstd + c);
  }
  // running (count, mean, M2) of this thread's tiles, Chan-combined tile by tile
  float rn = 0.f;
  f32x4 rmean = z4, rm2 = z4;
#pragma unroll 1
  for (int pass = 0; pass < NP; ++pass) {
    const int64_t t64 = (int64_t)blockIdx.x * SSP_WINO_TG + pass * TP + tl;
    const bool live = cok && (t64 < g.T);
    if (!live) continue;
    f32x4 Y[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = 0; j < N; ++j) Y[i][j] = z4;
    const unsigned t = (unsigned)t64;
    const unsigned q = ssp_div(t, g.div_tw);
    const int tx = (int)(t - q * (unsigned)g.tw);
    const int mb = (int)ssp_div(q, g.div_th);
    const int ty = (int)(q - (unsigned)mb * (unsigned)g.th);
    const float* src = p.Mw + t64 * p.Cout + c;
    const int64_t plane = g.T * p.Cout;
    // row i of the transform domain at a time: rr = M[i][:] A (n values), then Y[:][q] += A^T[:][i] rr[q]
    ssp_sfor<A>([&](auto I) {
      constexpr int i = decltype(I)::value;
      f32x4 m[A];
#pragma unroll
      for (int j = 0; j < A; ++j) m[j] = *reinterpret_cast<const f32x4*>(src + (int64_t)(i * A + j) * plane);
      ssp_sfor<N>([&](auto Q) {
        constexpr int qq = decltype(Q)::value;
        f32x4 rr = z4;
        ssp_sfor<A>([&](auto J) {
          constexpr int j = decltype(J)::value;
          { constexpr float cc_ = M_::AT[qq][j]; wino_mad(rr, cc_, m[j]); }
        });
        ssp_sfor<N>([&](auto Pp) {
          constexpr int pp = decltype(Pp)::value;
          { constexpr float cc_ = M_::AT[pp][i]; wino_mad(Y[pp][qq], cc_, rr); }
        });
      });
    });
    // the tile's valid pixels: epilogue + statistics of the raw (bias-free) values (count and mean, then M2)
    float cnt = 0.f;
    f32x4 sum = z4;
    unsigned valid = 0u;      // bit i * N + j: output (i, j) of the tile is a pixel of a real image
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        int b, y, x;
        if (wino_map(g, mb, N * ty + i, N * tx + j, b, y, x)) {
          valid |= 1u << (i * N + j);
          const f32x4 v = Y[i][j];
          cnt += 1.f;
          sum += v;
          const int64_t m = ((int64_t)b * g.H + y) * g.W + x;
          float* dst = p.out + m * p.ldout + c;
          f32x4 o = v * e4 + b4;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.f ? o[k] : o[k] * p.act_slope;
          if (p.accumulate) o += *reinterpret_cast<const f32x4*>(dst);
          *reinterpret_cast<f32x4*>(dst) = o;
          if (bnb) {
            const f32x4 xr = *reinterpret_cast<const f32x4*>(p.bn_raw + m * p.bn_ld + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float yy = xr[k] * bsc[k] + bsh[k];
              const float dyv = yy > 0.f ? o[k] : o[k] * p.bn_slope;
              s1[k] += dyv;
              s2[k] += dyv * ((xr[k] - bmu[k]) * bis[k]);
            }
          }
        }
      }
    }
    if (want_stats && cnt > 0.f) {
      const f32x4 mean = sum * (1.f / cnt);
      f32x4 m2 = z4;
#pragma unroll
      for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j)
          if ((valid >> (i * N + j)) & 1u) {
            const f32x4 dd = Y[i][j] - mean;
            m2 += dd * dd;
          }
      const float nt = rn + cnt, f = cnt / nt;
      const f32x4 dd = mean - rmean;
      rmean += dd * f;
      rm2 += m2 + dd * dd * (rn * f);
      rn = nt;
    }
  }
  __shared__ float red[TP][CS][9];   // [tile lane][channel quad][cnt, mean x4, m2 x4]  (or [-, s1 x4, s2 x4])
  if (bnb) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[tl][gl][1 + k] = s1[k]; red[tl][gl][5 + k] = s2[k]; }
    __syncthreads();
    if (tid < CS * 4) {
      const int q = tid >> 2, k = tid & 3;
      const int ch = blockIdx.y * (CS * 4) + tid;
      if (ch < p.Cout) {
        float a = 0.f, bb = 0.f;
#pragma unroll
        for (int w = 0; w < TP; ++w) { a += red[w][q][1 + k]; bb += red[w][q][5 + k]; }
        if ((int)gridDim.x > p.bn_nslot) {
          float* dst = p.bn_partial + ((int64_t)(blockIdx.x % p.bn_nslot) * p.Cout + ch) * 2;
          atomicAdd(dst, a);
          atomicAdd(dst + 1, bb);
        } else {
          float* dst = p.bn_partial + ((int64_t)blockIdx.x * p.Cout + ch) * 2;
          dst[0] = a;
          dst[1] = bb;
        }
      }
    }
    return;
  }
  if (!want_stats) return;
  red[tl][gl][0] = rn;
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[tl][gl][1 + k] = rmean[k]; red[tl][gl][5 + k] = rm2[k]; }
  __syncthreads();
  if (tid < CS * 4) {                     // one thread per channel of the slab
    const int q = tid >> 2, k = tid & 3;
    const int ch = blockIdx.y * (CS * 4) + tid;
    if (ch < p.Cout) {
      float n_ = red[0][q][0], mu = red[0][q][1 + k], ss = red[0][q][5 + k];
#pragma unroll
      for (int w = 1; w < TP; ++w) {
        const float nb = red[w][q][0], mb = red[w][q][1 + k], m2b = red[w][q][5 + k];
        const float nt = n_ + nb;
        if (nt > 0.f) {
          const float dd = mb - mu, f = nb / nt;
          mu += dd * f;
          ss += m2b + dd * dd * n_ * f;
          n_ = nt;
        }
      }
      float* st = p.stats + ((int64_t)blockIdx.x * p.Cout + ch) * 2;
      st[0] = mu;
      st[1] = ss;
      if (ch == 0) p.stats[(int64_t)gridDim.x * p.Cout * 2 + blockIdx.x] = n_;      // the group's pixel count
    }
  }
}

// ---- filter gradient in the transform domain --------------------------------------------------------------------------
// dL/dU_xi [Cout][Cin] = sum_t dM_xi[t][Cout] (x) V_xi[t][Cin]   with   dM = A dY A^T  (the adjoint of the inverse transform:
// an n x n tile of the output gradient spread over the transform domain) and V the transformed input of the forward pass;
// then dL/dg = G^T (dL/dU) G.  The P sums over the tiles are P pixel-contraction GEMMs - one batched launch of the
// LDS-direct filter-gradient kernel (conv_wgrad_dma.hip, R = 1, gridDim.y = P).
template <int N, int CV, bool NT = false>
__global__ void __launch_bounds__(256) wino_outgrad_kernel(const float* __restrict__ dy, int lddy, float* __restrict__ dM, int C,
                                                           WinoGeom g, SspFastDiv div_cg) {
  constexpr int A = N + 2;
  using M_ = WinoMat<N>;
  typedef float vec __attribute__((ext_vector_type(CV)));
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cgn = C / CV;
  if (gid >= g.T * cgn) return;
  const unsigned t = ssp_div((unsigned)gid, div_cg);
  const int c = (int)((unsigned)gid - t * (unsigned)cgn) * CV;
  const unsigned q = ssp_div(t, g.div_tw);
  const int tx = (int)(t - q * (unsigned)g.tw);
  const unsigned mb = ssp_div(q, g.div_th);
  const int ty = (int)(q - mb * (unsigned)g.th);
  const float* base = dy + c;
  vec d[N][N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      int b, y, x;
      const bool ok = wino_map(g, (int)mb, N * ty + i, N * tx + j, b, y, x);
      vec z;
#pragma unroll
      for (int k = 0; k < CV; ++k) z[k] = 0.f;
      d[i][j] = ok ? *reinterpret_cast<const vec*>(base + (((int64_t)b * g.H + y) * g.W + x) * lddy) : z;
    }
  // r = A d (A = (A^T)^T: r[i][q] = sum_p AT[p][i] d[p][q]), then dM = r A^T
  vec r[A][N];
  ssp_sfor<A>([&](auto I) {
    constexpr int i = decltype(I)::value;
    ssp_sfor<N>([&](auto Q) {
      constexpr int qq = decltype(Q)::value;
      vec a;
#pragma unroll
      for (int k = 0; k < CV; ++k) a[k] = 0.f;
      ssp_sfor<N>([&](auto Pp) {
        constexpr int pp = decltype(Pp)::value;
        { constexpr float cc_ = M_::AT[pp][i]; wino_mad(a, cc_, d[pp][qq]); }
      });
      r[i][qq] = a;
    });
  });
  float* dst = dM + (int64_t)t * C + c;
  const int64_t plane = g.T * C;
  ssp_sfor<A>([&](auto I) {
    constexpr int i = decltype(I)::value;
    ssp_sfor<A>([&](auto J) {
      constexpr int j = decltype(J)::value;
      vec a;
#pragma unroll
      for (int k = 0; k < CV; ++k) a[k] = 0.f;
      ssp_sfor<N>([&](auto Q) {
        constexpr int qq = decltype(Q)::value;
        { constexpr float cc_ = M_::AT[qq][j]; wino_mad(a, cc_, r[i][qq]); }
      });
      if constexpr (NT) __builtin_nontemporal_store(a, reinterpret_cast<vec*>(dst + (int64_t)(i * A + j) * plane));
      else *reinterpret_cast<vec*>(dst + (int64_t)(i * A + j) * plane) = a;
    });
  });
}

// dw[row][tap][k] += (G^T dU G)[tap]; thread = (row, 4 k's); dw is this launch's own (no other writer): plain read-add-write.
// Row i of dU at a time: h[b] = sum_j G[j][b] dU[i][j], acc[a][b] += G[i][a] h[b].
template <int N>
__global__ void __launch_bounds__(256) wino_wgrad_finish_kernel(const float* __restrict__ dU, float* dw, int rows, int K) {
  constexpr int A = N + 2;
  using M_ = WinoMat<N>;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int k4n = K >> 2;
  if (gid >= (int64_t)rows * k4n) return;
  const int row = (int)(gid / k4n);
  const int k = (int)(gid - (int64_t)row * k4n) * 4;
  const float* src = dU + (int64_t)row * K + k;
  const int64_t plane = (int64_t)rows * K;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) acc[a][bb] = z4;
  ssp_sfor<A>([&](auto I) {
    constexpr int i = decltype(I)::value;
    f32x4 u[A];
#pragma unroll
    for (int j = 0; j < A; ++j) u[j] = *reinterpret_cast<const f32x4*>(src + (int64_t)(i * A + j) * plane);
    ssp_sfor<3>([&](auto Bb) {
      constexpr int bb = decltype(Bb)::value;
      f32x4 h = z4;
      ssp_sfor<A>([&](auto J) {
        constexpr int j = decltype(J)::value;
        { constexpr float cc_ = M_::G[j][bb]; wino_mad(h, cc_, u[j]); }
      });
      ssp_sfor<3>([&](auto Aa) {
        constexpr int a = decltype(Aa)::value;
        { constexpr float cc_ = M_::G[i][a]; wino_mad(acc[a][bb], cc_, h); }
      });
    });
  });
  float* dst = dw + ((int64_t)row * 9) * K + k;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      f32x4* o = reinterpret_cast<f32x4*>(dst + (int64_t)(a * 3 + bb) * K);
      *o = *o + acc[a][bb];
    }
}

// ---- launchers -----------------------------------------------------------------------------------------------------------
static inline bool wino_tile_ok(int tile) { return tile == 2 || tile == 4; }
int64_t ssp_wino_tiles(int B, int H, int W, int tile) { return wino_geom(B, H, W, tile).T; }
int ssp_wino_planes(int tile) { return (tile + 2) * (tile + 2); }
int64_t ssp_wino_stat_groups(int B, int H, int W, int tile) { return (ssp_wino_tiles(B, H, W, tile) + SSP_WINO_TG - 1) / SSP_WINO_TG; }

int ssp_wino_outgrad_launch(const float* dy, int lddy, float* dM, int B, int H, int W, int C, int tile, hipStream_t stream) {
  SSP_CHECK_ARG(wino_tile_ok(tile), "wino_outgrad: tile must be 2 or 4");
  const WinoGeom g = wino_geom(B, H, W, tile);
  const int wv = ssp_option(SSP_OPT_WINO_VARIANT);      // experiments: bit 0 = 4 channels per thread at F(4x4), bit 1 = non-temporal stores
  const int cv = (tile == 2 || (wv & 1)) ? 4 : 2;
  SSP_CHECK_ARG(C % 4 == 0 && lddy % 4 == 0 && (((uintptr_t)dy) & 15) == 0 && (((uintptr_t)dM) & 15) == 0,
                "wino_outgrad: channels must be a multiple of 4 and the operands 16-byte aligned");
  SSP_CHECK_ARG(g.T * (C / cv) < (1ll << 31), "wino_outgrad: too many (tile, channel) pairs");
  const int P = ssp_wino_planes(tile);
  SspProfScope prof(SSP_PROF_WINO_WGRAD, stream, 4.0 * C * ((double)B * H * W + (double)P * g.T));
  const int64_t n = g.T * (C / cv);
  const SspFastDiv dc = ssp_fastdiv((unsigned)(C / cv));
  const dim3 grid((unsigned)((n + 255) / 256));
  if (tile == 2 && (wv & 2)) hipLaunchKernelGGL((wino_outgrad_kernel<2, 4, true>), grid, dim3(256), 0, stream, dy, lddy, dM, C, g, dc);
  else if (tile == 2) hipLaunchKernelGGL((wino_outgrad_kernel<2, 4>), grid, dim3(256), 0, stream, dy, lddy, dM, C, g, dc);
  else if (cv == 4 && (wv & 2)) hipLaunchKernelGGL((wino_outgrad_kernel<4, 4, true>), grid, dim3(256), 0, stream, dy, lddy, dM, C, g, dc);
  else if (cv == 4) hipLaunchKernelGGL((wino_outgrad_kernel<4, 4>), grid, dim3(256), 0, stream, dy, lddy, dM, C, g, dc);
  else if (wv & 2) hipLaunchKernelGGL((wino_outgrad_kernel<4, 2, true>), grid, dim3(256), 0, stream, dy, lddy, dM, C, g, dc);
  else hipLaunchKernelGGL((wino_outgrad_kernel<4, 2>), grid, dim3(256), 0, stream, dy, lddy, dM, C, g, dc);
  SSP_CHECK_LAUNCH("wino_outgrad");
  return SSP_OK;
}

int ssp_wino_wgrad_finish_launch(const float* dU, float* dw, int rows, int K, int tile, hipStream_t stream) {
  SSP_CHECK_ARG(wino_tile_ok(tile), "wino_wgrad_finish: tile must be 2 or 4");
  const int64_t n = (int64_t)rows * (K / 4);
  SspProfScope prof(SSP_PROF_WINO_WGRAD, stream, 4.0 * (double)rows * K * (ssp_wino_planes(tile) + 18.0));
  if (tile == 2) hipLaunchKernelGGL(wino_wgrad_finish_kernel<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dU, dw, rows, K);
  else hipLaunchKernelGGL(wino_wgrad_finish_kernel<4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dU, dw, rows, K);
  SSP_CHECK_LAUNCH("wino_wgrad_finish");
  return SSP_OK;
}

int ssp_wino_input_launch(const float* in, int ldin, float* V, int B, int H, int W, int C, int tile, int prof_kind, hipStream_t stream) {
  SSP_CHECK_ARG(wino_tile_ok(tile), "wino_input: tile must be 2 or 4");
  const WinoGeom g = wino_geom(B, H, W, tile);
  const int wv = ssp_option(SSP_OPT_WINO_VARIANT);
  const int cv = (tile == 2 || (wv & 1)) ? 4 : 2;
  SSP_CHECK_ARG(C % 4 == 0 && ldin % 4 == 0 && (((uintptr_t)in) & 15) == 0 && (((uintptr_t)V) & 15) == 0,
                "wino_input: channels must be a multiple of 4 and the operands 16-byte aligned");
  SSP_CHECK_ARG(g.T * (C / cv) < (1ll << 31), "wino_input: too many (tile, channel) pairs");
  const int P = ssp_wino_planes(tile);
  SspProfScope prof(prof_kind, stream, 4.0 * C * ((double)B * H * W + (double)P * g.T));
  const int64_t n = g.T * (C / cv);
  const SspFastDiv dc = ssp_fastdiv((unsigned)(C / cv));
  const dim3 grid((unsigned)((n + 255) / 256));
  if (tile == 2 && (wv & 2)) hipLaunchKernelGGL((wino_input_kernel<2, 4, true>), grid, dim3(256), 0, stream, in, ldin, V, C, g, dc);
  else if (tile == 2) hipLaunchKernelGGL((wino_input_kernel<2, 4>), grid, dim3(256), 0, stream, in, ldin, V, C, g, dc);
  else if (cv == 4 && (wv & 2)) hipLaunchKernelGGL((wino_input_kernel<4, 4, true>), grid, dim3(256), 0, stream, in, ldin, V, C, g, dc);
  else if (cv == 4) hipLaunchKernelGGL((wino_input_kernel<4, 4>), grid, dim3(256), 0, stream, in, ldin, V, C, g, dc);
  else if (wv & 2) hipLaunchKernelGGL((wino_input_kernel<4, 2, true>), grid, dim3(256), 0, stream, in, ldin, V, C, g, dc);
  else hipLaunchKernelGGL((wino_input_kernel<4, 2>), grid, dim3(256), 0, stream, in, ldin, V, C, g, dc);
  SSP_CHECK_LAUNCH("wino_input");
  return SSP_OK;
}

int ssp_wino_filter_launch(const float* w, float* U, int rows, int K, int tile, hipStream_t stream) {
  SSP_CHECK_ARG(wino_tile_ok(tile), "wino_filter: tile must be 2 or 4");
  SSP_CHECK_ARG(w != nullptr && U != nullptr && rows > 0 && K > 0 && K % 4 == 0 && (((uintptr_t)w) & 15) == 0 &&
                    (((uintptr_t)U) & 15) == 0,
                "wino_filter: [rows][9][K] filters with K % 4 == 0, 16-byte aligned operands");
  SspProfScope prof(SSP_PROF_LAYOUT, stream, 4.0 * (9.0 + ssp_wino_planes(tile)) * (double)rows * K);
  const int64_t n = (int64_t)rows * (K / 4);
  if (tile == 2) hipLaunchKernelGGL(wino_filter_kernel<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, U, rows, K);
  else hipLaunchKernelGGL(wino_filter_kernel<4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, U, rows, K);
  SSP_CHECK_LAUNCH("wino_filter");
  return SSP_OK;
}

// the finishing pass; `a` = the conv launch's own arguments (conv_igemm.hip fills a WinoOutArgs from its ConvArgs)
int ssp_wino_output_launch(const WinoOutArgs& a, int B, int H, int W, int tile, int prof_kind, hipStream_t stream) {
  SSP_CHECK_ARG(wino_tile_ok(tile), "wino_output: tile must be 2 or 4");
  const WinoGeom g = wino_geom(B, H, W, tile);
  const int P = ssp_wino_planes(tile);
  SspProfScope prof(prof_kind, stream, 4.0 * a.Cout * ((double)B * H * W * (a.accumulate ? 2.0 : 1.0) * (a.bn_partial ? 2.0 : 1.0) + (double)P * g.T));
  // 64-channel slabs.  256-channel slabs (1 KiB pieces of a plane per tile row; wino_variant bit 2) measured no better on
  // any layer and 5-10 % worse on the 52 x 52 / 26 x 26 ones (profiles/r04_wino_xform_variants.txt): experiment switch only
  const bool wide = a.Cout >= 256 && (ssp_option(SSP_OPT_WINO_VARIANT) & 4);
  const dim3 grid((unsigned)((g.T + SSP_WINO_TG - 1) / SSP_WINO_TG), (unsigned)ssp_cdiv(a.Cout, wide ? 256 : 64));
  if (tile == 2 && wide) hipLaunchKernelGGL((wino_output_kernel<2, 64>), grid, dim3(256), 0, stream, a, g);
  else if (tile == 2) hipLaunchKernelGGL((wino_output_kernel<2, 16>), grid, dim3(256), 0, stream, a, g);
  else if (wide) hipLaunchKernelGGL((wino_output_kernel<4, 64>), grid, dim3(256), 0, stream, a, g);
  else hipLaunchKernelGGL((wino_output_kernel<4, 16>), grid, dim3(256), 0, stream, a, g);
  SSP_CHECK_LAUNCH("wino_output");
  return SSP_OK;
}
