// First block of the network: conv 3x3 (3 -> 32 channels, image input) + BatchNorm + leaky + 2x2/2 max-pool, forward and
// backward WITHOUT ever materialising the 32-channel full-resolution conv output.
//
// Replaces, for training, what /root/reference/darknet.py:154-176 does with nn.Conv2d / BatchNorm2d / LeakyReLU /
// MaxPool2d on the 416 x 416 input and their autograd backward (train.py:103).  That map is the largest tensor of the
// network (22 MB per image, 1.42 GB at batch 64) while the convolution that produces it is tiny (K = 27): the generic
// path writes it (write-stream bound, 0.70 ms), reads it for BN + leaky + pool, re-reads it twice in BatchNorm-backward,
// writes a same-sized gradient and reads that once more for the filter gradient - ~8.5 GB of HBM traffic at the two
// ends of the step where nothing else overlaps.  SURVEY.md section 7 ("Layer 0 ... recompute is the HBM-optimal choice").
// Here every pass recomputes the convolution from the 4-channel input (177 MB) on the matrix cores:
//
//   ssp_first_fwd_stats   conv -> per-workgroup (mean, M2) of the raw output per channel       (ssp_bn_fwd_finalize input)
//   ssp_first_fwd_apply   conv -> scale/shift -> leaky -> 2x2 max -> pooled activation          (writes 1/4 of the map)
//   ssp_first_bwd_reduce  conv -> pool arg-max / leaky sign -> (sum dy, sum dy * xhat) partials (ssp_bn finalize input)
//   ssp_first_bwd_wgrad   conv -> dx in registers -> filter gradient                            (no dx in memory)
//
// One MFMA block (v_mfma_f32_32x32x2_f32) = 8 consecutive POOLED pixels of one pooled row x their 4 window positions
// x 32 output channels.  A-operand row i = q * 8 + pp (q = 2*dy + dx window position, pp = pooled pixel) so that in the
// C/D layout (row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31 = channel) a lane holds, for each of its 4 pooled
// pixels pp = 4*(lane>>5) + (r&3), all four window positions q = r>>2: the max-pool, its arg-max and the BatchNorm
// backward of a window are in-lane register work.  K = 36 = 9 taps x 4 (padded) input channels in 18 MFMAs; k-step
// (tap t, u) feeds lane half h with input channel 2h + u, so a lane's A operands are one 8-byte load per tap at a fixed
// lane offset plus a SCALAR (block base + tap) offset - no address VALU next to the MFMAs.  Blocks touching the image
// border (8 % of them) take a predicated path.  The convolution is evaluated by the same instruction sequence in all
// four kernels, so forward and backward see bit-identical raw values (pool winners / leaky signs cannot disagree).
#include "ssp_common.h"

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define FIRST_OOB 0x80000000u
constexpr int FIRST_COUT = 32;
constexpr int FIRST_BLOCK_PIX = 32;          // raw pixels per MFMA block (8 pooled x 4)

struct FirstArgs {
  const float* x;        // [B*H*W][4] input, channel 3 = 0
  const float* wt;       // [32][9][4] packed filters (ssp_repack_fwd)
  int B, H, W, Ho, Wo, nxb, nblocks, bpw;   // bpw = MFMA blocks per wave
  SspFastDiv div_nxb, div_Ho;
  // per-channel vectors (scale, shift: forward BN affine; mean, invstd; c1, c2: mean(dy), mean(dy*xhat))
  const float* scale; const float* shift; const float* mean; const float* invstd; const float* c1; const float* c2;
  float slope;
  float* out; int ldo;          // pooled activation [B*Ho*Wo][ldo]
  const float* g; int ldg;      // gradient wrt the pooled activation
  float* stats;                 // fwd_stats: [nwg][32][2] (mean, M2); bwd_reduce: [nwg][32][2] (sum dy, sum dy*xhat)
  float* dw;                    // [32][9][4] filter gradient (written by first_wgrad_finalize_kernel)
  float* wpart;                 // bwd_wgrad: [nwg][32 * 27] per-workgroup partial filter gradients
};

struct FirstBlock {
  int b, yo, xb;       // image, pooled row, 8-pixel column block
  bool border;
};

__device__ __forceinline__ FirstBlock first_decode(const FirstArgs& p, int blk) {
  FirstBlock k;
  const unsigned t = ssp_div((unsigned)blk, p.div_nxb);
  k.xb = blk - (int)t * p.nxb;
  const unsigned bb = ssp_div(t, p.div_Ho);
  k.yo = (int)(t - bb * (unsigned)p.Ho);
  k.b = (int)bb;
  k.border = (k.yo == 0) | (k.yo == p.Ho - 1) | (k.xb == 0) | (k.xb == p.nxb - 1);
  return k;
}

// The 9 A-operand loads (one 8-byte pair of input channels per tap) of one block.
// lane: i = lane&31 -> pp = i&7, q = i>>3; h = lane>>5 picks channels {2h, 2h+1}.
__device__ __forceinline__ void first_load(const FirstArgs& p, const __amdgpu_buffer_rsrc_t& rs, const FirstBlock& k,
                                           int lane_pix_off /* ((q>>1)*W + 2pp + (q&1)) */, int h, int pp, int q,
                                           u32x2 (&xa)[9]) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int base_pix = (k.b * p.H + 2 * k.yo) * p.W + 16 * k.xb;     // pixel of (pp = 0, q = 0); wave-uniform
  if (!k.border) {
    const unsigned voff = (unsigned)(lane_pix_off * 16 + h * 8);
    const int sbase = __builtin_amdgcn_readfirstlane(base_pix) * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int soff = sbase + ((t / 3 - 1) * p.W + (t % 3 - 1)) * 16;
      xa[t] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    }
  } else {
    const int y = 2 * k.yo + (q >> 1), x = 16 * k.xb + 2 * pp + (q & 1);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
      const bool ok = ((unsigned)yy < (unsigned)p.H) && ((unsigned)xx < (unsigned)p.W);
      const unsigned voff = ok ? (unsigned)((((k.b * p.H + yy) * p.W + xx) * 4 + h * 2) * 4) : FIRST_OOB;
      xa[t] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0);
    }
  }
#endif
}

// raw[r] for this lane's channel: rows pp = 4*(lane>>5) + (r&3), window position q = r>>2
__device__ __forceinline__ f32x16 first_conv(const u32x2 (&xa)[9], const float (&wreg)[18]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    // scalar copies first: __builtin_bit_cast on an ext-vector element mis-selects the element with this compiler
    const unsigned u0 = xa[t][0], u1 = xa[t][1];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, u0), wreg[2 * t], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, u1), wreg[2 * t + 1], acc, 0, 0, 0);
  }
  return acc;
}

__device__ __forceinline__ void first_load_weights(const FirstArgs& p, int cout, int h, float (&wreg)[18]) {
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    wreg[2 * t] = p.wt[cout * 36 + t * 4 + 2 * h];
    wreg[2 * t + 1] = p.wt[cout * 36 + t * 4 + 2 * h + 1];
  }
}

__device__ __forceinline__ void chan_combine_f(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
  const float nt = n + nb;
  if (nt > 0.f) {
    const float d = mb - mean, f = nb / nt;
    mean += d * f;
    m2 += m2b + d * d * n * f;
    n = nt;
  }
}

// MODE 0: forward statistics, 1: forward apply, 2: backward reduce, 3: backward filter gradient,
// 4: write the raw convolution output (checkers only: the training path never stores it)
template <int MODE>
__global__ void __launch_bounds__(256, 3) first_block_kernel(FirstArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int pp = li & 7, q = li >> 3;
  const int cout = li;                                   // C/D column = B-operand column = output channel
  const int lane_pix_off = (q >> 1) * p.W + 2 * pp + (q & 1);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)FIRST_OOB, 0x00020000);
  float wreg[18];
  first_load_weights(p, cout, lh, wreg);

  const int blk0 = (blockIdx.x * 4 + wid) * p.bpw;
  const int nblk = min(p.bpw, p.nblocks - blk0);

  float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f, k1 = 0.f, k2 = 0.f;
  if (MODE >= 1) { sc = p.scale[cout]; sh = p.shift[cout]; }
  if (MODE >= 2) { mu = p.mean[cout]; is = p.invstd[cout]; }
  if (MODE == 3) { k1 = p.c1[cout]; k2 = p.c2[cout]; }

  // running results
  float st_n = 0.f, st_mean = 0.f, st_m2 = 0.f;          // MODE 0
  double s1 = 0.0, s2 = 0.0;                             // MODE 2
  // MODE 3: dW[cout = row][j' = col = lane&31].  A wave's 64 blocks are 1024 MFMA steps into one accumulator - summed in
  // groups of 8 blocks (accw -> acct), as conv_igemm_dma.hip does: the terms cancel ~1e3 : 1 in this gradient (sum dx = 0
  // exactly over an all-positive image), so the length of the fp32 chain is what its error is made of
  f32x16 accw, acct;
#pragma unroll
  for (int r = 0; r < 16; ++r) { accw[r] = 0.f; acct[r] = 0.f; }
  // MODE 3 B operand (input patch value for column j' = tap*3 + ci of the filter gradient): fixed lane offset
  const int jt = li / 3, jc = li - jt * 3;               // lane's (tap, input channel); li >= 27: no such column
  // bytes from pixel (y-1, x-1) of the step's first pixel: never negative (a "negative" vector offset is out of range
  // for the buffer unit); the descriptor of this operand starts W+1 pixels before the image.  4*lh pooled = 8*lh raw px.
  const int w_lane_off = ((jt / 3) * p.W + (jt % 3) + 8 * lh) * 16 + jc * 4;
  const __amdgpu_buffer_rsrc_t rs_xs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - (int64_t)(p.W + 1) * 4), 0,
                                                                         (int)FIRST_OOB, 0x00020000);

  __amdgpu_buffer_rsrc_t rs_o = rs_x, rs_g = rs_x;
  if (MODE == 1 || MODE == 4) rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)FIRST_OOB, 0x00020000);
  if (MODE >= 2) rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, (int)FIRST_OOB, 0x00020000);

  // Two register sets (input pairs of the 9 taps + the 4 pooled gradients of the backward modes), the loop unrolled by two:
  // block n + 1's set is fetched - unconditionally: past the wave's last block the last one again, so that every path issues
  // the same loads and the compiler's vmcnt bookkeeping stays exact - in front of block n's MFMAs.
  auto fetch = [&](const FirstBlock& k, u32x2 (&xa)[9], float (&gv)[4]) {
    first_load(p, rs_x, k, lane_pix_off, lh, pp, q, xa);
    if constexpr (MODE == 2 || MODE == 3) {
      const int pooled_u = __builtin_amdgcn_readfirstlane((k.b * p.Ho + k.yo) * p.Wo + 8 * k.xb);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        gv[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, (unsigned)((4 * lh * p.ldg + cout) * 4),
                                                                                (pooled_u + c) * p.ldg * 4, 0));
    }
  };
  auto compute = [&](const FirstBlock& k, const FirstBlock& kn, const u32x2 (&xa)[9], const float (&gv)[4], u32x2 (&xn)[9],
                     float (&gn)[4]) {
    // MODE 3: the 16 input-patch values of the filter-gradient MFMAs (B operand: lane = column j' = (tap, channel), step r =
    // pixel pair) go first - they hit the lines this block's own taps just fetched and land under the 18 convolution MFMAs -
    // then the next block's set (a fresh HBM stream, a whole block of work ahead of its use)
    float bv[16];
    if constexpr (MODE == 3) {
      const int base_pix = __builtin_amdgcn_readfirstlane((k.b * p.H + 2 * k.yo) * p.W + 16 * k.xb);
      if (!k.border) {
        const unsigned voff = (li < 27) ? (unsigned)w_lane_off : FIRST_OOB;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = r & 3, w = r >> 2;
          bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
              rs_xs, voff, (base_pix + (w >> 1) * p.W + (w & 1) + 2 * c) * 16, 0));
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = r & 3, w = r >> 2;
          const int y = 2 * k.yo + (w >> 1) + jt / 3 - 1, x = 16 * k.xb + 2 * (c + 4 * lh) + (w & 1) + jt % 3 - 1;
          const bool ok = (li < 27) && ((unsigned)y < (unsigned)p.H) && ((unsigned)x < (unsigned)p.W);
          const unsigned voff = ok ? (unsigned)((((k.b * p.H + y) * p.W + x) * 4 + jc) * 4) : FIRST_OOB;
          bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, voff, 0, 0));
        }
      }
    }
    fetch(kn, xn, gn);
    const f32x16 acc = first_conv(xa, wreg);
    // pooled pixels of this block start at pooled_u (wave-uniform); the lane owns pixels 4*lh + {0,1,2,3}: the lane part
    // of an address sits in the (fixed) vector offset, the block / pixel part in the scalar offset
    const int pooled_u = __builtin_amdgcn_readfirstlane((k.b * p.Ho + k.yo) * p.Wo + 8 * k.xb);

    if constexpr (MODE == 4) {
      // raw[pixel][cout]: lane (cout, lh) holds pixels (pp = 4*lh + c, q = w) at reg c + 4*w
      const int base_pix = __builtin_amdgcn_readfirstlane((k.b * p.H + 2 * k.yo) * p.W + 16 * k.xb);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = r & 3, w = r >> 2;
        const float v = acc[r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_o, (unsigned)((8 * lh * p.ldo + cout) * 4),
                                              (base_pix + (w >> 1) * p.W + (w & 1) + 2 * c) * p.ldo * 4, 0);
      }
    } else if constexpr (MODE == 0) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[r];
      const float mb = sum * (1.f / 16.f);
      float m2b = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = acc[r] - mb; m2b += d * d; }
      chan_combine_f(st_n, st_mean, st_m2, 16.f, mb, m2b);
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float best = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float y = acc[c + 4 * w] * sc + sh;
          const float a = y > 0.f ? y : y * p.slope;
          best = a > best ? a : best;
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, best), rs_o,
                                              (unsigned)((4 * lh * p.ldo + cout) * 4), (pooled_u + c) * p.ldo * 4, 0);
      }
    } else {
      // backward: gradient of the 4 pooled pixels, pool winner (first maximum in (dy, dx) scan order, as ATen) and
      // the activation-side gradient dy at the winner
      float dyv[4], xh_sel[4];
      int sel[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float gvc = gv[c];
        float y0 = acc[c] * sc + sh;
        float best = y0 > 0.f ? y0 : y0 * p.slope, ybest = y0, xbest = acc[c];
        int s = 0;
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const float y = acc[c + 4 * w] * sc + sh;
          const float a = y > 0.f ? y : y * p.slope;
          if (a > best) { best = a; ybest = y; xbest = acc[c + 4 * w]; s = w; }
        }
        dyv[c] = ybest > 0.f ? gvc : gvc * p.slope;
        xh_sel[c] = (xbest - mu) * is;
        sel[c] = s;
      }
      if constexpr (MODE == 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { s1 += (double)dyv[c]; s2 += (double)dyv[c] * (double)xh_sel[c]; }
      } else {
        // dx of the 16 raw pixels, then dW[cout][j'] += sum_pixels dx[pixel][cout] * patch[pixel][j']:
        // MFMA step r takes pixel rows (r&3) + 8*(r>>2) (+4 for the upper lane half) - exactly where dx[r] already sits
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = r & 3, w = r >> 2;
          const float xh = (acc[r] - mu) * is;
          const float dyq = (sel[c] == w) ? dyv[c] : 0.f;
          const float dx = sc * (dyq - k1 - xh * k2);
          accw = __builtin_amdgcn_mfma_f32_32x32x2f32(dx, bv[r], accw, 0, 0, 0);
        }
      }
    }
  };

  u32x2 xa[9], xb[9];
  float ga[4], gb[4];
  const int last = blk0 + nblk - 1;
  FirstBlock ka = first_decode(p, nblk > 0 ? blk0 : 0), kb = ka;
  if (nblk > 0) fetch(ka, xa, ga);
  for (int g0 = 0; g0 < nblk; g0 += 8) {                 // (groups of 8 blocks: the fold of MODE 3 sits OUTSIDE the block loop -
    const int ge = min(nblk, g0 + 8);                    // inside, the compiler turned it into selects run on every block)
    for (int ib = g0; ib < ge; ib += 2) {
      kb = first_decode(p, min(blk0 + ib + 1, last));
      compute(ka, kb, xa, ga, xb, gb);
      ka = first_decode(p, min(blk0 + ib + 2, last));
      if (ib + 1 < nblk) compute(kb, ka, xb, gb, xa, ga);
    }
    if constexpr (MODE == 3) {
      acct += accw;
#pragma unroll
      for (int r = 0; r < 16; ++r) accw[r] = 0.f;
    }
  }
  if constexpr (MODE == 3) accw = acct;

  // ---- workgroup results ----
  __shared__ __attribute__((aligned(16))) float red[4][32][32];
  if constexpr (MODE == 0) {
    float on = __shfl_xor(st_n, 32), om = __shfl_xor(st_mean, 32), o2 = __shfl_xor(st_m2, 32);
    chan_combine_f(st_n, st_mean, st_m2, on, om, o2);
    if (lh == 0) { red[wid][cout][0] = st_n; red[wid][cout][1] = st_mean; red[wid][cout][2] = st_m2; }
    __syncthreads();
    if (tid < 32) {
      float n = red[0][tid][0], m = red[0][tid][1], m2 = red[0][tid][2];
#pragma unroll
      for (int w = 1; w < 4; ++w) chan_combine_f(n, m, m2, red[w][tid][0], red[w][tid][1], red[w][tid][2]);
      float* st = p.stats + ((int64_t)blockIdx.x * FIRST_COUT + tid) * 2;
      st[0] = m;
      st[1] = m2;
    }
  } else if constexpr (MODE == 2) {
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    double* redd = reinterpret_cast<double*>(&red[0][0][0]);      // [4][32][2]
    if (lh == 0) { redd[(wid * 32 + cout) * 2] = s1; redd[(wid * 32 + cout) * 2 + 1] = s2; }
    __syncthreads();
    if (tid < 32) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) { a += redd[(w * 32 + tid) * 2]; b += redd[(w * 32 + tid) * 2 + 1]; }
      float* st = p.stats + ((int64_t)blockIdx.x * FIRST_COUT + tid) * 2;
      st[0] = (float)a;
      st[1] = (float)b;
    }
  } else if constexpr (MODE == 3) {
    // accw: col = lane&31 = j', rows (r&3) + 8*(r>>2) + 4*lh = cout
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wid][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = accw[r];
    __syncthreads();
    // this workgroup's partial gradient, a row of its own: the workgroups are summed in float64 by
    // first_wgrad_finalize_kernel (1352 fp32 atomics per element at 416 x 416, batch 64, were one more long fp32 chain -
    // and made the result depend on the order the workgroups retired in)
    for (int e = tid; e < 32 * 27; e += 256) {
      const int co = e / 27, j = e - co * 27;
      p.wpart[(int64_t)blockIdx.x * (32 * 27) + e] = red[0][co][j] + red[1][co][j] + red[2][co][j] + red[3][co][j];
    }
  }
#endif
}

// dw[co][tap][ci] = sum over the workgroups' partials, in float64 (thread = (element of a group of 32, one of 8 slices of the
// workgroup range), LDS reduce over the slices); the padding channel of the [32][9][4] layout is written as zero
__global__ void __launch_bounds__(256) first_wgrad_finalize_kernel(const float* __restrict__ wpart, int nwg, float* __restrict__ dw) {
  const int tid = threadIdx.x, el = tid & 31, sl = tid >> 5;
  const int e = blockIdx.x * 32 + el;                    // 27 blocks x 32 = 864 elements
  double a = 0.0;
  for (int w = sl; w < nwg; w += 8) a += (double)wpart[(int64_t)w * (32 * 27) + e];
  __shared__ double red[8][32];
  red[sl][el] = a;
  __syncthreads();
  if (sl == 0) {
    double t = red[0][el];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][el];
    const int co = e / 27, j = e - co * 27;
    dw[co * 36 + (j / 3) * 4 + j % 3] = (float)t;
    if (e < 32 * 9) dw[e * 4 + 3] = 0.f;
  }
}

static int first_check(const float* x, const float* wt, int B, int H, int W, const char* who) {
  SSP_CHECK_ARG(x != nullptr && wt != nullptr && B > 0 && H > 0 && W > 0, "%s: null operand or empty shape", who);
  SSP_CHECK_ARG(H % 2 == 0 && W % 16 == 0, "%s: needs even H and W a multiple of 16 (8 pooled pixels per MFMA block)", who);
  SSP_CHECK_ARG((int64_t)B * H * W * 16 < (1ll << 31), "%s: input too large for 32-bit buffer offsets", who);
  SSP_CHECK_ARG((((uintptr_t)x) & 15) == 0, "%s: x must be 16-byte aligned", who);
  return SSP_OK;
}

static FirstArgs first_args(const float* x, const float* wt, int B, int H, int W, int bpw) {
  FirstArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.wt = wt; a.B = B; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2; a.nxb = a.Wo / 8;
  a.nblocks = B * a.Ho * a.nxb;
  a.bpw = bpw;
  a.div_nxb = ssp_fastdiv((unsigned)a.nxb);
  a.div_Ho = ssp_fastdiv((unsigned)a.Ho);
  a.slope = 1.f;
  return a;
}

// workgroups of the statistics / reduce kernels: 64 MFMA blocks = 2048 raw pixels each (the BatchNorm finalize kernels
// take one partial row per workgroup, tile_m = 2048)
int ssp_first_tile_pixels_impl(void) { return 64 * FIRST_BLOCK_PIX; }
int ssp_first_groups_impl(int B, int H, int W) { return ssp_cdiv((int64_t)B * (H / 2) * (W / 16), 64); }

int ssp_first_fwd_stats_launch(const float* x, const float* wt, float* stats, int B, int H, int W, hipStream_t stream) {
  if (int rc = first_check(x, wt, B, H, W, "first_fwd_stats")) return rc;
  FirstArgs a = first_args(x, wt, B, H, W, 16);
  a.stats = stats;
  SspProfScope prof(SSP_PROF_FIRST_FWD, stream, 2.0 * (double)B * H * W * FIRST_COUT * 27.0);   // algorithmic conv FLOPs (3 real input channels), booked once per direction
  hipLaunchKernelGGL(first_block_kernel<0>, dim3(ssp_cdiv(a.nblocks, 64)), dim3(256), 0, stream, a);
  SSP_CHECK_LAUNCH("first_fwd_stats");
  return SSP_OK;
}

int ssp_first_fwd_apply_launch(const float* x, const float* wt, const float* scale, const float* shift, float slope,
                               float* out, int ldo, int B, int H, int W, hipStream_t stream) {
  if (int rc = first_check(x, wt, B, H, W, "first_fwd_apply")) return rc;
  SSP_CHECK_ARG(out != nullptr && ldo >= FIRST_COUT && (int64_t)B * (H / 2) * (W / 2) * ldo * 4 < (1ll << 31),
                "first_fwd_apply: bad output (ldo >= 32, < 2 GiB)");
  FirstArgs a = first_args(x, wt, B, H, W, 16);
  a.scale = scale; a.shift = shift; a.slope = slope; a.out = out; a.ldo = ldo;
  SspProfScope prof(SSP_PROF_FIRST_FWD, stream, 0.0);
  hipLaunchKernelGGL(first_block_kernel<1>, dim3(ssp_cdiv(a.nblocks, 64)), dim3(256), 0, stream, a);
  SSP_CHECK_LAUNCH("first_fwd_apply");
  return SSP_OK;
}

int ssp_first_conv_raw_launch(const float* x, const float* wt, float* raw, int ldraw, int B, int H, int W,
                              hipStream_t stream) {
  if (int rc = first_check(x, wt, B, H, W, "first_conv_raw")) return rc;
  SSP_CHECK_ARG(raw != nullptr && ldraw >= FIRST_COUT && (int64_t)B * H * W * ldraw * 4 < (1ll << 31),
                "first_conv_raw: bad output (ldraw >= 32, < 2 GiB)");
  FirstArgs a = first_args(x, wt, B, H, W, 16);
  a.out = raw; a.ldo = ldraw;
  hipLaunchKernelGGL(first_block_kernel<4>, dim3(ssp_cdiv(a.nblocks, 64)), dim3(256), 0, stream, a);
  SSP_CHECK_LAUNCH("first_conv_raw");
  return SSP_OK;
}

int ssp_first_bwd_reduce_launch(const float* x, const float* wt, const float* g, int ldg, const float* scale,
                                const float* shift, const float* mean, const float* invstd, float slope, float* partial,
                                int B, int H, int W, hipStream_t stream) {
  if (int rc = first_check(x, wt, B, H, W, "first_bwd_reduce")) return rc;
  SSP_CHECK_ARG(g != nullptr && ldg >= FIRST_COUT && (int64_t)B * (H / 2) * (W / 2) * ldg * 4 < (1ll << 31),
                "first_bwd_reduce: bad gradient (ldg >= 32, < 2 GiB)");
  FirstArgs a = first_args(x, wt, B, H, W, 16);
  a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.slope = slope; a.g = g; a.ldg = ldg;
  a.stats = partial;
  SspProfScope prof(SSP_PROF_FIRST_BWD, stream, 0.0);
  hipLaunchKernelGGL(first_block_kernel<2>, dim3(ssp_cdiv(a.nblocks, 64)), dim3(256), 0, stream, a);
  SSP_CHECK_LAUNCH("first_bwd_reduce");
  return SSP_OK;
}

int64_t ssp_first_wgrad_workspace_floats_impl(int B, int H, int W) {
  return (int64_t)ssp_cdiv((int64_t)B * (H / 2) * (W / 16), 256) * (32 * 27);
}

int ssp_first_bwd_wgrad_launch(const float* x, const float* wt, const float* g, int ldg, const float* scale,
                               const float* shift, const float* mean, const float* invstd, const float* c1,
                               const float* c2, float slope, float* dw, float* workspace, int64_t workspace_floats, int B,
                               int H, int W, hipStream_t stream) {
  if (int rc = first_check(x, wt, B, H, W, "first_bwd_wgrad")) return rc;
  SSP_CHECK_ARG(g != nullptr && ldg >= FIRST_COUT && dw != nullptr, "first_bwd_wgrad: null gradient operand");
  SSP_CHECK_ARG(workspace != nullptr && workspace_floats >= ssp_first_wgrad_workspace_floats_impl(B, H, W),
                "first_bwd_wgrad: needs a workspace of %lld floats (ssp_first_wgrad_workspace_floats)",
                (long long)ssp_first_wgrad_workspace_floats_impl(B, H, W));
  // fewer, longer workgroups (64 blocks per wave): each ends with one 864-float row of the partial buffer
  FirstArgs a = first_args(x, wt, B, H, W, 64);
  a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.c1 = c1; a.c2 = c2; a.slope = slope;
  a.g = g; a.ldg = ldg; a.dw = dw; a.wpart = workspace;
  SspProfScope prof(SSP_PROF_FIRST_BWD, stream, 2.0 * (double)B * H * W * FIRST_COUT * 27.0);
  const int nwg = ssp_cdiv(a.nblocks, 256);
  hipLaunchKernelGGL(first_block_kernel<3>, dim3(nwg), dim3(256), 0, stream, a);
  SSP_CHECK_LAUNCH("first_bwd_wgrad");
  hipLaunchKernelGGL(first_wgrad_finalize_kernel, dim3(27), dim3(256), 0, stream, workspace, nwg, dw);
  SSP_CHECK_LAUNCH("first_wgrad_finalize");
  return SSP_OK;
}
