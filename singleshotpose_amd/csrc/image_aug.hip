// Training-time image augmentation on the GPU (second half): what
// /root/reference/image.py does per sample with Pillow on the host - change_background (:111-128: background resized to
// the image, composited through the mask), data_augmentation (:46-76: jitter crop with zero fill, resize to the network
// shape) and distort_image (:14-31: HSV jitter through Image.point tables) - as four launches per BATCH, byte-exact.
//
// All of it is Pillow's integer / float arithmetic, restated (oracle/image_ref.py pins the same restatement against Pillow
// over all 2^24 colours and against the reference's own outputs):
//   * resize = two passes of a per-output-index FIR with 22-bit fixed-point coefficients and an 8-bit intermediate
//     (ImagingResample: horizontal pass over the rows the vertical pass needs, then the vertical pass; bicubic a = -0.5).
//     The coefficient rows are computed on the host in double (singleshotpose_amd/image.py: a few KB per sample) - the
//     kernels apply them.  A crop is a window offset on the source of the horizontal pass; pixels outside the source are 0
//     (Image.crop).  A pass whose size does not change has the identity as its coefficients, so it is not special-cased.
//   * RGB -> HSV -> three 256-entry tables -> RGB per pixel, in the mixed float / double precision of Pillow's Convert.c.
// HBM-bound byte work: one thread per output pixel (3 channels), taps walk a contiguous row segment (horizontal) or a
// column with a fixed row stride (vertical: neighbouring threads read neighbouring bytes).  This file is compiled with
// -ffp-contract=off: the colour conversions must round every product and sum on their own, as the C code they restate.
#include "ssp_common.h"

#include "../../include/ssp_hip.h"

#define AUG_PRECISION_BITS 22

__device__ __forceinline__ unsigned char aug_clip8(int ss) {
  int v = ss >> AUG_PRECISION_BITS;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// Pillow rgb2hsv (Convert.c): float32 ratios, the hue sum evaluated in double and rounded to float32 twice
__device__ __forceinline__ void aug_rgb2hsv(int r, int g, int b, int& uh, int& us, int& uv) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  uv = maxc;
  if (maxc == minc) {
    uh = 0;
    us = 0;
    return;
  }
  const float cr = (float)(maxc - minc);
  const float s = __fdiv_rn(cr, (float)maxc);
  const double rc = (double)__fdiv_rn((float)(maxc - r), cr);
  const double gc = (double)__fdiv_rn((float)(maxc - g), cr);
  const double bc = (double)__fdiv_rn((float)(maxc - b), cr);
  double h64;
  if (r == maxc) h64 = bc - gc;
  else if (g == maxc) h64 = __dsub_rn(__dadd_rn(2.0, rc), bc);
  else h64 = __dsub_rn(__dadd_rn(4.0, gc), rc);
  float h = (float)h64;
  double t = __dadd_rn(__ddiv_rn((double)h, 6.0), 1.0);
  t = t - floor(t);                                  // fmod(t, 1.0) for t > 0
  h = (float)t;
  int ih = (int)__dmul_rn((double)h, 255.0);
  int is = (int)__fmul_rn(s, 255.0f);
  uh = ih < 0 ? 0 : (ih > 255 ? 255 : ih);
  us = is < 0 ? 0 : (is > 255 ? 255 : is);
}

__device__ __forceinline__ int aug_round_clip(double x) {      // C round() of a non-negative value, then CLIP8
  const int v = (int)floor(__dadd_rn(x, 0.5));
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// Pillow hsv2rgb (Convert.c, "following colorsys.py")
__device__ __forceinline__ void aug_hsv2rgb(int h, int s, int v, int& r, int& g, int& b) {
  if (s == 0) {
    r = g = b = v;
    return;
  }
  const double hh = __ddiv_rn(__dmul_rn((double)h, 6.0), 255.0);
  const double fi = floor(hh);
  const double f = __dsub_rn(hh, fi);
  const double fs = __ddiv_rn((double)s, 255.0);
  const double dv = (double)v;
  const int p = aug_round_clip(__dmul_rn(dv, __dsub_rn(1.0, fs)));
  const int q = aug_round_clip(__dmul_rn(dv, __dsub_rn(1.0, __dmul_rn(fs, f))));
  const int t = aug_round_clip(__dmul_rn(dv, __dsub_rn(1.0, __dmul_rn(fs, __dsub_rn(1.0, f)))));
  switch (((int)fi) % 6) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

// pass 0 = horizontal (taps along x, source window + zero fill), pass 1 = vertical (taps along y, dense source).
// epilogue 0 = store, 1 = composite (dst = mask >= 128 ? img : value, per channel), 2 = distort (HSV tables).
template <int PASS, int EPI>
__global__ void __launch_bounds__(256) aug_resample_kernel(const SspResampleDesc* __restrict__ descs) {
  const SspResampleDesc d = descs[blockIdx.y];
  const int npix = d.dst_w * d.dst_h;
  const int* __restrict__ bounds = reinterpret_cast<const int*>(d.bounds);
  const int* __restrict__ kk = reinterpret_cast<const int*>(d.kk);
  const unsigned char* __restrict__ src = reinterpret_cast<const unsigned char*>(d.src);
  unsigned char* __restrict__ dst = reinterpret_cast<unsigned char*>(d.dst);
  for (int pix = blockIdx.x * 256 + threadIdx.x; pix < npix; pix += gridDim.x * 256) {
    const int yo = pix / d.dst_w, xo = pix - yo * d.dst_w;
    const int o = PASS == 0 ? xo : yo;
    const int lo = bounds[2 * o], n = bounds[2 * o + 1];
    const int* __restrict__ k = kk + (int64_t)o * d.ksize;
    int s0 = 1 << (AUG_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    if (PASS == 0) {
      // logical source pixel (yo + row0, lo + x) sits at physical (y + y0, x + x0); outside the source = 0 (Image.crop)
      const int py = yo + d.row0 + d.y0;
      if (py >= 0 && py < d.src_h) {
        const unsigned char* row = src + (int64_t)py * d.src_pitch;
        for (int x = 0; x < n; ++x) {
          const int px = lo + x + d.x0;
          if (px >= 0 && px < d.src_w) {
            const unsigned char* p = row + 3 * px;
            const int c = k[x];
            s0 += p[0] * c; s1 += p[1] * c; s2 += p[2] * c;
          }
        }
      }
    } else {
      // dense source of d.src_h rows whose row 0 is logical row d.row0
      const unsigned char* p = src + (int64_t)(lo - d.row0) * d.src_pitch + 3 * xo;
      for (int y = 0; y < n; ++y, p += d.src_pitch) {
        const int c = k[y];
        s0 += p[0] * c; s1 += p[1] * c; s2 += p[2] * c;
      }
    }
    int r = aug_clip8(s0), g = aug_clip8(s1), b = aug_clip8(s2);
    if (EPI == 1) {
      const unsigned char* im = reinterpret_cast<const unsigned char*>(d.img) + (int64_t)yo * d.img_pitch + 3 * xo;
      const unsigned char* mk = reinterpret_cast<const unsigned char*>(d.mask) + (int64_t)yo * d.img_pitch + 3 * xo;
      r = mk[0] >= 128 ? im[0] : r;
      g = mk[1] >= 128 ? im[1] : g;
      b = mk[2] >= 128 ? im[2] : b;
    } else if (EPI == 2) {
      const unsigned char* lut = reinterpret_cast<const unsigned char*>(d.lut);
      int h, s, v;
      aug_rgb2hsv(r, g, b, h, s, v);
      aug_hsv2rgb(lut[h], lut[256 + s], lut[512 + v], r, g, b);
    }
    unsigned char* q = dst + (int64_t)yo * d.dst_pitch + 3 * xo;
    q[0] = (unsigned char)r; q[1] = (unsigned char)g; q[2] = (unsigned char)b;
  }
}

// mode 0: rgb -> hsv -> tables -> rgb (distort_image); 1: rgb -> hsv only; 2: hsv -> rgb only (checkers)
__global__ void __launch_bounds__(256) aug_distort_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                          int64_t npix, const unsigned char* __restrict__ lut, int mode) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    int a = in[3 * i], b = in[3 * i + 1], c = in[3 * i + 2];
    int x, y, z;
    if (mode == 2) {
      aug_hsv2rgb(a, b, c, x, y, z);
    } else {
      aug_rgb2hsv(a, b, c, x, y, z);
      if (mode == 0) aug_hsv2rgb(lut[x], lut[256 + y], lut[512 + z], x, y, z);
    }
    out[3 * i] = (unsigned char)x; out[3 * i + 1] = (unsigned char)y; out[3 * i + 2] = (unsigned char)z;
  }
}

int ssp_resample_u8_launch(const SspResampleDesc* descs, int count, int pass, int epilogue, int max_dst_pixels,
                           hipStream_t stream) {
  SSP_CHECK_ARG(descs != nullptr && count > 0 && max_dst_pixels > 0, "resample_u8: empty batch");
  SSP_CHECK_ARG((pass == 0 || pass == 1) && epilogue >= 0 && epilogue <= 2 && !(pass == 0 && epilogue != 0),
                "resample_u8: pass 0 (horizontal) stores plainly; pass 1 (vertical) takes epilogue 0 / 1 / 2");
  SSP_CHECK_ARG(count <= 65535, "resample_u8: at most 65535 samples per launch");
  SspProfScope prof(SSP_PROF_LAYOUT, stream, 0.0);
  int gx = (max_dst_pixels + 255) / 256;
  if (gx > 4096) gx = 4096;
  const dim3 grid(gx, count), block(256);
  if (pass == 0) hipLaunchKernelGGL((aug_resample_kernel<0, 0>), grid, block, 0, stream, descs);
  else if (epilogue == 0) hipLaunchKernelGGL((aug_resample_kernel<1, 0>), grid, block, 0, stream, descs);
  else if (epilogue == 1) hipLaunchKernelGGL((aug_resample_kernel<1, 1>), grid, block, 0, stream, descs);
  else hipLaunchKernelGGL((aug_resample_kernel<1, 2>), grid, block, 0, stream, descs);
  SSP_CHECK_LAUNCH("resample_u8");
  return SSP_OK;
}

int ssp_distort_u8_launch(const unsigned char* rgb, unsigned char* out, int64_t npix, const unsigned char* lut, int mode,
                          hipStream_t stream) {
  SSP_CHECK_ARG(rgb != nullptr && out != nullptr && npix > 0, "distort_u8: empty image");
  SSP_CHECK_ARG(mode >= 0 && mode <= 2 && (mode != 0 || lut != nullptr), "distort_u8: mode 0 needs the 768-byte table");
  SspProfScope prof(SSP_PROF_LAYOUT, stream, 6.0 * (double)npix);
  int64_t gx = (npix + 255) / 256;
  if (gx > 16384) gx = 16384;
  hipLaunchKernelGGL(aug_distort_kernel, dim3((unsigned)gx), dim3(256), 0, stream, rgb, out, npix, lut, mode);
  SSP_CHECK_LAUNCH("distort_u8");
  return SSP_OK;
}
