// RegionLoss head: decode + target building + masked-SSE loss + dL/d(output) in ONE launch, no host sync.
//
// Replaces RegionLoss.forward + build_targets (/root/reference/region_loss.py:9-78,95-175), their
// multi-object twins (/root/reference/multi_obj_pose_estimation/region_loss_multi.py:9-92,110-189),
// corner_confidences / corner_confidence (/root/reference/utils.py:138-187) and the autograd backward of the
// 19 MSELoss terms.  The reference copies all predictions to the host, loops in Python and uploads 23 tensors;
// here one workgroup owns one image: its cells live in registers/LDS, wave reductions give the scalars.
//
// Also the inference decode: per-image arg-max of the objectness (get_region_boxes, /root/reference/utils.py:216-296).
//
// Quirks reproduced on purpose: only keypoint 0 goes through the sigmoid; targets come from
// detached predictions; corner_confidences normalises by exp(2)-1 but corner_confidence by exp(2)-1+1e-5; both
// hard-code 640x480 px and th=80; a GT overwrites earlier GTs of the same cell; the GT list ends at the first
// row whose x0 == 0; in the multi-object loss tconf is taken from anchor "-1" of the PREVIOUS image
// (region_loss_multi.py:51,63: best_n is still -1 when pred_box is indexed).
#include "ssp_common.h"

#define SSP_MAX_GT 50
#define SSP_MAX_CELLS 4096

struct RegionArgs {
  int nB, nA, nC, nH, nW;
  float noobject_scale, object_scale, coord_scale, class_scale, thresh;
  int conf_on;       // epoch > pretrain_num_epochs
  int multi;         // multi-object semantics (anchor pick, CE term, best_n=-1 quirk)
  int tgt_stride;    // target elements per image (50*(2K+3))
  const float* anchors;  // device [nA*anchor_step] or nullptr
  int anchor_step;
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// mean over K keypoints of [d < 80] * (exp(2*(1 - d/80)) - 1) / conf0, d = pixel distance with x*640, y*480
template <int K>
__device__ __forceinline__ float corner_conf(const float* gt, const float* pr, float conf0) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float dx = (gt[2 * k] - pr[2 * k]) * 640.f;
    float dy = (gt[2 * k + 1] - pr[2 * k + 1]) * 480.f;
    float d = sqrtf(dx * dx + dy * dy);
    float c = (expf(2.f * (1.f - d / 80.f)) - 1.f) / conf0;
    s += (d < 80.f) ? c : 0.f;
  }
  return s / (float)K;
}

// decode the K (x,y) predictions of one cell into normalised image coordinates (region_loss.py:109-125)
template <int K>
__device__ __forceinline__ void decode_cell(const float* out, int64_t base, int64_t hw, int i, int j, int nW, int nH,
                                            float* pr) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float vx = out[base + (2 * k) * hw], vy = out[base + (2 * k + 1) * hw];
    if (k == 0) { vx = sigmoidf_(vx); vy = sigmoidf_(vy); }
    pr[2 * k] = (vx + (float)i) / (float)nW;
    pr[2 * k + 1] = (vy + (float)j) / (float)nH;
  }
}

template <int K, typename T>
__global__ void __launch_bounds__(256) region_loss_kernel(const float* __restrict__ out, const T* __restrict__ target,
                                                          float* __restrict__ grad, float* __restrict__ partials,
                                                          RegionArgs a) {
  constexpr int NL = 2 * K + 3;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nCh = 2 * K + 1 + a.nC;
  const int64_t hw = (int64_t)a.nH * a.nW;
  const int ncell = a.nA * a.nH * a.nW;
  const float conf0_all = expf(2.f) - 1.f;           // corner_confidences (utils.py:160)
  const float conf0_one = expf(2.f) - 1.f + 1e-5f;   // corner_confidence  (utils.py:184)

  __shared__ float gt_c[SSP_MAX_GT][2 * K];   // float32 GT corners (torch.FloatTensor(g))
  __shared__ float gt_t[SSP_MAX_GT][2 * K];   // targets tx/ty relative to the centroid cell
  __shared__ float gt_conf[SSP_MAX_GT];
  __shared__ int gt_cell[SSP_MAX_GT];         // a*hw + j*nW + i of the responsible cell
  __shared__ int gt_cls[SSP_MAX_GT];
  __shared__ int cell_gt[SSP_MAX_CELLS];
  __shared__ int s_ngt;
  __shared__ float red[8][4];

  const T* tg = target + (int64_t)b * a.tgt_stride;
  if (tid == 0) {
    int n = 0;
    while (n < SSP_MAX_GT && tg[n * NL + 1] != (T)0) ++n;
    s_ngt = n;
  }
  for (int c = tid; c < ncell; c += 256) cell_gt[c] = -1;
  __syncthreads();
  const int ngt = s_ngt;

  // ---- per-GT work: float32 corners, responsible cell, targets, tconf ----
  for (int t = tid; t < ngt; t += 256) {
    const T* g = tg + t * NL;
    float gc[2 * K];
#pragma unroll
    for (int k = 0; k < 2 * K; ++k) { gc[k] = (float)g[1 + k]; gt_c[t][k] = gc[k]; }
    T gx0 = g[1] * (T)a.nW, gy0 = g[2] * (T)a.nH;
    int gi0 = (int)gx0, gj0 = (int)gy0;
    gi0 = max(0, min(a.nW - 1, gi0));  // the reference would raise IndexError outside the grid
    gj0 = max(0, min(a.nH - 1, gj0));
#pragma unroll
    for (int k = 0; k < K; ++k) {
      gt_t[t][2 * k] = (float)(g[1 + 2 * k] * (T)a.nW - (T)gi0);
      gt_t[t][2 * k + 1] = (float)(g[2 + 2 * k] * (T)a.nH - (T)gj0);
    }
    int best_n = 0, pb = b, pa = 0;
    if (a.multi) {
      // anchor with the best IoU against the GT's 2D box, both centred at the origin (region_loss_multi.py:66-77)
      double gw = (double)g[NL - 2] * a.nW, gh = (double)g[NL - 1] * a.nH;
      double best_iou = 0.0;
      best_n = -1;
      for (int n = 0; n < a.nA; ++n) {
        double aw = (double)a.anchors[a.anchor_step * n], ah = (double)a.anchors[a.anchor_step * n + 1];
        double uw = fmax(aw / 2.0, gw / 2.0) - fmin(-aw / 2.0, -gw / 2.0);
        double uh = fmax(ah / 2.0, gh / 2.0) - fmin(-ah / 2.0, -gh / 2.0);
        double cw = aw + gw - uw, ch = ah + gh - uh;
        double iou = 0.0;
        if (!(cw <= 0 || ch <= 0)) {
          double carea = cw * ch;
          iou = carea / (aw * ah + gw * gh - carea);
        }
        if (iou > best_iou) { best_iou = iou; best_n = n; }
      }
      if (best_n < 0) best_n = a.nA - 1;  // Python negative index
      // pred_box = pred_corners[b*nAnchors + (-1)*nPixels + gj0*nW + gi0]: last anchor of the previous image,
      // wrapping to the last image of the batch for b == 0
      pb = (b + a.nB - 1) % a.nB;
      pa = a.nA - 1;
    }
    float pr[2 * K];
    decode_cell<K>(out, ((int64_t)(pb * a.nA + pa) * nCh) * hw + (int64_t)gj0 * a.nW + gi0, hw, gi0, gj0, a.nW, a.nH, pr);
    gt_conf[t] = corner_conf<K>(gc, pr, conf0_one);
    gt_cell[t] = (int)(best_n * hw + (int64_t)gj0 * a.nW + gi0);
    gt_cls[t] = (int)g[0];
  }
  __syncthreads();
  if (tid == 0)
    for (int t = 0; t < ngt; ++t) cell_gt[gt_cell[t]] = t;  // later GTs overwrite earlier ones
  __syncthreads();

  // ---- per-cell work ----
  float l_x = 0.f, l_y = 0.f, l_conf = 0.f, l_cls = 0.f;
  int n_prop = 0;
  for (int c = tid; c < ncell; c += 256) {
    const int an = c / (int)hw;
    const int rem = c - an * (int)hw;
    const int j = rem / a.nW, i = rem - j * a.nW;
    const int64_t base = ((int64_t)(b * a.nA + an) * nCh) * hw + rem;
    float raw[2 * K];
#pragma unroll
    for (int k = 0; k < 2 * K; ++k) raw[k] = out[base + k * hw];
    const float x0 = sigmoidf_(raw[0]), y0 = sigmoidf_(raw[1]);
    const float conf = sigmoidf_(out[base + (2 * K) * hw]);
    float pr[2 * K];
    pr[0] = (x0 + (float)i) / (float)a.nW;
    pr[1] = (y0 + (float)j) / (float)a.nH;
#pragma unroll
    for (int k = 1; k < K; ++k) {
      pr[2 * k] = (raw[2 * k] + (float)i) / (float)a.nW;
      pr[2 * k + 1] = (raw[2 * k + 1] + (float)j) / (float)a.nH;
    }
    float cur = 0.f;
    for (int t = 0; t < ngt; ++t) cur = fmaxf(cur, corner_conf<K>(gt_c[t], pr, conf0_all));
    float cmask = (cur > a.thresh) ? 0.f : a.noobject_scale;
    float tconf = 0.f;
    const int t = cell_gt[c];
    if (conf > 0.25f) ++n_prop;
    if (t >= 0) {
      cmask = a.object_scale;
      tconf = gt_conf[t];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float px = (k == 0) ? x0 : raw[2 * k], py = (k == 0) ? y0 : raw[2 * k + 1];
        float ex = px - gt_t[t][2 * k], ey = py - gt_t[t][2 * k + 1];
        l_x += a.coord_scale * ex * ex * 0.5f;
        l_y += a.coord_scale * ey * ey * 0.5f;
        float gx = a.coord_scale * ex, gy = a.coord_scale * ey;
        if (k == 0) { gx *= x0 * (1.f - x0); gy *= y0 * (1.f - y0); }
        grad[base + (2 * k) * hw] = gx;
        grad[base + (2 * k + 1) * hw] = gy;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 2 * K; ++k) grad[base + k * hw] = 0.f;
    }
    {
      // MSELoss(conf*sqrt(mask), tconf*sqrt(mask))/2 (region_loss.py:141,152)
      float sq = sqrtf(cmask);
      float e = conf * sq - tconf * sq;
      l_conf += e * e * 0.5f;
      grad[base + (2 * K) * hw] = a.conf_on ? e * sq * conf * (1.f - conf) : 0.f;
    }
    if (a.multi && t >= 0) {
      // class_scale * CrossEntropyLoss(sum) on the nC logits of a GT cell (region_loss_multi.py:169)
      float mx = -INFINITY;
      for (int q = 0; q < a.nC; ++q) mx = fmaxf(mx, out[base + (2 * K + 1 + q) * hw]);
      float se = 0.f;
      for (int q = 0; q < a.nC; ++q) se += expf(out[base + (2 * K + 1 + q) * hw] - mx);
      float lse = logf(se) + mx;
      int cls = gt_cls[t];
      for (int q = 0; q < a.nC; ++q) {
        float z = out[base + (2 * K + 1 + q) * hw];
        float p = expf(z - lse);
        grad[base + (2 * K + 1 + q) * hw] = a.class_scale * (p - (q == cls ? 1.f : 0.f));
        if (q == cls) l_cls += a.class_scale * (lse - z);
      }
    } else {
      for (int q = 0; q < a.nC; ++q) grad[base + (2 * K + 1 + q) * hw] = 0.f;
    }
  }

  // ---- workgroup reduction of the scalars ----
  float v[5] = {l_x, l_y, l_conf, l_cls, (float)n_prop};
#pragma unroll
  for (int q = 0; q < 5; ++q)
    for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off);
  const int lane = tid & 63, wid = tid >> 6;
  if (lane == 0) {
    red[0][wid] = v[0]; red[1][wid] = v[1]; red[2][wid] = v[2]; red[3][wid] = v[3]; red[4][wid] = v[4];
  }
  __syncthreads();
  if (tid == 0) {
    float* p = partials + (int64_t)b * 8;
    for (int q = 0; q < 5; ++q) p[q] = red[q][0] + red[q][1] + red[q][2] + red[q][3];
    int ncorrect = 0;
    for (int t = 0; t < ngt; ++t) ncorrect += gt_conf[t] > 0.5f ? 1 : 0;
    p[5] = (float)ngt;
    p[6] = (float)ncorrect;
    p[7] = 0.f;
  }
}

// stats[8] = {loss_x, loss_y, loss_conf, loss_cls, total, nGT, nCorrect, nProposals}
__global__ void region_loss_sum_kernel(const float* partials, int nB, int conf_on, int multi, float* stats) {
  const int q = threadIdx.x;
  __shared__ float s[8];
  if (q < 8) {
    float acc = 0.f;
    for (int b = 0; b < nB; ++b) acc += partials[b * 8 + q];
    s[q] = acc;
  }
  __syncthreads();
  if (q == 0) {
    float total = s[0] + s[1] + (multi ? s[3] : 0.f) + (conf_on ? s[2] : 0.f);
    stats[0] = s[0]; stats[1] = s[1]; stats[2] = s[2]; stats[3] = s[3];
    stats[4] = total; stats[5] = s[5]; stats[6] = s[6]; stats[7] = s[4];
  }
}

// Inference decode: one workgroup per image finds the cell with the largest confidence (first maximum in
// (cy, cx, anchor) scan order, strict '>' as utils.py:273) and emits its 2K+3 numbers + the confidence used.
// boxes[b] = {x0/w, y0/h, ..., det_conf, cls_max_conf, cls_max_id, conf}
template <int K>
__global__ void __launch_bounds__(256) region_decode_argmax_kernel(const float* __restrict__ out, int nA, int nC, int nH,
                                                                   int nW, int only_objectness, float* boxes) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nCh = 2 * K + 1 + nC;
  const int64_t hw = (int64_t)nH * nW;
  const int ncell = nA * nH * nW;
  float best = -INFINITY;
  int best_key = 0x7fffffff;  // scan-order key: (cy*nW + cx)*nA + anchor
  for (int key = tid; key < ncell; key += 256) {
    int an = key % nA, rem = key / nA;
    int64_t base = ((int64_t)(b * nA + an) * nCh) * hw + rem;
    float conf = sigmoidf_(out[base + (2 * K) * hw]);
    if (!only_objectness) {
      float mx = -INFINITY;
      for (int q = 0; q < nC; ++q) mx = fmaxf(mx, out[base + (2 * K + 1 + q) * hw]);
      float se = 0.f;
      for (int q = 0; q < nC; ++q) se += expf(out[base + (2 * K + 1 + q) * hw] - mx);
      conf = conf * (1.f / se);  // max softmax probability = exp(0)/se
    }
    if (conf > best) { best = conf; best_key = key; }  // keys ascend per thread: first maximum kept
  }
  __shared__ float sb[256];
  __shared__ int sk[256];
  sb[tid] = best; sk[tid] = best_key;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      float ob = sb[tid + off];
      int ok = sk[tid + off];
      if (ob > sb[tid] || (ob == sb[tid] && ok < sk[tid])) { sb[tid] = ob; sk[tid] = ok; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    int key = sk[0];
    // every confidence NaN (a diverged network): no cell ever compared greater, the key is still the sentinel.  Decode
    // cell 0 (in bounds) and leave conf = -inf in the last slot: the host raises, as the reference does (utils.py:296
    // returns a `box` that was never bound)
    if ((unsigned)key >= (unsigned)ncell) key = 0;
    int an = key % nA, rem = key / nA;
    int j = rem / nW, i = rem % nW;
    int64_t base = ((int64_t)(b * nA + an) * nCh) * hw + rem;
    float* o = boxes + (int64_t)b * (2 * K + 4);
    for (int k = 0; k < K; ++k) {
      float vx = out[base + (2 * k) * hw], vy = out[base + (2 * k + 1) * hw];
      if (k == 0) { vx = sigmoidf_(vx); vy = sigmoidf_(vy); }
      o[2 * k] = (vx + (float)i) / (float)nW;
      o[2 * k + 1] = (vy + (float)j) / (float)nH;
    }
    o[2 * K] = sigmoidf_(out[base + (2 * K) * hw]);
    float mx = -INFINITY;
    int arg = 0;
    for (int q = 0; q < nC; ++q) {
      float z = out[base + (2 * K + 1 + q) * hw];
      if (z > mx) { mx = z; arg = q; }
    }
    float se = 0.f;
    for (int q = 0; q < nC; ++q) se += expf(out[base + (2 * K + 1 + q) * hw] - mx);
    o[2 * K + 1] = 1.f / se;
    o[2 * K + 2] = (float)arg;
    o[2 * K + 3] = sb[0];
  }
}

// Dense decode for get_multi_region_boxes (utils_multi.py:266-382): every (cell, anchor) in the reference's scan order
// key = (cy*nW + cx)*nA + anchor -> rows[b][key] = {2K coords / (w,h), det_conf, cls_max_conf, cls_max_id, softmax[nC]}.
// The variable-length box lists (threshold, fallback box) are assembled on the host from this one tensor.
template <int K>
__global__ void __launch_bounds__(256) region_decode_all_kernel(const float* __restrict__ out, int nA, int nC, int nH,
                                                                int nW, float* rows) {
  const int b = blockIdx.y;
  const int nCh = 2 * K + 1 + nC;
  const int W = 2 * K + 3 + nC;
  const int64_t hw = (int64_t)nH * nW;
  const int ncell = nA * nH * nW;
  for (int key = blockIdx.x * blockDim.x + threadIdx.x; key < ncell; key += gridDim.x * blockDim.x) {
    int an = key % nA, rem = key / nA;
    int j = rem / nW, i = rem % nW;
    int64_t base = ((int64_t)(b * nA + an) * nCh) * hw + rem;
    float* o = rows + ((int64_t)b * ncell + key) * W;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float vx = out[base + (2 * k) * hw], vy = out[base + (2 * k + 1) * hw];
      if (k == 0) { vx = sigmoidf_(vx); vy = sigmoidf_(vy); }
      o[2 * k] = (vx + (float)i) / (float)nW;
      o[2 * k + 1] = (vy + (float)j) / (float)nH;
    }
    o[2 * K] = sigmoidf_(out[base + (2 * K) * hw]);
    float mx = -INFINITY;
    int arg = 0;
    for (int q = 0; q < nC; ++q) {
      float z = out[base + (2 * K + 1 + q) * hw];
      if (z > mx) { mx = z; arg = q; }
    }
    float se = 0.f;
    for (int q = 0; q < nC; ++q) se += expf(out[base + (2 * K + 1 + q) * hw] - mx);
    for (int q = 0; q < nC; ++q) o[2 * K + 3 + q] = expf(out[base + (2 * K + 1 + q) * hw] - mx) / se;
    o[2 * K + 1] = 1.f / se;
    o[2 * K + 2] = (float)arg;
  }
}

int ssp_region_decode_all_launch(const float* out, float* rows, int nB, int nA, int nC, int nH, int nW,
                                 int num_keypoints, hipStream_t stream) {
  SSP_CHECK_ARG(num_keypoints == 9, "region_decode_all: only num_keypoints == 9 is built (got %d)", num_keypoints);
  SSP_CHECK_ARG(nC >= 1 && nA >= 1, "region_decode_all: need at least one class and one anchor");
  SspProfScope prof(SSP_PROF_REGION, stream, 0.0);
  const int ncell = nA * nH * nW;
  hipLaunchKernelGGL((region_decode_all_kernel<9>), dim3(ssp_cdiv(ncell, 256), nB), dim3(256), 0, stream, out, nA, nC,
                     nH, nW, rows);
  SSP_CHECK_LAUNCH("region_decode_all");
  return SSP_OK;
}

int ssp_region_loss_launch(const float* out, const void* target, int target_is_f64, float* grad, float* partials,
                           float* stats, int nB, int nA, int nC, int nH, int nW, int num_keypoints,
                           float noobject_scale, float object_scale, float coord_scale, float class_scale, float thresh,
                           int conf_on, int multi, const float* anchors, int anchor_step, hipStream_t stream) {
  SSP_CHECK_ARG(num_keypoints == 9, "region_loss: only num_keypoints == 9 is built (got %d)", num_keypoints);
  SSP_CHECK_ARG(nA * nH * nW <= SSP_MAX_CELLS, "region_loss: more than %d cells per image", SSP_MAX_CELLS);
  SSP_CHECK_ARG(!multi || (anchors != nullptr && anchor_step >= 2), "region_loss: multi-object mode needs anchors");
  RegionArgs a;
  a.nB = nB; a.nA = nA; a.nC = nC; a.nH = nH; a.nW = nW;
  a.noobject_scale = noobject_scale; a.object_scale = object_scale; a.coord_scale = coord_scale;
  a.class_scale = class_scale; a.thresh = thresh; a.conf_on = conf_on; a.multi = multi;
  a.tgt_stride = SSP_MAX_GT * (2 * num_keypoints + 3);
  a.anchors = anchors; a.anchor_step = anchor_step;
  SspProfScope prof(SSP_PROF_REGION, stream, 0.0);
  if (target_is_f64)
    hipLaunchKernelGGL((region_loss_kernel<9, double>), dim3(nB), dim3(256), 0, stream, out, (const double*)target, grad,
                       partials, a);
  else
    hipLaunchKernelGGL((region_loss_kernel<9, float>), dim3(nB), dim3(256), 0, stream, out, (const float*)target, grad,
                       partials, a);
  SSP_CHECK_LAUNCH("region_loss");
  hipLaunchKernelGGL(region_loss_sum_kernel, dim3(1), dim3(64), 0, stream, partials, nB, conf_on, multi, stats);
  SSP_CHECK_LAUNCH("region_loss_sum");
  return SSP_OK;
}

int ssp_region_decode_argmax_launch(const float* out, float* boxes, int nB, int nA, int nC, int nH, int nW,
                                    int num_keypoints, int only_objectness, hipStream_t stream) {
  SSP_CHECK_ARG(num_keypoints == 9, "region_decode: only num_keypoints == 9 is built (got %d)", num_keypoints);
  SSP_CHECK_ARG(nC >= 1, "region_decode: need at least one class");
  SspProfScope prof(SSP_PROF_REGION, stream, 0.0);
  hipLaunchKernelGGL((region_decode_argmax_kernel<9>), dim3(nB), dim3(256), 0, stream, out, nA, nC, nH, nW,
                     only_objectness, boxes);
  SSP_CHECK_LAUNCH("region_decode_argmax");
  return SSP_OK;
}
