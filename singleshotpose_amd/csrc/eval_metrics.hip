// Evaluation maths of valid.py:152-177 on the device, batched over poses.
//
// pose_errors: per pose pair (gt, predicted), over the N mesh vertices
//   pixel_dist  = mean_i || proj(K [R|t]_gt v_i) - proj(K [R|t]_pr v_i) ||    (utils.py:40-45 compute_projection: the
//                 projections are stored as float32, valid.py:163-164 takes the float32 norm)
//   vertex_dist = mean_i || [R|t]_gt v_i - [R|t]_pr v_i ||                     (utils.py:47-48, valid.py:168-172, float64)
//   trans_dist  = || t_gt - t_pr ||                                            (valid.py:148)
//   angle_dist  = deg(acos((trace(R_gt R_pr^T) - 1) / 2))                       (utils.py:31-35; NaN when rounding pushes
//                 the argument above 1 for identical rotations, as numpy's arccos does)
// pts_diameter: largest pairwise distance of the mesh (utils.py:50-58), O(N^2) pairs.
//
// One workgroup per pose; vertices stream from HBM/L2 (N*24 bytes, shared by every pose), the means are fp64 tree
// reductions in LDS.  Latency-bound sizes (N ~ 6 k for LINEMOD ape): reported as time per call, not a roofline.
#include "ssp_common.h"

__device__ __forceinline__ double block_sum(double v, double* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) red[tid] += red[tid + off];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(256) pose_errors_kernel(const double* __restrict__ verts, int N,
                                                          const double* __restrict__ Rt_gt,
                                                          const double* __restrict__ Rt_pr,
                                                          const double* __restrict__ Kmat, int k_per_pose,
                                                          double* __restrict__ out) {
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  // Rt stored as R (9, row-major) | t (3): the layout ssp_pnp_batched writes
  double Rg[9], tg[3], Rp[9], tp[3], K[9], Pg[12], Pp[12];
#pragma unroll
  for (int i = 0; i < 9; ++i) { Rg[i] = Rt_gt[b * 12 + i]; Rp[i] = Rt_pr[b * 12 + i]; K[i] = Kmat[(k_per_pose ? b * 9 : 0) + i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { tg[i] = Rt_gt[b * 12 + 9 + i]; tp[i] = Rt_pr[b * 12 + 9 + i]; }
  // P = K [R|t]  (3x4), as internal_calibration.dot(transformation)
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double sg = 0.0, sp = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        sg += K[r * 3 + k] * (c < 3 ? Rg[k * 3 + c] : tg[k]);
        sp += K[r * 3 + k] * (c < 3 ? Rp[k * 3 + c] : tp[k]);
      }
      Pg[r * 4 + c] = sg;
      Pp[r * 4 + c] = sp;
    }
  double s2d = 0.0, s3d = 0.0;
  for (int i = tid; i < N; i += 256) {
    const double x = verts[i * 3], y = verts[i * 3 + 1], z = verts[i * 3 + 2];
    double cg[3], cp[3], vg[3], vp[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      cg[r] = Pg[r * 4] * x + Pg[r * 4 + 1] * y + Pg[r * 4 + 2] * z + Pg[r * 4 + 3];
      cp[r] = Pp[r * 4] * x + Pp[r * 4 + 1] * y + Pp[r * 4 + 2] * z + Pp[r * 4 + 3];
      vg[r] = Rg[r * 3] * x + Rg[r * 3 + 1] * y + Rg[r * 3 + 2] * z + tg[r];
      vp[r] = Rp[r * 3] * x + Rp[r * 3 + 1] * y + Rp[r * 3 + 2] * z + tp[r];
    }
    // float32 projections, float32 difference and norm (compute_projection's dtype='float32' array)
    const float dx = __fsub_rn((float)(cg[0] / cg[2]), (float)(cp[0] / cp[2]));
    const float dy = __fsub_rn((float)(cg[1] / cg[2]), (float)(cp[1] / cp[2]));
    s2d += (double)__fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    const double ex = vg[0] - vp[0], ey = vg[1] - vp[1], ez = vg[2] - vp[2];
    s3d += sqrt(ex * ex + ey * ey + ez * ez);
  }
  s2d = block_sum(s2d, red);
  s3d = block_sum(s3d, red);
  if (tid == 0) {
    out[b * 4 + 0] = s2d / (double)N;
    out[b * 4 + 1] = s3d / (double)N;
    const double dx = tg[0] - tp[0], dy = tg[1] - tp[1], dz = tg[2] - tp[2];
    out[b * 4 + 2] = sqrt(dx * dx + dy * dy + dz * dz);
    double trace = 0.0;   // trace(R_gt R_pr^T) = sum_ij Rg[i][j] Rp[i][j]
#pragma unroll
    for (int i = 0; i < 3; ++i) trace += Rg[i * 3] * Rp[i * 3] + Rg[i * 3 + 1] * Rp[i * 3 + 1] + Rg[i * 3 + 2] * Rp[i * 3 + 2];
    out[b * 4 + 3] = acos((trace - 1.0) / 2.0) * (180.0 / 3.14159265358979323846);
  }
}

// max_ij |p_i - p_j|^2, exact fp64 with the reference's summation order ((dx*dx + dy*dy) + dz*dz, no contraction):
// each thread owns one i, all threads sweep j through LDS tiles; the result is order-independent (a max), so the
// atomic on the bit pattern of the non-negative double is exact.
__global__ void __launch_bounds__(256) pts_diameter_kernel(const double* __restrict__ pts, int N,
                                                           unsigned long long* __restrict__ best_bits) {
  __shared__ double tile[256 * 3];
  const int tid = threadIdx.x;
  const int i = blockIdx.x * 256 + tid;
  double xi = 0.0, yi = 0.0, zi = 0.0;
  if (i < N) { xi = pts[i * 3]; yi = pts[i * 3 + 1]; zi = pts[i * 3 + 2]; }
  double best = 0.0;
  // pairs (i, j >= i) only, as the reference's triangular sweep: start at this workgroup's own tile
  for (int j0 = blockIdx.x * 256; j0 < N; j0 += 256) {
    const int j = j0 + tid;
    if (j < N) { tile[tid * 3] = pts[j * 3]; tile[tid * 3 + 1] = pts[j * 3 + 1]; tile[tid * 3 + 2] = pts[j * 3 + 2]; }
    __syncthreads();
    const int cnt = min(256, N - j0);
    if (i < N) {
      for (int k = 0; k < cnt; ++k) {
        const double dx = xi - tile[k * 3], dy = yi - tile[k * 3 + 1], dz = zi - tile[k * 3 + 2];
        const double d2 = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
        best = d2 > best ? d2 : best;
      }
    }
    __syncthreads();
  }
  // wave max, then one atomic per wave
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(best, off, 64);
    best = o > best ? o : best;
  }
  if ((tid & 63) == 0) atomicMax(best_bits, (unsigned long long)__double_as_longlong(best));
}

__global__ void diameter_finish_kernel(const unsigned long long* best_bits, double* out) {
  out[0] = sqrt(__longlong_as_double((long long)best_bits[0]));
}

int ssp_pose_errors_launch(const double* verts, int N, const double* Rt_gt, const double* Rt_pr, const double* K,
                           int k_per_pose, int n, double* out, hipStream_t stream) {
  SSP_CHECK_ARG(verts && Rt_gt && Rt_pr && K && out, "pose_errors: null buffer");
  SSP_CHECK_ARG(N > 0 && n > 0, "pose_errors: need N > 0 vertices and n > 0 poses");
  SspProfScope prof(SSP_PROF_REGION, stream, 0.0);
  hipLaunchKernelGGL(pose_errors_kernel, dim3(n), dim3(256), 0, stream, verts, N, Rt_gt, Rt_pr, K, k_per_pose, out);
  SSP_CHECK_LAUNCH("pose_errors");
  return SSP_OK;
}

int ssp_pts_diameter_launch(const double* pts, int N, double* out, double* scratch, hipStream_t stream) {
  SSP_CHECK_ARG(pts && out && scratch, "pts_diameter: null buffer");
  SSP_CHECK_ARG(N > 0, "pts_diameter: empty point set");
  SspProfScope prof(SSP_PROF_REGION, stream, 0.0);
  if (hipMemsetAsync(scratch, 0, 8, stream) != hipSuccess) {
    ssp_set_error("pts_diameter: hipMemsetAsync failed");
    return SSP_ERR_HIP;
  }
  hipLaunchKernelGGL(pts_diameter_kernel, dim3(ssp_cdiv(N, 256)), dim3(256), 0, stream, pts, N,
                     reinterpret_cast<unsigned long long*>(scratch));
  SSP_CHECK_LAUNCH("pts_diameter");
  hipLaunchKernelGGL(diameter_finish_kernel, dim3(1), dim3(1), 0, stream,
                     reinterpret_cast<const unsigned long long*>(scratch), out);
  SSP_CHECK_LAUNCH("pts_diameter_finish");
  return SSP_OK;
}
