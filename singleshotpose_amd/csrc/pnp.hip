// Batched Perspective-n-Point pose recovery in fp64, one problem per thread.
//
// Replaces cv2.solvePnP(objectPoints, imagePoints, K, zeros(8,1)) with the default SOLVEPNP_ITERATIVE flag followed by
// cv2.Rodrigues (/root/reference/utils.py:86-100; callers valid.py:152-153, train.py:203-204).  OpenCV is a
// third-party dependency of the reference that is not vendored (README.md:30 "opencv-python", unpinned), so this
// restates the published algorithm of its non-planar branch:
//   1. normalise image points with K^-1 (zero distortion);
//   2. DLT: smallest right singular vector of the 2N x 12 system -> [R|t] up to scale, sign fixed by det(R) > 0,
//      R projected on SO(3), t rescaled by |R_orth| / |R_dlt|;
//   3. Levenberg-Marquardt on the pixel reprojection error (<= max_iter accepted steps, lambda0 = 1e-3, x10 / /10,
//      diag(JtJ) scaling, stop when |step| / |params| < FLT_EPSILON), as CvLevMarq does for cvFindExtrinsicCameraParams2.
// The rotation update uses the left-multiplicative exponential map instead of OpenCV's Rodrigues-vector chart; both
// descend to the same least-squares minimum.
//
// Everything a thread touches is indexed at compile time (fully unrolled loops over 12 / 6 / 3), so the 12 x 12 normal
// matrix, its Cholesky factor and the 6 x 6 LM system live in VGPRs: the first version kept them in dynamically indexed
// arrays, i.e. in scratch memory, and took 2.1 ms for one pose.  The DLT null vector comes from inverse iteration on
// the Cholesky-factored (lightly shifted) normal matrix instead of a full Jacobi eigen-decomposition (the smallest
// eigen-pair is all that is used; the LM refinement that follows makes the result independent of how it was found),
// the nearest rotation from the Newton iteration for the polar factor, the LM step from an unpivoted 6 x 6 Cholesky.
#include "ssp_common.h"

#define PNP_MAXN 16

#define TRI(a, b) ((a) * ((a) + 1) / 2 + (b))   // packed lower triangle, a >= b

__device__ static void so3_exp(const double* w, double* R) {
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double th = sqrt(th2);
  double a, b;  // R = I + a [w]x + b [w]x^2
  if (th < 1e-8) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; }
  else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
  double wx = w[0], wy = w[1], wz = w[2];
  R[0] = 1.0 - b * (wy * wy + wz * wz); R[1] = -a * wz + b * wx * wy;        R[2] = a * wy + b * wx * wz;
  R[3] = a * wz + b * wx * wy;          R[4] = 1.0 - b * (wx * wx + wz * wz); R[5] = -a * wx + b * wy * wz;
  R[6] = -a * wy + b * wx * wz;         R[7] = a * wx + b * wy * wz;          R[8] = 1.0 - b * (wx * wx + wy * wy);
}

__device__ __forceinline__ void mat3_mul(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

// rotation-vector norm^2 of R (for the relative-step stopping rule: |params| = |(rvec, t)|)
__device__ static double rvec_norm2(const double* R) {
  double c = 0.5 * (R[0] + R[4] + R[8] - 1.0);
  c = fmin(1.0, fmax(-1.0, c));
  double th = acos(c);
  return th * th;
}

// |err|^2 of the pixel reprojection; WITH_J also accumulates the 6x6 normal equations (lower triangle, packed) for (dw, dt)
template <bool WITH_J>
__device__ __forceinline__ double reproj(const double* R, const double* t, const double* __restrict__ X,
                                         const double* __restrict__ uv, int N, double fx, double fy, double cx,
                                         double cy, double* JtJ, double* Jte) {
  if (WITH_J) {
#pragma unroll
    for (int i = 0; i < 21; ++i) JtJ[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) Jte[i] = 0.0;
  }
  double e2 = 0.0;
  for (int i = 0; i < N; ++i) {
    const double P0 = X[3 * i], P1 = X[3 * i + 1], P2 = X[3 * i + 2];
    double rx = R[0] * P0 + R[1] * P1 + R[2] * P2;
    double ry = R[3] * P0 + R[4] * P1 + R[5] * P2;
    double rz = R[6] * P0 + R[7] * P1 + R[8] * P2;
    double x = rx + t[0], y = ry + t[1], z = rz + t[2];
    double iz = 1.0 / z;
    double eu = fx * x * iz + cx - uv[2 * i], ev = fy * y * iz + cy - uv[2 * i + 1];
    e2 += eu * eu + ev * ev;
    if (WITH_J) {
      // d(u,v)/d(x,y,z)
      double ux = fx * iz, uz = -fx * x * iz * iz, vy = fy * iz, vz = -fy * y * iz * iz;
      // d(x,y,z)/dw = -[R P]x ; d/dt = I
      double Ju[6], Jv[6];
      Ju[0] = uz * ry;            Ju[1] = ux * rz - uz * rx;  Ju[2] = -ux * ry;
      Jv[0] = -vy * rz + vz * ry; Jv[1] = -vz * rx;           Jv[2] = vy * rx;
      Ju[3] = ux; Ju[4] = 0.0; Ju[5] = uz;
      Jv[3] = 0.0; Jv[4] = vy; Jv[5] = vz;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        Jte[a] += Ju[a] * eu + Jv[a] * ev;
#pragma unroll
        for (int b = 0; b <= a; ++b) JtJ[TRI(a, b)] += Ju[a] * Ju[b] + Jv[a] * Jv[b];
      }
    }
  }
  return e2;
}

// in-place Cholesky of a packed lower-triangular SPD matrix (compile-time n); false when a pivot is not positive
template <int n>
__device__ __forceinline__ bool chol_packed(double* A) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < n; ++j) {
    double s = A[TRI(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= A[TRI(j, k)] * A[TRI(j, k)];
    ok = ok && (s > 0.0);
    const double g = sqrt(fmax(s, 1e-300));
    const double ig = 1.0 / g;
    A[TRI(j, j)] = g;
#pragma unroll
    for (int i = j + 1; i < n; ++i) {
      double v = A[TRI(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= A[TRI(i, k)] * A[TRI(j, k)];
      A[TRI(i, j)] = v * ig;
    }
  }
  return ok;
}

// x <- (G G^T)^-1 x
template <int n>
__device__ __forceinline__ void chol_solve(const double* G, double* x) {
#pragma unroll
  for (int i = 0; i < n; ++i) {
    double s = x[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= G[TRI(i, k)] * x[k];
    x[i] = s / G[TRI(i, i)];
  }
#pragma unroll
  for (int i = n - 1; i >= 0; --i) {
    double s = x[i];
#pragma unroll
    for (int k = i + 1; k < n; ++k) s -= G[TRI(k, i)] * x[k];
    x[i] = s / G[TRI(i, i)];
  }
}

__global__ void __launch_bounds__(64) pnp_kernel(const double* __restrict__ pts3d, const double* __restrict__ pts2d,
                                                 const double* __restrict__ Kmat, double* __restrict__ Rt, int n, int N,
                                                 int max_iter) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n) return;
  const double* X = pts3d + (int64_t)id * N * 3;
  const double* uv = pts2d + (int64_t)id * N * 2;
  const double* Km = Kmat + (int64_t)id * 9;
  const double fx = Km[0], fy = Km[4], cx = Km[2], cy = Km[5];

  // ---- DLT: L = M^T M of the 2N x 12 system (packed lower triangle) ----
  double L[78];
#pragma unroll
  for (int i = 0; i < 78; ++i) L[i] = 0.0;
  for (int i = 0; i < N; ++i) {
    const double xn = (uv[2 * i] - cx) / fx, yn = (uv[2 * i + 1] - cy) / fy;
    const double P[4] = {X[3 * i], X[3 * i + 1], X[3 * i + 2], 1.0};
    double r0[12], r1[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      r0[k] = P[k]; r0[4 + k] = 0.0; r0[8 + k] = -xn * P[k];
      r1[k] = 0.0;  r1[4 + k] = P[k]; r1[8 + k] = -yn * P[k];
    }
#pragma unroll
    for (int a = 0; a < 12; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) L[TRI(a, b)] += r0[a] * r0[b] + r1[a] * r1[b];
  }
  // smallest eigenvector by inverse iteration on L + mu I (mu ~ 1e-11 of the mean eigenvalue keeps the factorisation
  // positive when the data are exact and L is singular to rounding; convergence factor (l_min + mu) / (l_2 + mu))
  double tr = 0.0;
#pragma unroll
  for (int a = 0; a < 12; ++a) tr += L[TRI(a, a)];
  const double mu = 1e-11 * tr / 12.0;
#pragma unroll
  for (int a = 0; a < 12; ++a) L[TRI(a, a)] += mu;
  chol_packed<12>(L);
  double v[12] = {0.3010, -0.5236, 0.1729, 0.4142, -0.2718, 0.1618, 0.5772, -0.3679, 0.2236, -0.1414, 0.6931, 0.3333};
  for (int it = 0; it < 16; ++it) {
    chol_solve<12>(L, v);
    double nn = 0.0;
#pragma unroll
    for (int a = 0; a < 12; ++a) nn += v[a] * v[a];
    const double inv = 1.0 / sqrt(nn);
#pragma unroll
    for (int a = 0; a < 12; ++a) v[a] *= inv;
  }
  double RR[9], tt[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) RR[r * 3 + c] = v[r * 4 + c];
    tt[r] = v[r * 4 + 3];
  }
  double det = RR[0] * (RR[4] * RR[8] - RR[5] * RR[7]) - RR[1] * (RR[3] * RR[8] - RR[5] * RR[6]) +
               RR[2] * (RR[3] * RR[7] - RR[4] * RR[6]);
  if (det < 0.0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) RR[i] = -RR[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) tt[i] = -tt[i];
  }
  double sc = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) sc += RR[i] * RR[i];
  sc = sqrt(sc);
  // nearest rotation = orthogonal polar factor of RR: Newton iteration Q <- (Q + Q^-T) / 2 from Q0 = RR * sqrt(3) / |RR|
  double R[9], t[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = RR[i] * (sqrt(3.0) / sc);
  for (int it = 0; it < 12; ++it) {
    double C[9];   // cofactor matrix = det * Q^-T
    C[0] = R[4] * R[8] - R[5] * R[7]; C[1] = R[5] * R[6] - R[3] * R[8]; C[2] = R[3] * R[7] - R[4] * R[6];
    C[3] = R[2] * R[7] - R[1] * R[8]; C[4] = R[0] * R[8] - R[2] * R[6]; C[5] = R[1] * R[6] - R[0] * R[7];
    C[6] = R[1] * R[5] - R[2] * R[4]; C[7] = R[2] * R[3] - R[0] * R[5]; C[8] = R[0] * R[4] - R[1] * R[3];
    const double d = R[0] * C[0] + R[1] * C[1] + R[2] * C[2];
    const double id_ = 1.0 / d;
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = 0.5 * (R[i] + C[i] * id_);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = tt[i] * (sqrt(3.0) / sc);

  // ---- Levenberg-Marquardt refinement of the pixel reprojection error ----
  double JtJ[21], Jte[6];
  double err = reproj<true>(R, t, X, uv, N, fx, fy, cx, cy, JtJ, Jte);
  int lambda_lg10 = -3, iters = 0;
  for (int guard = 0; guard < 200; ++guard) {
    double A[21], d[6];
    const double lambda = pow(10.0, (double)lambda_lg10);
#pragma unroll
    for (int i = 0; i < 21; ++i) A[i] = JtJ[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) { A[TRI(i, i)] *= 1.0 + lambda; d[i] = Jte[i]; }
    if (!chol_packed<6>(A)) break;
    chol_solve<6>(A, d);
    double w[3] = {-d[0], -d[1], -d[2]}, dR[9], Rn[9], tn[3];
    so3_exp(w, dR);
    mat3_mul(dR, R, Rn);
#pragma unroll
    for (int i = 0; i < 3; ++i) tn[i] = t[i] - d[3 + i];
    double err_n = reproj<false>(Rn, tn, X, uv, N, fx, fy, cx, cy, nullptr, nullptr);
    if (!(err_n <= err)) {
      if (++lambda_lg10 > 16) break;
      continue;
    }
    lambda_lg10 = max(lambda_lg10 - 1, -16);
    double pn = sqrt(rvec_norm2(R) + t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    double dn = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) dn += d[i] * d[i];
    dn = sqrt(dn);
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = Rn[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = tn[i];
    if (++iters >= max_iter || dn < 1.1920928955078125e-07 * pn) break;
    err = reproj<true>(R, t, X, uv, N, fx, fy, cx, cy, JtJ, Jte);
  }
  double* o = Rt + (int64_t)id * 12;
#pragma unroll
  for (int i = 0; i < 9; ++i) o[i] = R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) o[9 + i] = t[i];
}

int ssp_pnp_batched_launch(const double* pts3d, const double* pts2d, const double* K, double* Rt, int n, int N,
                           int max_iter, hipStream_t stream) {
  SSP_CHECK_ARG(N >= 6 && N <= PNP_MAXN, "pnp: the DLT initialisation needs 6..%d non-coplanar points (got %d)", PNP_MAXN, N);
  SSP_CHECK_ARG(n > 0, "pnp: empty batch");
  SspProfScope prof(SSP_PROF_REGION, stream, 0.0);
  hipLaunchKernelGGL(pnp_kernel, dim3(ssp_cdiv(n, 64)), dim3(64), 0, stream, pts3d, pts2d, K, Rt, n, N, max_iter);
  SSP_CHECK_LAUNCH("pnp");
  return SSP_OK;
}
