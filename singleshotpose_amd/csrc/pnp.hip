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
#include "ssp_common.h"

#define PNP_MAXN 16

__device__ static void jacobi_eig(double* A, double* V, int n) {
  // cyclic Jacobi on symmetric A (n x n, row-major, destroyed: eigenvalues end on the diagonal); V columns = eigenvectors
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) {
      diag += A[i * n + i] * A[i * n + i];
      for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
    }
    if (off <= 1e-34 * diag || off == 0.0) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        double apq = A[p * n + q];
        if (apq == 0.0) continue;
        double app = A[p * n + p], aqq = A[q * n + q];
        double tau = (aqq - app) / (2.0 * apq);
        double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
        for (int k = 0; k < n; ++k) {
          double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq;
          V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
}

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

__device__ static void mat3_mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

// rotation-vector norm^2 of R (for the relative-step stopping rule: |params| = |(rvec, t)|)
__device__ static double rvec_norm2(const double* R) {
  double c = 0.5 * (R[0] + R[4] + R[8] - 1.0);
  c = fmin(1.0, fmax(-1.0, c));
  double th = acos(c);
  return th * th;
}

__device__ static double reproj(const double* R, const double* t, const double* X, const double* uv, int N, double fx,
                                double fy, double cx, double cy, double* JtJ, double* Jte) {
  // returns |err|^2; when JtJ != nullptr also accumulates the 6x6 normal equations for (dw, dt)
  if (JtJ) {
    for (int i = 0; i < 36; ++i) JtJ[i] = 0.0;
    for (int i = 0; i < 6; ++i) Jte[i] = 0.0;
  }
  double e2 = 0.0;
  for (int i = 0; i < N; ++i) {
    const double* P = X + 3 * i;
    double rx = R[0] * P[0] + R[1] * P[1] + R[2] * P[2];
    double ry = R[3] * P[0] + R[4] * P[1] + R[5] * P[2];
    double rz = R[6] * P[0] + R[7] * P[1] + R[8] * P[2];
    double x = rx + t[0], y = ry + t[1], z = rz + t[2];
    double iz = 1.0 / z;
    double eu = fx * x * iz + cx - uv[2 * i], ev = fy * y * iz + cy - uv[2 * i + 1];
    e2 += eu * eu + ev * ev;
    if (JtJ) {
      // d(u,v)/d(x,y,z)
      double ux = fx * iz, uz = -fx * x * iz * iz, vy = fy * iz, vz = -fy * y * iz * iz;
      // d(x,y,z)/dw = -[R P]x ; d/dt = I
      double Ju[6], Jv[6];
      Ju[0] = uz * ry;            Ju[1] = ux * rz - uz * rx;  Ju[2] = -ux * ry;
      Jv[0] = -vy * rz + vz * ry; Jv[1] = -vz * rx;           Jv[2] = vy * rx;
      Ju[3] = ux; Ju[4] = 0.0; Ju[5] = uz;
      Jv[3] = 0.0; Jv[4] = vy; Jv[5] = vz;
      for (int a = 0; a < 6; ++a) {
        Jte[a] += Ju[a] * eu + Jv[a] * ev;
        for (int b = 0; b < 6; ++b) JtJ[a * 6 + b] += Ju[a] * Ju[b] + Jv[a] * Jv[b];
      }
    }
  }
  return e2;
}

__device__ static bool solve6(const double* A_, const double* b_, double* x) {
  double A[36], b[6];
  for (int i = 0; i < 36; ++i) A[i] = A_[i];
  for (int i = 0; i < 6; ++i) b[i] = b_[i];
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    double mx = fabs(A[k * 6 + k]);
    for (int i = k + 1; i < 6; ++i)
      if (fabs(A[i * 6 + k]) > mx) { mx = fabs(A[i * 6 + k]); piv = i; }
    if (mx < 1e-300) return false;
    if (piv != k) {
      for (int j = 0; j < 6; ++j) { double tmp = A[k * 6 + j]; A[k * 6 + j] = A[piv * 6 + j]; A[piv * 6 + j] = tmp; }
      double tmp = b[k]; b[k] = b[piv]; b[piv] = tmp;
    }
    for (int i = k + 1; i < 6; ++i) {
      double f = A[i * 6 + k] / A[k * 6 + k];
      for (int j = k; j < 6; ++j) A[i * 6 + j] -= f * A[k * 6 + j];
      b[i] -= f * b[k];
    }
  }
  for (int i = 5; i >= 0; --i) {
    double s = b[i];
    for (int j = i + 1; j < 6; ++j) s -= A[i * 6 + j] * x[j];
    x[i] = s / A[i * 6 + i];
  }
  return true;
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

  // ---- DLT ----
  double L[144], V[144];
  for (int i = 0; i < 144; ++i) L[i] = 0.0;
  for (int i = 0; i < N; ++i) {
    double xn = (uv[2 * i] - cx) / fx, yn = (uv[2 * i + 1] - cy) / fy;
    double P[4] = {X[3 * i], X[3 * i + 1], X[3 * i + 2], 1.0};
    double r0[12], r1[12];
    for (int k = 0; k < 4; ++k) {
      r0[k] = P[k]; r0[4 + k] = 0.0; r0[8 + k] = -xn * P[k];
      r1[k] = 0.0;  r1[4 + k] = P[k]; r1[8 + k] = -yn * P[k];
    }
    for (int a = 0; a < 12; ++a)
      for (int b = 0; b < 12; ++b) L[a * 12 + b] += r0[a] * r0[b] + r1[a] * r1[b];
  }
  jacobi_eig(L, V, 12);
  int imin = 0;
  for (int i = 1; i < 12; ++i)
    if (L[i * 12 + i] < L[imin * 12 + imin]) imin = i;
  double RR[9], tt[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) RR[r * 3 + c] = V[(r * 4 + c) * 12 + imin];
    tt[r] = V[(r * 4 + 3) * 12 + imin];
  }
  double det = RR[0] * (RR[4] * RR[8] - RR[5] * RR[7]) - RR[1] * (RR[3] * RR[8] - RR[5] * RR[6]) +
               RR[2] * (RR[3] * RR[7] - RR[4] * RR[6]);
  if (det < 0.0) {
    for (int i = 0; i < 9; ++i) RR[i] = -RR[i];
    for (int i = 0; i < 3; ++i) tt[i] = -tt[i];
  }
  double sc = 0.0;
  for (int i = 0; i < 9; ++i) sc += RR[i] * RR[i];
  sc = sqrt(sc);
  // nearest rotation: R = RR (RR^T RR)^(-1/2)
  double S[9], E[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) S[i * 3 + j] = RR[i] * RR[j] + RR[3 + i] * RR[3 + j] + RR[6 + i] * RR[6 + j];
  jacobi_eig(S, E, 3);
  double Sinv[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += E[i * 3 + k] * E[j * 3 + k] / sqrt(fmax(S[k * 3 + k], 1e-300));
      Sinv[i * 3 + j] = s;
    }
  double R[9], t[3];
  mat3_mul(RR, Sinv, R);
  for (int i = 0; i < 3; ++i) t[i] = tt[i] * (sqrt(3.0) / sc);

  // ---- Levenberg-Marquardt refinement of the pixel reprojection error ----
  double JtJ[36], Jte[6];
  double err = reproj(R, t, X, uv, N, fx, fy, cx, cy, JtJ, Jte);
  int lambda_lg10 = -3, iters = 0;
  for (int guard = 0; guard < 200; ++guard) {
    double A[36], d[6];
    double lambda = pow(10.0, (double)lambda_lg10);
    for (int i = 0; i < 36; ++i) A[i] = JtJ[i];
    for (int i = 0; i < 6; ++i) A[i * 6 + i] *= 1.0 + lambda;
    if (!solve6(A, Jte, d)) break;
    double w[3] = {-d[0], -d[1], -d[2]}, dR[9], Rn[9], tn[3];
    so3_exp(w, dR);
    mat3_mul(dR, R, Rn);
    for (int i = 0; i < 3; ++i) tn[i] = t[i] - d[3 + i];
    double err_n = reproj(Rn, tn, X, uv, N, fx, fy, cx, cy, nullptr, nullptr);
    if (err_n > err) {
      if (++lambda_lg10 > 16) break;
      continue;
    }
    lambda_lg10 = max(lambda_lg10 - 1, -16);
    double pn = sqrt(rvec_norm2(R) + t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    double dn = 0.0;
    for (int i = 0; i < 6; ++i) dn += d[i] * d[i];
    dn = sqrt(dn);
    for (int i = 0; i < 9; ++i) R[i] = Rn[i];
    for (int i = 0; i < 3; ++i) t[i] = tn[i];
    if (++iters >= max_iter || dn < 1.1920928955078125e-07 * pn) break;
    err = reproj(R, t, X, uv, N, fx, fy, cx, cy, JtJ, Jte);
  }
  double* o = Rt + (int64_t)id * 12;
  for (int i = 0; i < 9; ++i) o[i] = R[i];
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
