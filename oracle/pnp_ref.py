"""numpy restatement of cv2.solvePnP(SOLVEPNP_ITERATIVE) + cv2.Rodrigues - TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the algorithm lives in OpenCV, a third-party dependency of the reference that is neither vendored
nor version-pinned (README.md:30 lists "opencv-python" only; call sites /root/reference/utils.py:94-99).  cv2 is not
installable in this environment and the reference has no test or golden vector for a PnP output, so this file
restates OpenCV's published non-planar algorithm (calib3d: DLT initialisation, det>0 sign fix, SVD projection on
SO(3), translation rescale, then Levenberg-Marquardt on the pixel reprojection error with CvLevMarq's schedule:
lambda 1e-3, x10 on a worse step, /10 on a better one, <= 20 accepted steps, stop at |dp|/|p| < FLT_EPSILON) and is
checked by synthetic round trips (project a known pose, recover it) instead of by reference outputs, plus a
cross-check against an independent minimiser of the same objective (scipy MINPACK LM, tests/test_host.py).
"""
import numpy as np


def rodrigues(rvec):
    th = np.linalg.norm(rvec)
    if th < 1e-12:
        return np.eye(3)
    k = rvec / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx.dot(Kx)


def _residual_jac(R, t, X, uv, fx, fy, cx, cy):
    q = X.dot(R.T)
    p = q + t
    iz = 1.0 / p[:, 2]
    err = np.stack([fx * p[:, 0] * iz + cx - uv[:, 0], fy * p[:, 1] * iz + cy - uv[:, 1]], 1).reshape(-1)
    J = np.zeros((2 * len(X), 6))
    for i in range(len(X)):
        dudp = np.array([fx * iz[i], 0, -fx * p[i, 0] * iz[i] ** 2])
        dvdp = np.array([0, fy * iz[i], -fy * p[i, 1] * iz[i] ** 2])
        qx = np.array([[0, -q[i, 2], q[i, 1]], [q[i, 2], 0, -q[i, 0]], [-q[i, 1], q[i, 0], 0]])
        dpdw = -qx
        J[2 * i, :3], J[2 * i, 3:] = dudp.dot(dpdw), dudp
        J[2 * i + 1, :3], J[2 * i + 1, 3:] = dvdp.dot(dpdw), dvdp
    return err, J


def solve_pnp_ref(X, uv, K, max_iter=20):
    """X (N,3), uv (N,2), K (3,3) -> R (3,3), t (3,1) float64."""
    X = np.asarray(X, np.float64)
    uv = np.asarray(uv, np.float64)
    K = np.asarray(K, np.float64)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    xn, yn = (uv[:, 0] - cx) / fx, (uv[:, 1] - cy) / fy
    N = len(X)
    L = np.zeros((2 * N, 12))
    Xh = np.concatenate([X, np.ones((N, 1))], 1)
    L[0::2, 0:4], L[0::2, 8:12] = Xh, -xn[:, None] * Xh
    L[1::2, 4:8], L[1::2, 8:12] = Xh, -yn[:, None] * Xh
    _, _, Vt = np.linalg.svd(L.T.dot(L))
    RRt = Vt[11].reshape(3, 4)
    RR, tt = RRt[:, :3], RRt[:, 3]
    if np.linalg.det(RR) < 0:
        RR, tt = -RR, -tt
    sc = np.linalg.norm(RR)
    U, _, Vt3 = np.linalg.svd(RR)
    R = U.dot(Vt3)
    t = tt * (np.linalg.norm(R) / sc)

    err, J = _residual_jac(R, t, X, uv, fx, fy, cx, cy)
    e2 = err.dot(err)
    lam_lg, iters = -3, 0
    for _ in range(200):
        JtJ, Jte = J.T.dot(J), J.T.dot(err)
        A = JtJ.copy()
        A[np.diag_indices(6)] *= 1.0 + 10.0 ** lam_lg
        d = np.linalg.solve(A, Jte)
        Rn, tn = rodrigues(-d[:3]).dot(R), t - d[3:]
        err_n, J_n = _residual_jac(Rn, tn, X, uv, fx, fy, cx, cy)
        e2n = err_n.dot(err_n)
        if e2n > e2:
            lam_lg += 1
            if lam_lg > 16:
                break
            continue
        lam_lg = max(lam_lg - 1, -16)
        th = np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))
        pn = np.sqrt(th * th + t.dot(t))
        R, t, err, J, e2 = Rn, tn, err_n, J_n, e2n
        iters += 1
        if iters >= max_iter or np.linalg.norm(d) < 1.1920928955078125e-07 * pn:
            break
    return R, t.reshape(3, 1)


def project(X, R, t, K):
    p = np.asarray(X, np.float64).dot(R.T) + np.asarray(t).reshape(1, 3)
    uvw = p.dot(np.asarray(K, np.float64).T)
    return uvw[:, :2] / uvw[:, 2:3]
