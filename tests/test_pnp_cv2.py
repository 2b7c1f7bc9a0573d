"""PnP against real OpenCV - runs only where `cv2` is importable (it is not in the build image nor on the GPU box:
un-versioned `opencv-python`, /root/reference/README.md:30, no wheel, no network).  The first box that has it closes
SURVEY.md section 8(c)'s "parity unpinned" for a17: ssp_pnp_batched and oracle/pnp_ref.py against
cv2.solvePnP(..., flags=SOLVEPNP_ITERATIVE) + cv2.Rodrigues exactly as /root/reference/utils.py:86-100 calls them, on
the 64 synthetic LINEMOD-range poses of SURVEY.md section 8(d) config 4; bar = north_star's 1e-3 px reprojection."""
import importlib.util

import numpy as np
import pytest


def _real_cv2():
    spec = importlib.util.find_spec('cv2')
    if spec is None or (spec.origin or '').replace('\\', '/').endswith('dropin/cv2.py'):
        return None
    import cv2
    return cv2 if hasattr(cv2, 'solvePnP') and hasattr(cv2, 'Rodrigues') else None


def _poses(n=64, noise_px=1.0, seed=7):
    rs = np.random.RandomState(seed)
    half = np.array([0.038, 0.039, 0.046])
    X = np.concatenate([np.zeros((1, 3)), np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) * half], 0)
    K = np.array([[572.4114, 0, 325.2611], [0, 573.5704, 242.0489], [0, 0, 1.0]])
    out = []
    for _ in range(n):
        ax = rs.standard_normal(3)
        ax /= np.linalg.norm(ax)
        ang = rs.uniform(0, np.pi / 3)
        Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx.dot(Kx)
        t = np.array([rs.uniform(-.1, .1), rs.uniform(-.1, .1), rs.uniform(0.6, 1.2)])
        c = X.dot(R.T) + t
        uv = np.stack([K[0, 0] * c[:, 0] / c[:, 2] + K[0, 2], K[1, 1] * c[:, 1] / c[:, 2] + K[1, 2]], 1)
        out.append((uv + rs.uniform(-noise_px, noise_px, uv.shape)).astype(np.float32))
    return X.astype(np.float32), np.stack(out), K.astype(np.float32)


def _reproject(X, R, t, K):
    c = X.astype(np.float64).dot(R.T) + t.reshape(1, 3)
    return np.stack([K[0, 0] * c[:, 0] / c[:, 2] + K[0, 2], K[1, 1] * c[:, 1] / c[:, 2] + K[1, 2]], 1)


def _cv2_pnp(cv2, X, uv, K):
    # utils.py:86-100, verbatim call shape
    dist = np.zeros((8, 1), dtype='float32')
    _, R_exp, t = cv2.solvePnP(X, np.ascontiguousarray(uv[:, :2]).reshape((-1, 1, 2)), K, dist)
    R, _ = cv2.Rodrigues(R_exp)
    return R, t


@pytest.mark.skipif(_real_cv2() is None, reason="cv2 (opencv-python) is not installed here: PnP parity stays unpinned")
def test_oracle_pnp_matches_opencv():
    from oracle.pnp_ref import solve_pnp_ref
    cv2 = _real_cv2()
    X, uvs, K = _poses()
    for uv in uvs:
        R0, t0 = _cv2_pnp(cv2, X, uv, K)
        R1, t1 = solve_pnp_ref(X, uv, K)
        assert np.abs(_reproject(X, R0, t0, K) - _reproject(X, R1, t1, K)).max() < 1e-3


@pytest.mark.gpu
@pytest.mark.skipif(_real_cv2() is None, reason="cv2 (opencv-python) is not installed here: PnP parity stays unpinned")
def test_hip_pnp_matches_opencv():
    from singleshotpose_amd.utils import pnp_batched
    cv2 = _real_cv2()
    X, uvs, K = _poses()
    Rs, ts = pnp_batched(np.broadcast_to(X, (len(uvs),) + X.shape), uvs, K)
    for uv, R1, t1 in zip(uvs, Rs, ts):
        R0, t0 = _cv2_pnp(cv2, X, uv, K)
        assert np.abs(_reproject(X, R0, t0, K) - _reproject(X, R1, t1, K)).max() < 1e-3
