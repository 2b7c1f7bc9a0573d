"""Head decode, PnP and evaluation helpers - host-side mirror of the reference's utils.py.

On the hot path and therefore on HIP kernels (no CPU fallback):
  get_region_boxes   /root/reference/utils.py:216-296   -> ssp_region_decode_argmax
  pnp                /root/reference/utils.py:86-100    -> ssp_pnp_batched (cv2.solvePnP ITERATIVE + cv2.Rodrigues)
Kept as plain Python / numpy with the reference's names and argument meaning because train.py / valid.py do
`from utils import *`: projection / metric helpers (utils.py:31-84), file-format readers (utils.py:299-358),
logging, convert2cpu.  cv2 is never imported: the only use the reference makes of it is inside pnp().
"""
import math
import os
import struct
import sys
import time

import numpy as np
import torch

from . import _lib


def makedirs(path):
    if not os.path.exists(path):
        os.makedirs(path)


def get_all_files(directory):
    files = []
    for f in os.listdir(directory):
        p = os.path.join(directory, f)
        if os.path.isfile(p):
            files.append(p)
        else:
            files.extend(get_all_files(p))
    return files


def calcAngularDistance(gt_rot, pr_rot):
    rot_diff = np.dot(gt_rot, np.transpose(pr_rot))
    return np.rad2deg(np.arccos((np.trace(rot_diff) - 1.0) / 2.0))


def get_camera_intrinsic(u0, v0, fx, fy):
    return np.array([[fx, 0.0, u0], [0.0, fy, v0], [0.0, 0.0, 1.0]])


def compute_projection(points_3D, transformation, internal_calibration):
    """4xN homogeneous points -> 2xN float32 pixel projections through K [R|t] (utils.py:40-45)."""
    cam = internal_calibration.dot(transformation).dot(points_3D)
    proj = np.zeros((2, points_3D.shape[1]), dtype='float32')
    proj[0, :] = cam[0, :] / cam[2, :]
    proj[1, :] = cam[1, :] / cam[2, :]
    return proj


def compute_transformation(points_3D, transformation):
    return transformation.dot(points_3D)


def calc_pts_diameter(pts):
    """Largest pairwise distance of an (N,3) point set (utils.py:50-58).

    Same arithmetic as the reference, element for element: difference of the two points, square, sum over x,y,z, max,
    one square root - so the result is bit-identical to its row-at-a-time loop (only blocked over rows).  On the GPU
    box big meshes (valid.py:72 passes the 5.8 k vertices of the object model) go through ssp_pts_diameter, which is
    pinned bit-exact to the same golden value (tests/test_gpu_head.py)."""
    pts = np.asarray(pts)
    n = pts.shape[0]
    if n >= 1024 and pts.dtype == np.float64 and torch.cuda.is_available():
        return calc_pts_diameter_gpu(pts)
    best = -1
    step = max(1, (1 << 21) // max(n, 1))
    for i in range(0, n, step):
        diff = pts[i:i + step, None, :] - pts[None, :, :]
        d2 = (diff * diff).sum(axis=2).max()
        best = max(best, d2)
    return math.sqrt(best) if n else -1


def adi(pts_est, pts_gt):
    from scipy import spatial
    nn_dists, _ = spatial.cKDTree(pts_est).query(pts_gt, k=1)
    return nn_dists.mean()


def get_3D_corners(vertices):
    """4x8 homogeneous bounding-box corners in the order (min,min,min) ... (max,max,max) (utils.py:66-84)."""
    mn = [np.min(vertices[i, :]) for i in range(3)]
    mx = [np.max(vertices[i, :]) for i in range(3)]
    corners = np.array([[(mx if a else mn)[0], (mx if b else mn)[1], (mx if c else mn)[2]]
                        for a in (0, 1) for b in (0, 1) for c in (0, 1)])
    return np.concatenate((np.transpose(corners), np.ones((1, 8))), axis=0)


def pnp_batched(points_3D, points_2D, cameraMatrix, max_iter=20):
    """n independent PnP problems on the GPU.

    points_3D (n,N,3), points_2D (n,N,2), cameraMatrix (3,3) or (n,3,3) -> R (n,3,3), t (n,3,1) float64 ndarrays.
    """
    p3 = torch.as_tensor(np.ascontiguousarray(points_3D, dtype=np.float64))
    p2 = torch.as_tensor(np.ascontiguousarray(np.asarray(points_2D)[..., :2], dtype=np.float64))
    n, N = p3.shape[0], p3.shape[1]
    assert p2.shape[0] == n and p2.shape[1] == N, 'points 3D and points 2D must have same number of vertices'
    K = np.asarray(cameraMatrix, dtype=np.float64)
    if K.ndim == 2:
        K = np.broadcast_to(K, (n, 3, 3))
    Kt = torch.as_tensor(np.array(K, dtype=np.float64, order='C', copy=True)).contiguous()
    if not torch.cuda.is_available():
        raise RuntimeError("pnp runs on the MI355X HIP kernel only (no CPU fallback; cv2 is not used)")
    dev = torch.device('cuda', torch.cuda.current_device())
    p3, p2, Kt = p3.to(dev), p2.to(dev), Kt.to(dev)
    Rt = torch.empty(n, 12, dtype=torch.float64, device=dev)
    _lib.call('ssp_pnp_batched', p3.data_ptr(), p2.data_ptr(), Kt.data_ptr(), Rt.data_ptr(), n, N, max_iter,
              torch.cuda.current_stream().cuda_stream)
    Rt = Rt.cpu().numpy()
    return Rt[:, :9].reshape(n, 3, 3).copy(), Rt[:, 9:].reshape(n, 3, 1).copy()


def _to_dev_f64(a):
    if not torch.cuda.is_available():
        raise RuntimeError("the batched evaluation maths runs on the MI355X HIP kernels only (no CPU fallback)")
    t = a if torch.is_tensor(a) else torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64))
    return t.to(device=torch.device('cuda', torch.cuda.current_device()), dtype=torch.float64).contiguous()


def pose_errors_batched(vertices, R_gt, t_gt, R_pr, t_pr, internal_calibration):
    """valid.py:146-172 for n pose pairs in one launch.

    vertices (4,N) homogeneous or (3,N) (the mesh valid.py:60-61 loads), R_* (n,3,3), t_* (n,3,1) or (n,3),
    K (3,3) or (n,3,3) - numpy arrays or tensors (device tensors, e.g. straight from the PnP kernel, are used in
    place) -> float64 ndarray (n,4): [pixel_dist (errs_2d), vertex_dist (errs_3d), trans_dist, angle_dist in degrees].
    """
    v = _to_dev_f64(vertices)
    if v.dim() != 2 or v.size(0) not in (3, 4):
        raise ValueError("vertices must be (3,N) or (4,N)")
    v = v[:3].t().contiguous()
    Rg, Rp = _to_dev_f64(R_gt).reshape(-1, 9), _to_dev_f64(R_pr).reshape(-1, 9)
    tg, tp = _to_dev_f64(t_gt).reshape(-1, 3), _to_dev_f64(t_pr).reshape(-1, 3)
    n = Rg.size(0)
    if not (Rp.size(0) == tg.size(0) == tp.size(0) == n):
        raise ValueError("R_gt, t_gt, R_pr, t_pr must hold the same number of poses")
    K = _to_dev_f64(internal_calibration).reshape(-1, 9)
    if K.size(0) not in (1, n):
        raise ValueError("internal_calibration must be (3,3) or (n,3,3)")
    Rt_gt = torch.cat((Rg, tg), dim=1).contiguous()
    Rt_pr = torch.cat((Rp, tp), dim=1).contiguous()
    out = torch.empty(n, 4, dtype=torch.float64, device=v.device)
    _lib.call('ssp_pose_errors', v.data_ptr(), v.size(0), Rt_gt.data_ptr(), Rt_pr.data_ptr(), K.data_ptr(),
              1 if (K.size(0) == n and n > 1) else 0, n, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return out.cpu().numpy()


def calc_pts_diameter_gpu(pts):
    """calc_pts_diameter (utils.py:50-58) on the device: exact fp64 maximum over all N(N+1)/2 pairs, one launch
    instead of the reference's N numpy passes (minutes for a 6 k vertex mesh)."""
    p = _to_dev_f64(pts)
    if p.dim() != 2 or p.size(1) != 3:
        raise ValueError("pts must be (N,3)")
    out = torch.empty(1, dtype=torch.float64, device=p.device)
    scratch = torch.empty(1, dtype=torch.float64, device=p.device)
    _lib.call('ssp_pts_diameter', p.data_ptr(), p.size(0), out.data_ptr(), scratch.data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    return float(out.item())


def pnp(points_3D, points_2D, cameraMatrix):
    """(N,3) object points, (N,2) image points, (3,3) K -> R (3,3), t (3,1) float64 (utils.py:86-100)."""
    assert points_3D.shape[0] == points_2D.shape[0], 'points 3D and points 2D must have same number of vertices'
    R, t = pnp_batched(np.asarray(points_3D)[None], np.asarray(points_2D)[None], cameraMatrix)
    return R[0], t[0]


def get_2d_bb(box, size):
    pts = np.reshape(box, [-1, 2])
    w = np.max(pts[:, 0]) - np.min(pts[:, 0])
    h = np.max(pts[:, 1]) - np.min(pts[:, 1])
    return [box[0] * size, box[1] * size, w * size, h * size]


def compute_2d_bb(pts):
    min_x, max_x = np.min(pts[0, :]), np.max(pts[0, :])
    min_y, max_y = np.min(pts[1, :]), np.max(pts[1, :])
    return [(max_x + min_x) / 2.0, (max_y + min_y) / 2.0, max_x - min_x, max_y - min_y]


def compute_2d_bb_from_orig_pix(pts, size):
    min_x, max_x = np.min(pts[0, :]) / 640.0, np.max(pts[0, :]) / 640.0
    min_y, max_y = np.min(pts[1, :]) / 480.0, np.max(pts[1, :]) / 480.0
    return [(max_x + min_x) / 2.0 * size, (max_y + min_y) / 2.0 * size, (max_x - min_x) * size, (max_y - min_y) * size]


def corner_confidences(gt_corners, pr_corners, th=80, sharpness=2, im_width=640, im_height=480):
    """(2K x nA) GT and predicted corners -> (nA,) mean confidence (utils.py:138-165).

    API mirror only: the loss kernel (csrc/region.hip) computes this in-kernel.
    """
    nA = gt_corners.size(1)
    dist = (gt_corners - pr_corners).t().contiguous().view(nA, -1, 2)
    scale = torch.tensor([float(im_width), float(im_height)], dtype=dist.dtype, device=dist.device)
    d = torch.sqrt(((dist * scale) ** 2).sum(dim=2))
    conf = torch.exp(sharpness * (1 - d / th)) - 1
    conf0 = math.exp(sharpness) - 1
    return ((d < th).to(conf.dtype) * conf / conf0).mean(dim=1)


def corner_confidence(gt_corners, pr_corners, th=80, sharpness=2, im_width=640, im_height=480):
    """(2K,) GT list and predicted corners -> scalar confidence; normaliser exp(s)-1+1e-5 (utils.py:167-187)."""
    pr = torch.as_tensor(pr_corners, dtype=torch.float32)
    dist = (torch.as_tensor([float(g) for g in gt_corners], dtype=torch.float32) - pr.cpu()).view(-1, 2)
    scale = torch.tensor([float(im_width), float(im_height)])
    d = torch.sqrt(((dist * scale) ** 2).sum(dim=1))
    conf = torch.exp(sharpness * (1.0 - d / th)) - 1
    conf0 = math.exp(sharpness) - 1 + 1e-5
    return ((d < th).float() * conf / conf0).mean()


def sigmoid(x):
    return 1.0 / (math.exp(-x) + 1.)


def softmax(x):
    x = torch.exp(x - torch.max(x))
    return x / x.sum()


def fix_corner_order(corners2D_gt):
    order = [0, 1, 3, 5, 7, 2, 4, 6, 8]
    out = np.zeros((9, 2), dtype='float32')
    for dst, src in enumerate(order):
        out[dst, :] = corners2D_gt[src, :]
    return out


def convert2cpu(gpu_matrix):
    return torch.FloatTensor(gpu_matrix.size()).copy_(gpu_matrix)


def convert2cpu_long(gpu_matrix):
    return torch.LongTensor(gpu_matrix.size()).copy_(gpu_matrix)


_BOX_PINNED = {}


def region_boxes_batched(output, num_classes, num_keypoints, num_anchors=1, only_objectness=1):
    """Per-image best cell: (B, 2K+4) device tensor {2K coords, det_conf, cls_max_conf, cls_max_id, conf}."""
    if output.dim() == 3:
        output = output.unsqueeze(0)
    if not output.is_cuda:
        raise RuntimeError("get_region_boxes runs on the MI355X HIP kernel only: got a %s tensor (no CPU fallback)" % output.device)
    assert output.size(1) == (2 * num_keypoints + 1 + num_classes) * num_anchors
    out = output.detach().to(torch.float32).contiguous()
    B, h, w = out.size(0), out.size(2), out.size(3)
    boxes = torch.empty(B, 2 * num_keypoints + 4, dtype=torch.float32, device=out.device)
    _lib.call('ssp_region_decode_argmax', out.data_ptr(), boxes.data_ptr(), B, num_anchors, num_classes, h, w,
              num_keypoints, 1 if only_objectness else 0, torch.cuda.current_stream().cuda_stream)
    return boxes


def get_region_boxes(output, num_classes, num_keypoints, only_objectness=1, validation=True):
    """Best box of the whole batch: list of 2K+3 zero-dim tensors (utils.py:216-296).

    The reference keeps ONE box for the batch: the first cell in (b, cy, cx) order whose confidence is strictly
    larger than every earlier one (utils.py:262-288).  The per-image arg-max runs on the GPU; picking the first
    best image is a B-element host step after the single device->host copy.
    """
    dev_boxes = region_boxes_batched(output, num_classes, num_keypoints, 1, only_objectness)
    # one device->host copy of B x (2K+4) floats into a cached pinned buffer (88 bytes for valid.py's batch of 1)
    key = (tuple(dev_boxes.shape), dev_boxes.device.index)
    host = _BOX_PINNED.get(key)
    if host is None:
        if len(_BOX_PINNED) > 8:
            _BOX_PINNED.clear()
        host = _BOX_PINNED[key] = torch.empty(dev_boxes.shape, dtype=torch.float32).pin_memory()
    host.copy_(dev_boxes, non_blocking=True)
    torch.cuda.current_stream(dev_boxes.device).synchronize()
    K = num_keypoints
    confs = host[:, 2 * K + 3].tolist()
    best = 0
    for b in range(1, len(confs)):
        if confs[b] > confs[best]:
            best = b
    if not confs[best] > -sys.maxsize:
        # no confidence compared greater than the reference's -sys.maxsize start value (every one NaN: a diverged
        # network): the reference's loop never binds `box` and its `return box` raises this (utils.py:229,262-296)
        raise UnboundLocalError("local variable 'box' referenced before assignment (get_region_boxes: no cell has a "
                                "confidence > -sys.maxsize; the network output is NaN)")
    row = host[best].clone()
    box = list(row[:2 * K + 2].unbind(0))
    box.append(row[2 * K + 2].long())
    return box


def read_truths(lab_path, num_keypoints=9):
    num_labels = 2 * num_keypoints + 3
    if os.path.getsize(lab_path):
        truths = np.loadtxt(lab_path)
        return truths.reshape(truths.size // num_labels, num_labels)
    return np.array([])


def read_truths_args(lab_path, num_keypoints=9):
    # keeps the reference's packing: the first 2K+1 numbers of each row, back to back (utils.py:308-315)
    num_labels = 2 * num_keypoints + 1
    truths = read_truths(lab_path)
    return np.array([truths[i][j] for i in range(truths.shape[0]) for j in range(num_labels)])


def read_pose(lab_path):
    if os.path.getsize(lab_path):
        return np.loadtxt(lab_path)
    return np.array([])


def load_class_names(namesfile):
    with open(namesfile, 'r') as fp:
        return [line.rstrip() for line in fp.readlines()]


def image2torch(img):
    arr = np.frombuffer(img.tobytes(), dtype=np.uint8).reshape(img.height, img.width, 3)
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous().view(1, 3, img.height, img.width).float().div(255.0)


def read_data_cfg(datacfg):
    options = {'gpus': '0', 'num_workers': '10'}
    with open(datacfg, 'r') as fp:
        for line in fp.readlines():
            line = line.strip()
            if line == '':
                continue
            key, value = line.split('=')
            options[key.strip()] = value.strip()
    return options


def scale_bboxes(bboxes, width, height):
    import copy
    dets = copy.deepcopy(bboxes)
    for d in dets:
        d[0], d[1], d[2], d[3] = d[0] * width, d[1] * height, d[2] * width, d[3] * height
    return dets


def file_lines(thefilepath):
    count = 0
    with open(thefilepath, 'rb') as f:
        while True:
            buf = f.read(8192 * 1024)
            if not buf:
                break
            count += buf.count(b'\n')
    return count


def get_image_size(fname):
    """(width, height) of a PNG / GIF / JPEG from its header, None otherwise (utils.py:381-414; imghdr-free)."""
    with open(fname, 'rb') as fh:
        head = fh.read(24)
        if len(head) != 24:
            return None
        if head[:8] == b'\x89PNG\r\n\x1a\n':
            return struct.unpack('>ii', head[16:24])
        if head[:6] in (b'GIF87a', b'GIF89a'):
            return struct.unpack('<HH', head[6:10])
        if head[:2] == b'\xff\xd8':
            try:
                fh.seek(0)
                size, ftype = 2, 0
                while not 0xc0 <= ftype <= 0xcf:
                    fh.seek(size, 1)
                    byte = fh.read(1)
                    while ord(byte) == 0xff:
                        byte = fh.read(1)
                    ftype = ord(byte)
                    size = struct.unpack('>H', fh.read(2))[0] - 2
                fh.seek(1, 1)
                height, width = struct.unpack('>HH', fh.read(4))
                return width, height
            except Exception:
                return None
    return None


def logging(message):
    print('%s %s' % (time.strftime("%Y-%m-%d %H:%M:%S", time.localtime()), message))
