"""Oracle (TEST INFRASTRUCTURE ONLY): the torch_data batch path, numpy.

Follows (reference file:line):
  resample matrix                       lib/batched_dataset/data_transform.py:57-76   (_compute_resample_matrix)
  pinhole->pinhole bilinear resampler   lib/batched_dataset/data_transform.py:79-144  (_resample_images_batched)
  crop matrices per view                lib/batched_dataset/data_transform.py:147-212 (_gen_crop_matrices)
  per-sequence crop                     lib/batched_dataset/data_transform.py:215-283 (_perspective_crop_images)
  time-step batching                    run_inference_torch_data.py:39-85             (_unpack_batched_data)

Pinned by tests/golden/torch_data.npz (made by oracle/gen_goldens.py from the reference's own functions; the
first four are importable in the build container).  _unpack_batched_data lives in a script that imports
lib.common.hand_skinning (pytorch3d, absent): it is restated from the text only - index bookkeeping, no arithmetic.

dtype note: the reference keeps float32 through the crop-matrix chain (np.linalg.inv of a float32 array is
float32); this restatement follows numpy's own promotion by calling the same numpy operations on the same dtypes,
so it tracks the reference to a few float32 ulps (LAPACK call order aside).
"""
from typing import Dict, List, Tuple

import numpy as np

from . import ref_camera


def k_matrix(f, c) -> np.ndarray:
    return np.array([[f[0], 0, c[0]], [0, f[1], c[1]], [0, 0, 1]])


def compute_resample_matrix(cam_orig: dict, cam_new: dict) -> np.ndarray:
    k_inv_new = np.eye(4, 4)
    k_inv_new[0:3, 0:3] = np.linalg.inv(k_matrix(cam_new["f"], cam_new["c"]))
    k_orig = np.eye(4, 4)
    k_orig[0:3, 0:3] = k_matrix(cam_orig["f"], cam_orig["c"])
    xf = k_orig @ np.linalg.inv(cam_orig["T"]) @ cam_new["T"] @ k_inv_new
    return xf.astype(np.float32)


def _crop_camera_f32(cam: dict, pts_world: np.ndarray, size, mirror_x: bool, focal_multiplier: float) -> dict:
    """gen_crop_parameters_from_points (lib/common/crop.py:31-82, lib/common/affine.py:34-76) with camera_angle 0,
    operating on the float32 arrays of the torch_data path with numpy's own promotion rules."""
    t = cam["T"]
    w2e = np.linalg.inv(t)
    center = (pts_world.min(axis=0) + pts_world.max(axis=0)) / 2.0
    c_eye = (center.reshape(-1, 3) @ w2e[:3, :3].T).reshape(center.shape) + w2e[:3, 3]
    eps = 5.43e-20

    def nrm(v):
        return v / np.maximum(eps, np.sqrt(np.sum(v * v, axis=-1, keepdims=True)))
    a = nrm(np.array([0, 0, 1], dtype=center.dtype))
    b = nrm(c_eye / np.linalg.norm(c_eye))
    v = np.cross(a, b)
    s = np.linalg.norm(v)
    k = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=v.dtype)
    d_r = np.eye(3) + k + (k @ k) * (1 - np.dot(a, b)) / max(s * s, 1e-15)
    e2w = np.linalg.inv(w2e)
    new_e2w = e2w.copy()
    new_e2w[0:3, 0:3] = e2w[0:3, 0:3] @ d_r @ np.eye(3)         # roll by 0 degrees
    new_w2e = np.linalg.inv(new_e2w)
    if mirror_x:
        mx = np.eye(4, dtype=np.float32)
        mx[0, 0] = -1
        new_w2e = mx @ new_w2e
    pe = (pts_world.reshape(-1, 3) @ new_w2e[:3, :3].T) + new_w2e[:3, 3]
    ndc = pe[..., 0:2] / pe[..., 2:]
    cxy = (np.array([size[0], size[1]], dtype=pe.dtype) - 1) / 2
    fxy = cxy / np.absolute(ndc).max()
    if np.any(pe[..., 2:] < 0.0001) or np.any(fxy < 5):
        raise ValueError("Unable to create crop camera", fxy)
    return {"w": int(size[0]), "h": int(size[1]), "f": tuple(focal_multiplier * fxy), "c": tuple(cxy), "k": None,
            "T": np.linalg.inv(new_w2e)}


def gen_crop_matrices(orig_extrinsics: np.ndarray, orig_intrinsics: np.ndarray, crop_points: np.ndarray, mirror_image: bool,
                      crop_size: Tuple[int, int], focal_multiplier: float = 0.95):
    n_views = orig_extrinsics.shape[0]
    ext = np.empty([n_views, 4, 4], np.float32)
    intr = np.empty([n_views, 3, 3], np.float32)
    res = np.empty([n_views, 4, 4], np.float32)
    for v in range(n_views):
        k = orig_intrinsics[v]
        cam_orig = {"w": crop_size[0], "h": crop_size[1], "f": (k[0, 0], k[1, 1]), "c": (k[0, 2], k[1, 2]), "k": None,
                    "T": np.linalg.inv(orig_extrinsics[v])}
        cam_new = _crop_camera_f32(cam_orig, crop_points, crop_size, mirror_image, focal_multiplier)
        ext[v] = np.linalg.inv(cam_new["T"])
        intr[v] = k_matrix(cam_new["f"], cam_new["c"])
        res[v] = compute_resample_matrix(cam_orig, cam_new)
    return ext, intr, res


def resample_images_batched(images_orig: np.ndarray, out_hw: Tuple[int, int], resample_xfs: np.ndarray) -> np.ndarray:
    """Dense restatement (every pixel computed, masked afterwards) of the reference's scatter formulation; the
    per-pixel arithmetic and its dtypes are the reference's.  images_orig float32 [n,H,W] -> float32 [n,h,w]."""
    n, h_orig, w_orig = images_orig.shape
    h_new, w_new = out_hw
    r = resample_xfs[:, 0:3, 0:3]
    t = resample_xfs[:, 0:3, 3]
    grid = np.ones((h_new, w_new, 3), dtype=np.int32)
    grid[:, :, 0:2] = np.mgrid[0:w_new:1, 0:h_new:1].transpose(2, 1, 0)
    mul = np.tensordot(r, grid, axes=([2], [2])).transpose(0, 2, 3, 1)
    g = (mul.reshape(n, -1, 3) + np.expand_dims(t, axis=1)).reshape(mul.shape)
    with np.errstate(divide="ignore", invalid="ignore"):
        hc = g[..., 0:2] / g[..., 2:]
    x, y = hc[..., 0], hc[..., 1]
    mask = (x >= 0) & (x < (w_orig - 1)) & (y >= 0) & (y < (h_orig - 1))
    xs, ys = np.where(mask, x, 0.0), np.where(mask, y, 0.0)
    x0 = xs.astype(np.int32)
    y0 = ys.astype(np.int32)
    x1, y1 = x0 + 1, y0 + 1
    idx = np.arange(n)[:, None, None]
    f00, f01 = images_orig[idx, y0, x0], images_orig[idx, y1, x0]
    f10, f11 = images_orig[idx, y0, x1], images_orig[idx, y1, x1]
    val = (f00 * (x1 - xs) * (y1 - ys) + f10 * (xs - x0) * (y1 - ys) + f01 * (x1 - xs) * (ys - y0)
           + f11 * (xs - x0) * (ys - y0)) / ((x1 - x0) * (y1 - y0))
    out = np.zeros((n, h_new, w_new), np.float32)
    out[mask] = val[mask]
    return out


def perspective_crop_images(orig_images: np.ndarray, orig_extrinsics: np.ndarray, orig_intrinsics: np.ndarray,
                            crop_points: np.ndarray, hand_idx: int, crop_size: Tuple[int, int]):
    n_frames, n_views = orig_images.shape[:2]
    ext = np.empty([n_frames, n_views, 4, 4], np.float32)
    intr = np.empty([n_frames, n_views, 3, 3], np.float32)
    res = np.empty([n_frames, n_views, 4, 4], np.float32)
    for f in range(n_frames):
        ext[f], intr[f], res[f] = gen_crop_matrices(orig_extrinsics[f], orig_intrinsics[f], crop_points[f], hand_idx == 1,
                                                    crop_size)
    img = resample_images_batched(orig_images.reshape(-1, *orig_images.shape[2:]).astype(np.float32), crop_size,
                                  res.reshape(-1, 4, 4))
    img = img.reshape(n_frames, n_views, *crop_size) / 255
    return img, ext, intr


def unpack_batched_data(left_images: np.ndarray, intrinsics: np.ndarray, extrinsics_xf: np.ndarray, hand_idx: np.ndarray,
                        axes: np.ndarray, rest: np.ndarray, seq_mode: str) -> List[Dict[str, np.ndarray]]:
    """[bs,seq,V,...] batch -> one model call per time step (run_inference_torch_data.py:39-85)."""
    bs, seq_len = left_images.shape[:2]
    if seq_mode not in ("multiv", "singlev"):
        raise ValueError(f"Unknown sequence mode: {seq_mode}")
    nv = 2 if seq_mode == "multiv" else 1
    steps = []
    for i in range(seq_len):
        steps.append({
            "images": left_images[:, i, 0:nv].reshape(bs * nv, *left_images.shape[3:]),
            "intrinsics": intrinsics[:, i, 0:nv].reshape(bs * nv, 3, 3),
            "extrinsics": extrinsics_xf[:, i, 0:nv].reshape(bs * nv, 4, 4),
            "sample_range": np.array([(b * nv, (b + 1) * nv) for b in range(bs)], np.int64),
            "memory_idx": np.arange(bs, dtype=np.int64),
            "use_memory": np.full(bs, i != 0),
            "hand_idx": hand_idx[:, i].astype(np.int64),
            "axes": axes[:, i], "rest": rest[:, i]})
    return steps
