"""Oracle (TEST INFRASTRUCTURE ONLY): camera geometry, crop-camera generation and the
fisheye->pinhole crop resampler, numpy float64 like the reference.

Follows (reference file:line):
  pinhole unproject / fisheye project / Fisheye62 distortion  lib/common/camera.py:61-85,122-143,296-329
  world<->eye                                                 lib/common/camera.py:296-306, lib/common/affine.py:11-19
  warp coordinate map + depth mask                            lib/tracker/tracker.py:61-89
  look-at, two-vector rotation                                lib/common/affine.py:22-76
  crop intrinsics / crop camera                               lib/common/crop.py:15-82
  view ranking + crop points                                  lib/tracker/perspective_crop.py:19-180
  network input packing                                       lib/tracker/tracker.py:315-368
Cameras are plain dicts: {"w","h","f":(fx,fy),"c":(cx,cy),"k":8 coeffs or None,"T":cam_to_world 4x4}
with "k" ordered k1,k2,k3,k4,p1,p2,k5,k6 (lib/common/camera.py:109-116).

cv2.remap itself is NOT available (opencv absent from the image): `remap_bilinear`
restates two arithmetics - exact float bilinear, and OpenCV's documented 8-bit
path (coordinates rounded to 1/32 px, 15-bit fixed-point weights, rounded u8) -
PARITY UNPINNED for that step.
"""
import math
from typing import Dict, List, Optional

import numpy as np

from . import ref_fk


# ----------------------------------------------------------------------------- projection
def pinhole_window_to_eye(cam: dict, w: np.ndarray) -> np.ndarray:
    q = (np.asarray(w, np.float64) - np.asarray(cam["c"])) / np.asarray(cam["f"])
    v = np.concatenate([q, np.ones(q.shape[:-1] + (1,))], -1)
    d = np.maximum(5.43e-20, np.sqrt((v * v).sum(-1, keepdims=True)))       # affine.py:22-24
    return v / d


def eye_to_world(cam: dict, v: np.ndarray) -> np.ndarray:
    t = np.asarray(cam["T"], np.float64)
    return v @ t[:3, :3].T + t[:3, 3]


def world_to_eye(cam: dict, v: np.ndarray) -> np.ndarray:
    t = np.asarray(cam["T"], np.float64)
    return (v - t[:3, 3]) @ t[:3, :3]                                       # R^T (v - t), camera.py:296-300


def fisheye62_distort(k, p: np.ndarray) -> np.ndarray:
    k1, k2, k3, k4, p1, p2, k5, k6 = k
    r2 = np.clip((p * p).sum(-1, keepdims=True), -math.pi ** 2, math.pi ** 2)
    r4 = r2 * r2
    r6 = r2 * r4
    radial = 1 + k1 * r2 + k2 * r4 + k3 * r6 + k4 * (r4 * r4) + k5 * (r4 * r6) + k6 * (r6 * r6)
    uv = p * radial
    x, y = uv[..., 0], uv[..., 1]
    x2, y2, xy = x * x, y * y, x * y
    rr = x2 + y2
    xo = x + 2 * p2 * xy + p1 * (rr + 2 * x2)
    yo = y + 2 * p1 * xy + p2 * (rr + 2 * y2)
    return np.stack((xo, yo), -1)


def eye_to_window(cam: dict, v: np.ndarray) -> np.ndarray:
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    if cam.get("k") is None:                                                # perspective, camera.py:61-66
        p = np.stack((x / z, y / z), -1)
    else:                                                                   # arctan, camera.py:78-85
        r = np.sqrt(x * x + y * y)
        s = np.arctan2(r, z) / np.maximum(r, 2.0 ** -128)
        p = fisheye62_distort(cam["k"], np.stack((x * s, y * s), -1))
    return p * np.asarray(cam["f"]) + np.asarray(cam["c"])


# ----------------------------------------------------------------------------- resampler
def warp_map(src_cam: dict, dst_cam: dict) -> np.ndarray:
    """tracker.py:61-85: float32 [H,W,2] source window coordinates for every dst pixel."""
    w, h = int(dst_cam["w"]), int(dst_cam["h"])
    px, py = np.meshgrid(np.arange(w), np.arange(h))
    dst = np.column_stack((px.ravel(), py.ravel()))
    eye = world_to_eye(src_cam, eye_to_world(dst_cam, pinhole_window_to_eye(dst_cam, dst)))
    win = eye_to_window(src_cam, eye)
    win[eye[:, 2] < 0] = -1
    return win.astype(np.float32).reshape(h, w, 2)


def remap_bilinear(src: np.ndarray, m: np.ndarray, mode: str = "cv2") -> np.ndarray:
    """Bilinear sample of u8 `src` [H,W] at float32 map `m` [h,w,2]=(x,y), border constant 0.
    mode "float": exact float bilinear, float32 result in [0,255] (not rounded).
    mode "cv2": OpenCV's CV_8U INTER_LINEAR arithmetic - sx=round(x*32) (saturating int),
      integer part >>5, 5-bit fractions, weights from the 32x32 table of round-to-nearest
      15-bit fixed-point products normalised to sum 32768, result (sum + 16384) >> 15."""
    hs, ws = src.shape
    mx, my = m[..., 0].astype(np.float32), m[..., 1].astype(np.float32)
    s = src.astype(np.int64)

    def tap(ix, iy):
        ok = (ix >= 0) & (ix < ws) & (iy >= 0) & (iy < hs)
        return np.where(ok, s[np.clip(iy, 0, hs - 1), np.clip(ix, 0, ws - 1)], 0)

    if mode == "float":
        x0, y0 = np.floor(mx), np.floor(my)
        fx, fy = (mx - x0).astype(np.float32), (my - y0).astype(np.float32)
        ix, iy = x0.astype(np.int64), y0.astype(np.int64)
        one = np.float32(1)
        v = (tap(ix, iy) * ((one - fx) * (one - fy)) + tap(ix + 1, iy) * (fx * (one - fy))
             + tap(ix, iy + 1) * ((one - fx) * fy) + tap(ix + 1, iy + 1) * (fx * fy))
        return v.astype(np.float32)
    sx = np.clip(np.rint(mx.astype(np.float64) * 32), -2 ** 31, 2 ** 31 - 1).astype(np.int64)
    sy = np.clip(np.rint(my.astype(np.float64) * 32), -2 ** 31, 2 ** 31 - 1).astype(np.int64)
    ix, iy, ax, ay = sx >> 5, sy >> 5, sx & 31, sy & 31
    tab = cv2_bilinear_tab()
    wts = tab[ay, ax]                                                        # [...,4] int
    v = (tap(ix, iy) * wts[..., 0] + tap(ix + 1, iy) * wts[..., 1]
         + tap(ix, iy + 1) * wts[..., 2] + tap(ix + 1, iy + 1) * wts[..., 3])
    return ((v + (1 << 14)) >> 15).astype(np.uint8)


_TAB = None


def cv2_bilinear_tab() -> np.ndarray:
    """[32,32,4] int weights (w00,w01,w10,w11), each row summing to 32768: float32 products
    of the 1-D taps, saturate_cast<short>(v*32768) with round-half-even, then the residual is
    added to the largest weight of the 2x2 block (OpenCV imgwarp.cpp initInterTab2D)."""
    global _TAB
    if _TAB is None:
        t = np.zeros((32, 32, 4), np.int64)
        f = (np.arange(32, dtype=np.float32) / np.float32(32))
        one = np.float32(1)
        for a in range(32):          # y fraction
            for b in range(32):      # x fraction
                wf = np.array([(one - f[a]) * (one - f[b]), (one - f[a]) * f[b],
                               f[a] * (one - f[b]), f[a] * f[b]], np.float32)
                wi = np.clip(np.rint(wf * np.float32(32768)), -32768, 32767).astype(np.int64)
                diff = int(wi.sum()) - 32768
                if diff != 0:
                    # imgwarp.cpp: subtract from the max tap if the sum is too large, add to
                    # the min tap if too small (2x2 neighbourhood around the centre)
                    if diff < 0:
                        wi[int(np.argmax(wi))] -= diff
                    else:
                        wi[int(np.argmin(wi))] -= diff
                t[a, b] = wi
        _TAB = t
    return _TAB


def warp_image(src_cam: dict, dst_cam: dict, src_img: np.ndarray, mode: str = "cv2") -> np.ndarray:
    return remap_bilinear(src_img, warp_map(src_cam, dst_cam), mode)


# ----------------------------------------------------------------------------- crop cameras
def from_two_vectors(a, b) -> np.ndarray:
    def nrm(v):
        return v / np.maximum(5.43e-20, np.sqrt((v * v).sum()))
    a, b = nrm(np.asarray(a, np.float64)), nrm(np.asarray(b, np.float64))
    v = np.cross(a, b)
    s, c = np.linalg.norm(v), float(np.dot(a, b))
    k = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    return np.eye(3) + k + (k @ k) * (1 - c) / max(s * s, 1e-15)             # affine.py:34-44


def look_at(world_to_eye_xf: np.ndarray, center: np.ndarray, camera_angle_deg: float) -> np.ndarray:
    # affine.py:47-76 (scipy Rotation.from_euler("z", a, degrees=True) == Rz(a))
    c_loc = world_to_eye_xf[:3, :3] @ center + world_to_eye_xf[:3, 3]
    dr = from_two_vectors(np.array([0.0, 0.0, 1.0]), c_loc / np.linalg.norm(c_loc))
    e2w = np.linalg.inv(world_to_eye_xf)
    a = math.radians(camera_angle_deg)
    rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
    new = e2w.copy()
    new[:3, :3] = e2w[:3, :3] @ dr @ rz
    return np.linalg.inv(new)


def crop_camera_from_points(cam: dict, pts_world: np.ndarray, size, mirror_x: bool,
                            camera_angle: float, focal_multiplier: float) -> dict:
    # crop.py:31-82 + :15-28
    w2e = np.linalg.inv(np.asarray(cam["T"], np.float64))
    center = (pts_world.min(0) + pts_world.max(0)) / 2.0
    new_w2e = look_at(w2e, center, camera_angle)
    if mirror_x:
        mx = np.eye(4, dtype=np.float32)
        mx[0, 0] = -1
        new_w2e = mx @ new_w2e
    pe = pts_world @ new_w2e[:3, :3].T + new_w2e[:3, 3]
    ndc = pe[:, :2] / pe[:, 2:]
    cxy = (np.array([size[0], size[1]], pe.dtype) - 1) / 2
    fxy = cxy / np.abs(ndc).max()
    if np.any(pe[:, 2:] < 0.0001) or np.any(fxy < 5):
        raise ValueError("Unable to create crop camera", fxy)
    return {"w": int(size[0]), "h": int(size[1]), "f": tuple(focal_multiplier * fxy), "c": tuple(cxy),
            "k": None, "T": np.linalg.inv(new_w2e)}


def landmarks_from_pose(hand_model: dict, joint_angles, wrist_xf, hand_idx: int) -> np.ndarray:
    # perspective_crop.py:40-51 (float32 FK; right hand: negate column 0)
    xf = np.array(wrist_xf, np.float64, copy=True)
    if hand_idx == 1:
        xf[:, 0] *= -1
    return ref_fk.skin_landmarks(hand_model, np.asarray(joint_angles, np.float32), xf.astype(np.float32))


def rank_cameras(cams: List[dict], lm_world: np.ndarray, min_vis: int) -> List[int]:
    # perspective_crop.py:54-86 (stable sort by visible count, descending)
    counts, keep = [], []
    for i, cam in enumerate(cams):
        eye = world_to_eye(cam, lm_world.astype(np.float64))
        win = eye_to_window(cam, eye)
        n = int(((win[:, 0] >= 0) & (win[:, 0] <= cam["w"] - 1) & (win[:, 1] >= 0)
                 & (win[:, 1] <= cam["h"] - 1) & (eye[:, 2] > 0)).sum())
        counts.append(n)
        if n >= min_vis:
            keep.append(i)
    keep.sort(reverse=True, key=lambda i: counts[i])
    return keep


def gen_crop_cameras(cams: List[dict], camera_angles, hand_model: dict, joint_angles, wrist_xf,
                     hand_idx: int, size=(96, 96), max_views: int = 2, focal_multiplier: float = 0.8,
                     min_vis: int = 19) -> Dict[int, dict]:
    # perspective_crop.py:89-180 with num_crop_points=63, sort_camera_index=True (tracker.py:243-256)
    lim = np.asarray(hand_model["joint_limits"], np.float32)
    neutral = lim[:, 0] * np.float32(0.5) + lim[:, 1] * np.float32(0.5)
    pts = np.concatenate([landmarks_from_pose(hand_model, ja, wrist_xf, hand_idx)
                          for ja in (joint_angles, neutral, np.zeros(22, np.float32))], 0)
    order = sorted(rank_cameras(cams, landmarks_from_pose(hand_model, joint_angles, wrist_xf, hand_idx), min_vis))
    out: Dict[int, dict] = {}
    for ci in order:
        out[ci] = crop_camera_from_points(cams[ci], pts, size, hand_idx == 1, camera_angles[ci], focal_multiplier)
        if len(out) == max_views:
            break
    return out


def network_inputs_for_crop(crop_cam: dict):
    """tracker.py:333-337: K [3,3] and world->eye extrinsics with translation in metres."""
    k = np.array([[crop_cam["f"][0], 0, crop_cam["c"][0]], [0, crop_cam["f"][1], crop_cam["c"][1]], [0, 0, 1.0]])
    ext = np.linalg.inv(np.asarray(crop_cam["T"], np.float64))
    ext[:3, 3] *= 0.001
    return k.astype(np.float32), ext.astype(np.float32)


def camera_from_json(js: dict, cam_to_world: Optional[np.ndarray] = None) -> dict:
    # camera.py:423-444
    k = None
    if js["DistortionModel"] == "FishEye62":
        k = tuple(js[n] for n in ("k1", "k2", "k3", "k4", "p1", "p2", "k5", "k6"))
    return {"w": js["ImageSizeX"], "h": js["ImageSizeY"], "f": (js["fx"], js["fy"]), "c": (js["cx"], js["cy"]),
            "k": k, "T": np.eye(4) if cam_to_world is None else np.asarray(cam_to_world, np.float64)}
