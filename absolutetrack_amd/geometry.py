"""Host-side camera geometry with the reference's call surface (numpy, float64).

Mirrors the public interface of lib/common/camera.py (CameraModel and its two concrete
models, read_camera_from_json), lib/common/affine.py and lib/common/crop.py so that the
reference's scripts keep working through the `lib.common.*` shims.  This is per-frame
parameter plumbing (a few dozen flops per camera); the per-pixel work - the 96x96
coordinate map and the resampling - runs in csrc/warp.hip from the packed parameter rows
produced by `pack_source_camera` / `pack_crop_camera`.
"""
import json
import math
from typing import Sequence, Tuple

import numpy as np

_NORM_EPS = 5.43e-20          # lib/common/affine.py:22
_ATAN_EPS = 2.0 ** -128       # lib/common/camera.py:80


# ----------------------------------------------------------------------------- affine helpers
def transform_vec3(m: np.ndarray, v: np.ndarray) -> np.ndarray:
    """Rotate points by the 3x3 block of m (lib/common/affine.py:15-19)."""
    if m.ndim == 2:
        return (v.reshape(-1, 3) @ m[:3, :3].T).reshape(v.shape)
    return np.einsum("...ij,...j->...i", m[..., :3, :3], v)


def transform3(m: np.ndarray, v: np.ndarray) -> np.ndarray:
    return transform_vec3(m, v) + m[..., :3, 3]


def normalized(v: np.ndarray, axis: int = -1, eps: float = _NORM_EPS) -> np.ndarray:
    return v / np.maximum(eps, np.sqrt(np.sum(v * v, axis=axis, keepdims=True)))


def skew_matrix(v: np.ndarray) -> np.ndarray:
    x, y, z = v
    return np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]], dtype=v.dtype)


def from_two_vectors(a_orig: np.ndarray, b_orig: np.ndarray) -> np.ndarray:
    """Smallest rotation taking direction a to direction b (lib/common/affine.py:34-44)."""
    a, b = normalized(a_orig), normalized(b_orig)
    axis = np.cross(a, b)
    sin_ab = np.linalg.norm(axis)
    k = skew_matrix(axis)
    return np.eye(3) + k + (k @ k) * (1 - np.dot(a, b)) / max(sin_ab * sin_ab, 1e-15)


def make_look_at_matrix(orig_world_to_eye: np.ndarray, center: np.ndarray, camera_angle: float = 0) -> np.ndarray:
    """Re-aim a camera at `center` keeping its position, then roll it by camera_angle degrees
    about the new optical axis (lib/common/affine.py:47-76)."""
    c_eye = transform3(orig_world_to_eye, center)
    aim = from_two_vectors(np.array([0, 0, 1], dtype=center.dtype), c_eye / np.linalg.norm(c_eye))
    ang = math.radians(camera_angle)
    roll = np.array([[math.cos(ang), -math.sin(ang), 0.0], [math.sin(ang), math.cos(ang), 0.0], [0.0, 0.0, 1.0]])
    eye_to_world = np.linalg.inv(orig_world_to_eye)
    eye_to_world[:3, :3] = eye_to_world[:3, :3] @ aim @ roll
    return np.linalg.inv(eye_to_world)


# ----------------------------------------------------------------------------- distortion
class NoDistortion(tuple):
    _fields = ()

    def __new__(cls):
        return super().__new__(cls, ())

    def evaluate(self, p):
        return p

    def undistort(self, q):
        return q


class Fisheye62Distortion(tuple):
    """6 radial + 2 tangential coefficients (k1,k2,k3,k4,p1,p2,k5,k6), lib/common/camera.py:105-143."""
    _fields = ("k1", "k2", "k3", "k4", "p1", "p2", "k5", "k6")

    def __new__(cls, *coeffs):
        if len(coeffs) != 8:
            raise TypeError("Fisheye62 takes 8 coefficients")
        return super().__new__(cls, tuple(float(c) for c in coeffs))

    def evaluate(self, p: np.ndarray) -> np.ndarray:
        k1, k2, k3, k4, p1, p2, k5, k6 = self
        rr = np.clip(np.sum(p * p, axis=-1, keepdims=True), -math.pi ** 2, math.pi ** 2)
        r4, r6 = rr * rr, rr * rr * rr
        uv = p * (1 + k1 * rr + k2 * r4 + k3 * r6 + k4 * (r4 * r4) + k5 * (r4 * r6) + k6 * (r6 * r6))
        x, y = uv[..., 0], uv[..., 1]
        xx, yy, xy = x * x, y * y, x * y
        s = xx + yy
        return np.stack((x + 2 * p2 * xy + p1 * (s + 2 * xx), y + 2 * p1 * xy + p2 * (s + 2 * yy)), axis=-1)

    def undistort(self, q: np.ndarray, solver_iters: int = 5) -> np.ndarray:
        """Fixed-point inverse of the radial part only (lib/common/camera.py:146-181)."""
        k1, k2, k3, k4, _p1, _p2, k5, k6 = self
        x, y = q[..., 0].copy(), q[..., 1].copy()
        for _ in range(5):
            r2 = x ** 2 + y ** 2
            rad = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3 + k4 * r2 ** 4 + k5 * r2 ** 5 + k6 * r2 ** 6
            x, y = q[..., 0] / rad, q[..., 1] / rad
        return np.stack([x, y], axis=-1)


# ----------------------------------------------------------------------------- camera models
class CameraModel:
    """width/height, focal f=(fx,fy), centre c=(cx,cy), distortion, camera_to_world_xf (4x4).
    Same attributes and methods as lib/common/camera.py:204-372."""

    distortion_model = NoDistortion

    def __init__(self, width, height, f, c, distort_coeffs, camera_to_world_xf=None):
        self.width, self.height = width, height
        self.f = tuple(np.broadcast_to(f, 2))
        self.c = tuple(c)
        self.camera_to_world_xf = np.eye(4) if camera_to_world_xf is None else camera_to_world_xf
        if hasattr(distort_coeffs, "evaluate"):
            self.distort = distort_coeffs
        else:
            self.distort = self.distortion_model(*distort_coeffs)

    def __repr__(self):
        return f"{type(self).__name__}({self.width}x{self.height}, f={self.f} c={self.c}"

    # -- projections are supplied by the subclasses
    @staticmethod
    def project(v):
        raise NotImplementedError

    @staticmethod
    def unproject(p):
        raise NotImplementedError

    def copy(self, camera_to_world_xf=None):
        return self.crop(0, 0, self.width, self.height, camera_to_world_xf=camera_to_world_xf)

    def world_to_eye(self, v):
        t = self.camera_to_world_xf
        return transform_vec3(t.T, v - t[:3, 3])

    def eye_to_world(self, v):
        return transform3(self.camera_to_world_xf, v)

    def eye_to_window(self, v):
        return self.distort.evaluate(self.project(v)) * self.f + self.c

    def eye_to_window_undistorted(self, v):
        return self.project(v) * self.f + self.c

    def window_to_eye(self, w):
        return self.unproject(self.distort.undistort((np.asarray(w) - self.c) / self.f))

    def crop(self, src_x, src_y, target_width, target_height, scale=1, camera_to_world_xf=None):
        return type(self)(target_width, target_height, np.asarray(self.f) * scale,
                          (np.array(self.c) - (src_x, src_y) + 0.5) * scale - 0.5, self.distort,
                          self.camera_to_world_xf if camera_to_world_xf is None else camera_to_world_xf)


class PinholePlaneCameraModel(CameraModel):
    distortion_model = NoDistortion

    @staticmethod
    def project(v):
        return v[..., :2] / v[..., 2, None]

    @staticmethod
    def unproject(p):
        ray = np.concatenate([p, np.ones(p.shape[:-1] + (1,), dtype=p.dtype)], axis=-1)
        return normalized(ray, axis=-1)

    def uv_to_window_matrix(self):
        return np.array([[self.f[0], 0, self.c[0]], [0, self.f[1], self.c[1]], [0, 0, 1]])


class Fisheye62CameraModel(CameraModel):
    distortion_model = Fisheye62Distortion

    @staticmethod
    def project(p, eps: float = _ATAN_EPS):
        x, y, z = p[..., 0], p[..., 1], p[..., 2]
        r = np.sqrt(x * x + y * y)
        s = np.arctan2(r, z) / np.maximum(r, eps)
        return np.stack((x * s, y * s), axis=-1)

    @staticmethod
    def unproject(uv):
        u, v = uv[..., 0], uv[..., 1]
        r = np.sqrt(u * u + v * v)
        s = np.sinc(r / np.pi)
        return np.stack([u * s, v * s, np.cos(r)], axis=-1)


def read_camera_from_json(js):
    if isinstance(js, str):
        js = json.loads(js)
    js = js.get("Camera", js)
    cls = {"PinholePlane": PinholePlaneCameraModel, "FishEye62": Fisheye62CameraModel}[js["DistortionModel"]]
    coeffs = [js[name] for name in cls.distortion_model._fields]
    return cls(js["ImageSizeX"], js["ImageSizeY"], (js["fx"], js["fy"]), (js["cx"], js["cy"]), coeffs)


# ----------------------------------------------------------------------------- crop cameras
def gen_intrinsics_from_bounding_pts(pts_eye: np.ndarray, image_w: int, image_h: int, min_focal: float = 5
                                     ) -> Tuple[np.ndarray, np.ndarray]:
    """Largest focal that keeps every point inside the image (lib/common/crop.py:15-28)."""
    ndc = pts_eye[..., 0:2] / pts_eye[..., 2:]
    cx_cy = (np.array([image_w, image_h], dtype=pts_eye.dtype) - 1) / 2
    fx_fy = cx_cy / np.absolute(ndc).max()
    if np.any(pts_eye[..., 2:] < 0.0001) or np.any(fx_fy < min_focal):
        raise ValueError("Unable to create crop camera", fx_fy)
    return fx_fy, cx_cy


def gen_crop_parameters_from_points(camera_orig: CameraModel, pts_world, new_image_size: Tuple[int, int],
                                    mirror_img_x: bool, camera_angle: float = 0, focal_multiplier: float = 0.95
                                    ) -> PinholePlaneCameraModel:
    """lib/common/crop.py:31-82."""
    world_to_eye = np.linalg.inv(camera_orig.camera_to_world_xf)
    center = (pts_world.min(axis=0) + pts_world.max(axis=0)) / 2.0
    new_w2e = make_look_at_matrix(world_to_eye, center, camera_angle)
    if mirror_img_x:
        flip = np.eye(4, dtype=np.float32)
        flip[0, 0] = -1
        new_w2e = flip @ new_w2e
    fx_fy, cx_cy = gen_intrinsics_from_bounding_pts(transform3(new_w2e, pts_world), new_image_size[0], new_image_size[1])
    return PinholePlaneCameraModel(width=new_image_size[0], height=new_image_size[1], f=focal_multiplier * fx_fy,
                                   c=cx_cy, distort_coeffs=[], camera_to_world_xf=np.linalg.inv(new_w2e))


# ----------------------------------------------------------------------------- native parameter rows
def pack_source_camera(f: Sequence[float], c: Sequence[float], k, cam_to_world: np.ndarray) -> np.ndarray:
    """[32] f64 row of ut_warp_crops' cam_params: fx fy cx cy | k1 k2 k3 k4 p1 p2 k5 k6 | R(9) t(3) | pad."""
    row = np.zeros(32, np.float64)
    row[0:2], row[2:4] = f, c
    if k is None or len(k) == 0:
        raise ValueError("the resampler's source camera must be a Fisheye62 model")
    row[4:12] = k
    t = np.asarray(cam_to_world, np.float64)
    row[12:21] = t[:3, :3].reshape(-1)
    row[21:24] = t[:3, 3]
    return row


def pack_crop_camera(f: Sequence[float], c: Sequence[float], cam_to_world: np.ndarray) -> np.ndarray:
    """[24] f64 row of ut_warp_crops' crop_params: fx fy cx cy | R(9) t(3) | pad."""
    row = np.zeros(24, np.float64)
    row[0:2], row[2:4] = f, c
    t = np.asarray(cam_to_world, np.float64)
    row[4:13] = t[:3, :3].reshape(-1)
    row[13:16] = t[:3, 3]
    return row


def pack_camera_model(cam: CameraModel) -> np.ndarray:
    if isinstance(cam, PinholePlaneCameraModel):
        return pack_crop_camera(cam.f, cam.c, cam.camera_to_world_xf)
    return pack_source_camera(cam.f, cam.c, tuple(cam.distort), cam.camera_to_world_xf)
