"""Evaluation metrics and the results format of the eval scripts.

Mirrors lib/common/metric_utils.py:18-112 (PCK curve, normalised AUC - host numpy, tiny), load_eval.py:19-89
(per-frame keypoint error and accelerations - one GPU launch, ut_keypoint_metrics - and the directory aggregate)
and the result files of run_eval_known_skeleton.py:96-104 / run_eval_unknown_skeleton.py (a pickled dict of three
arrays named *.npy), so load_eval.py can consume this package's outputs unchanged.
"""
import fnmatch
import io
import os
import pickle
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from . import _native, bundles

MAX_LANDMARK_ERROR_MM = 50
PCK_THRESHOLDS = np.linspace(0, MAX_LANDMARK_ERROR_MM, 101)


# ----------------------------------------------------------------------------- lib/common/metric_utils.py
def _safe_div(x, y, eps: float = 1e-6, default_val: int = 0):
    assert x.shape == y.shape
    if np.isscalar(x):
        return default_val if y < eps else x / y
    with np.errstate(divide="ignore", invalid="ignore"):
        z = x / y
    z[y < eps] = default_val
    return z


def _PCK_curve(errors: np.ndarray, mask: np.ndarray, thresholds: np.ndarray) -> np.ndarray:
    pcks = [_safe_div(((errors <= t) * mask).sum(axis=-1), mask.sum(axis=-1)) for t in thresholds]
    return np.stack(pcks).T


def PCK_curve(errors: np.ndarray, thresholds: np.ndarray, mask: Optional[np.ndarray] = None,
              axis: Optional[int] = None) -> np.ndarray:
    """Fraction of errors <= each threshold; with `axis`, one curve per element along it."""
    if mask is None:
        mask = np.ones_like(errors)
    if axis is None:
        return _PCK_curve(errors.reshape(-1), mask.reshape(-1), thresholds)
    n = errors.shape[axis]
    return _PCK_curve(np.moveaxis(errors, axis, 0).reshape(n, -1), np.moveaxis(mask, axis, 0).reshape(n, -1), thresholds)


def normalized_AUC(x: np.ndarray, y: np.ndarray, y_max: float = 1.0) -> np.ndarray:
    """Trapezoid area under curves sharing the x axis, divided by the (x range) x y_max rectangle."""
    out_shape = y.shape[:-1]
    y = y.reshape(-1, y.shape[-1])
    auc = ((x[1:] - x[:-1]).reshape(1, -1) * ((y[..., 1:] + y[..., :-1]) * 0.5)).sum(axis=-1)
    return (auc / ((x[-1] - x[0]) * y_max)).reshape(out_shape)


# ----------------------------------------------------------------------------- load_eval.py
@dataclass
class Metrics:
    keypoint_errors: np.ndarray
    keypoint_accelerations: np.ndarray
    gt_keypoint_accelerations: np.ndarray


def _compute_metrics(gt_keypoints, tracked_keypoints, valid_tracking, device=None) -> Metrics:
    """gt / tracked [hands, frames, 21, 3] (numpy or tensors), valid [hands, frames] bool (load_eval.py:26-45)."""
    if not torch.cuda.is_available():
        raise _native.NativeLibraryError("keypoint metrics run on a HIP device; there is no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    t = lambda a, dt: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(dev, dt)
    valid = t(valid_tracking, torch.bool)
    m = _native.keypoint_metrics(t(gt_keypoints, torch.float32), t(tracked_keypoints, torch.float32), valid)
    va = m["valid_acc"]
    return Metrics(keypoint_errors=m["err"][valid].cpu().numpy(), keypoint_accelerations=m["acc"][va].cpu().numpy(),
                   gt_keypoint_accelerations=m["gt_acc"][va].cpu().numpy())


# ----------------------------------------------------------------------------- result files
def save_eval_results(output_path: str, tracked_keypoints: np.ndarray, gt_keypoints: np.ndarray,
                      valid_tracking: np.ndarray) -> None:
    """The `.npy`-named pickle the eval scripts write (run_eval_known_skeleton.py:94-104)."""
    d = os.path.dirname(output_path)
    if d and not os.path.exists(d):
        os.makedirs(d)
    with io.open(output_path, "wb") as fp:
        pickle.dump({"tracked_keypoints": np.asarray(tracked_keypoints), "gt_keypoints": np.asarray(gt_keypoints),
                     "valid_tracking": np.asarray(valid_tracking)}, fp)


class _ArraysOnlyUnpickler(pickle.Unpickler):
    """Rebuilds numpy arrays and nothing else: any other global in the stream is refused."""
    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy", "ndarray"), ("numpy", "dtype")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to load {module}.{name}: result files hold plain numpy arrays only")


def load_eval_results(path: str) -> Dict[str, np.ndarray]:
    with io.open(path, "rb") as fp:
        data = _ArraysOnlyUnpickler(fp).load()
    if not isinstance(data, dict) or not {"tracked_keypoints", "gt_keypoints", "valid_tracking"} <= set(data):
        raise ValueError(f"{path} is not an eval result file")
    return data


def aggregate_metrics(output_dir: str, verbose: bool = True) -> Optional[Dict[str, float]]:
    """load_eval.py:48-89 over every *.npy result under output_dir; returns what the reference prints."""
    valid_all, metrics_all = [], []
    for cur_dir, _, filenames in os.walk(output_dir):
        for fname in sorted(fnmatch.filter(filenames, "*.npy")):
            data = load_eval_results(os.path.join(cur_dir, fname))
            valid_all.append(data["valid_tracking"])
            metrics_all.append(_compute_metrics(data["gt_keypoints"], data["tracked_keypoints"], data["valid_tracking"]))
    if not metrics_all:
        return None
    combined = bundles.group(metrics_all, np.concatenate)
    pck = PCK_curve(combined.keypoint_errors, PCK_THRESHOLDS) * 100.0
    valid_cat = np.concatenate(valid_all, axis=1)
    out = {"n_total": int(valid_cat.size), "n_valid": int(valid_cat.sum()),
           "success_rate_percent": float(valid_cat.sum() / valid_cat.size * 100),
           "mean_keypoint_error": float(combined.keypoint_errors.mean()),
           "auc_score": float(normalized_AUC(PCK_THRESHOLDS, pck)),
           "mean_keypoint_acceleration": float(combined.keypoint_accelerations.mean()),
           "gt_mean_keypoint_acceleration": float(combined.gt_keypoint_accelerations.mean())}
    if verbose:
        print(f"  Tracked {out['n_valid']} out of {out['n_total']}, success rate: {out['success_rate_percent']}%")
        print(f"  Mean keypoint error: {out['mean_keypoint_error']}")
        print(f"  AUC score: {out['auc_score']}")
        print(f"  Mean keypoint accelerations: {out['mean_keypoint_acceleration']}")
        print(f"  GT mean keypoint accelerations: {out['gt_mean_keypoint_acceleration']}")
    return out
