"""Deterministic synthetic weights and inputs (no checkpoint ships with the
reference: pretrained_models/pretrained_weights.torch is listed in
/root/reference/.MISSING_LARGE_BLOBS).

A counter-based generator keyed by (state-dict key, flat index, seed) so that the
same numbers can be produced in any process without depending on torch's RNG
stream: golden generation (reference side), parity tests and bench.py all call
`synthetic_state_dict`. Distributions follow SURVEY.md section 8(d): conv weights
N(0, sqrt(2/(k*k*Cout))) as in lib/models/backbone_resnet.py:117-123, BN
gamma~U[0.5,1.5], beta~N(0,0.1), mean~N(0,0.1), var~U[0.5,1.5], linear weights and
all biases U(+-1/sqrt(fan_in)).
"""
import math
from typing import Dict

import numpy as np

from . import arch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _mix(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def counter_uniform(key: str, n: int, seed: int = 0, stream: int = 0) -> np.ndarray:
    """n doubles in [0,1), element i depends only on (key, i, seed, stream)."""
    base = np.uint64((_fnv1a(key) + 0x9E3779B97F4A7C15 * (seed * 4 + stream + 1)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + base
    bits = _mix(_mix(ctr))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def counter_normal(key: str, n: int, seed: int = 0) -> np.ndarray:
    u1 = counter_uniform(key, n, seed, 0)
    u2 = counter_uniform(key, n, seed, 1)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)


_PROJ_GAIN = 1.0
_BIAS_AMP = 0.02     # BN beta/mean and conv biases: small, so activations stay input-driven


def _output_row_gain(cout: int) -> np.ndarray:
    """Per-output gain of the regressor's last conv: finger angles vary by a few 0.1 rad from
    sample to sample, wrist points by ~1-2 cm around their rigid pattern, log-scale by ~0.05."""
    g = np.empty(cout)
    g[0:20] = 10.0
    g[20:41] = 0.7
    if cout == 63:
        g[41] = 2.0
        g[42:] = 10.0
    else:
        g[41:] = 10.0
    return g



def _output_bias(key: str, n: int, seed: int) -> np.ndarray:
    """Bias of the regressor's last 1x1 conv, chosen so that outputs sit in the regime a trained
    network produces: finger angles in [-0.2,1.2] rad, the 7 wrist points close to a rigid
    image (rotation R0, translation ~0.35 m in front of cam0) of the fixed source points of
    lib/models/regressor.py:19-47, log skeleton scale ~0, sigma logits ~U(-1,1)."""
    u = counter_uniform(key, n, seed)
    b = np.zeros(n)
    b[0:20] = -0.2 + 1.4 * u[0:20]
    src = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1],
                    [-1, -1, 0], [-1, 0, -1], [0, -1, -1]], np.float64)
    nrm = np.linalg.norm(src, axis=1, keepdims=True)
    src = np.where(nrm > 0, src / np.maximum(nrm, 1e-30) * 0.1, src)
    v = np.array([0.4, -0.7, 0.5])
    th = np.linalg.norm(v)
    k = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    r0 = np.eye(3) + np.sin(th) / th * k + (1 - np.cos(th)) / th ** 2 * (k @ k)
    b[20:41] = (src @ r0.T + np.array([0.03, -0.05, 0.35])).reshape(-1)
    if n == 63:
        b[41] = 0.05
        b[42:] = u[42:] * 2 - 1
    else:
        b[41:] = u[41:] * 2 - 1
    return b


def synthetic_state_dict(seed: int = 0) -> Dict[str, np.ndarray]:
    """Full 252-entry state dict as numpy arrays (float32; int64 for counters)."""
    sd: Dict[str, np.ndarray] = {}
    for key, shape, kind in arch.state_dict_spec():
        n = int(np.prod(shape)) if len(shape) else 1
        if kind == "conv_w":
            cout, cin, kh, kw = shape
            if kh == 1 and "downsample" not in key:
                # biased 1x1 convs (projection, fusion, temporal, regressor output): fan-in
                # scaling so that the head keeps activations O(1)
                v = counter_normal(key, n, seed) * math.sqrt(1.0 / cin)
                if key.endswith("_pose_regression_layers.2.weight"):
                    v = (v.reshape(cout, cin) * _output_row_gain(cout)[:, None]).reshape(-1)
                elif key.endswith("_image_backbone.1.weight"):
                    v = v * _PROJ_GAIN
            else:
                v = counter_normal(key, n, seed) * math.sqrt(2.0 / (kh * kw * cout))
        elif kind == "lin_w":
            v = (counter_uniform(key, n, seed) * 2 - 1) / math.sqrt(shape[1])
        elif kind in ("conv_b", "lin_b"):
            # fan_in is not recoverable from the bias shape; a fixed small range keeps
            # activations O(1)
            v = (counter_uniform(key, n, seed) * 2 - 1) * _BIAS_AMP
            if key.endswith("_pose_regression_layers.2.bias"):
                v = _output_bias(key, n, seed)
        elif kind == "bn_w":
            v = 0.5 + counter_uniform(key, n, seed)
            if key.endswith(".bn2.weight"):
                v = v * 0.3     # damp the residual branch: without trained statistics the
                                # variance would otherwise double in each of the 12+4 blocks
        elif kind in ("bn_b", "bn_mean"):
            v = counter_normal(key, n, seed) * _BIAS_AMP
        elif kind == "bn_var":
            v = 0.5 + counter_uniform(key, n, seed)
        elif kind == "bn_nbt":
            sd[key] = np.array(1, dtype=np.int64)
            continue
        else:
            raise KeyError(kind)
        sd[key] = v.astype(np.float32).reshape(shape)
    return sd


def synthetic_crops(n: int, seed: int = 0) -> np.ndarray:
    """[n,96,96] float32 in {0..255}/255 - the value set the reference's
    `crop.astype(float32)/255` (lib/tracker/tracker.py:332) can produce.  Each crop is a
    different low-frequency pattern (random blobs + oriented waves) plus 20% pixel noise, so
    that features differ from crop to crop at every depth of the network."""
    c = arch.CROP
    yy, xx = np.mgrid[0:c, 0:c].astype(np.float64) / c
    par = counter_uniform("crops.par", n * 16, seed).reshape(n, 16)
    noise = counter_uniform("crops.noise", n * c * c, seed).reshape(n, c, c)
    out = np.empty((n, c, c), np.float32)
    for i in range(n):
        p = par[i]
        img = 0.35 + 0.3 * (p[0] - 0.5)
        for b in range(3):      # gaussian blobs of random position / width / sign
            cx, cy, sg, am = p[1 + 4 * b], p[2 + 4 * b], 0.08 + 0.25 * p[3 + 4 * b], p[4 + 4 * b] - 0.35
            img = img + am * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sg * sg))
        ang, fr, ph = 6.28 * p[13], 2 + 10 * p[14], 6.28 * p[15]
        img = img + 0.15 * np.sin(6.28 * fr * (xx * np.cos(ang) + yy * np.sin(ang)) + ph)
        img = 0.8 * img + 0.2 * noise[i]
        out[i] = np.clip(np.floor(img * 256.0), 0, 255).astype(np.float32) / np.float32(255.0)
    return out


def synthetic_frames(f: int, n_cams: int = 4, h: int = 480, w: int = 636, seed: int = 0) -> np.ndarray:
    """[f,n_cams,h,w] uint8 frames: smooth low-frequency pattern + noise, so that the
    bilinear resampler sees both gradients and high-frequency content."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.empty((f, n_cams, h, w), dtype=np.uint8)
    ph = counter_uniform("frames.phase", f * n_cams * 4, seed).reshape(f, n_cams, 4)
    for i in range(f):
        noise = counter_uniform(f"frames.noise.{i}", n_cams * h * w, seed).reshape(n_cams, h, w)
        for c in range(n_cams):
            a, b, p, q = ph[i, c]
            smooth = 0.5 + 0.25 * np.sin(xx * (0.01 + 0.04 * a) + 6.28 * p) \
                + 0.25 * np.cos(yy * (0.01 + 0.04 * b) + 6.28 * q)
            img = 0.7 * smooth + 0.3 * noise[c]
            out[i, c] = np.clip(np.floor(img * 256.0), 0, 255).astype(np.uint8)
    return out


def channel_rescaled_state_dict(sd: Dict[str, np.ndarray], spread: int, seed: int = 0, inner: bool = True,
                                trunk: bool = True) -> Dict[str, np.ndarray]:
    """The same fp32 function as `sd`, written with other per-channel scales inside the backbone: every inner channel c of
    every BasicBlock (bn1 output / conv2 input) is divided by 2^k_c - bn1.{weight,bias}[c] x 2^-k_c, conv2.weight[:, c] x 2^k_c
    - and every trunk channel of a layer (stem / bn2 / downsample-bn rows of channel c across the layer's blocks; conv1 columns
    of the layer's later blocks and of whatever reads the layer: the next layer's first conv1 and shortcut, the projection) by
    2^t_c, with k_c, t_c independent random integers in [-spread, spread].  Powers of two commute with ReLU and every fp32
    rounding (lib/models/backbone_resnet.py:56-72), so an exact-fp32 implementation returns the same bits for both; a near-dead
    BatchNorm channel whose consumer weights compensate is the trained-checkpoint case this stands for."""
    out = dict(sd)
    pre = "_feature_extractor._image_backbone.0._layers."
    nb = [2, 3, 5, 2]
    planes = [32, 64, 128, 256]

    def ints(tag, n):
        u = counter_uniform("rescale." + tag, n, seed)
        return np.floor(u * (2 * spread + 1)).astype(np.int64) - spread

    def mul(key, factor, axis):
        v = out[key]
        shape = [1] * v.ndim
        shape[axis] = -1
        out[key] = (v * factor.reshape(shape)).astype(np.float32)

    for li in range(4):
        l = li + 1
        ch = planes[li]
        if trunk:
            t = ints(f"trunk.{l}", ch)
            down, up = np.exp2(-t).astype(np.float32), np.exp2(t).astype(np.float32)
            if l == 1:
                mul(pre + "0.1.weight", down, 0)
                mul(pre + "0.1.bias", down, 0)
            else:
                mul(pre + f"{l}.0.downsample.1.weight", down, 0)
                mul(pre + f"{l}.0.downsample.1.bias", down, 0)
            for b in range(nb[li]):
                mul(pre + f"{l}.{b}.bn2.weight", down, 0)
                mul(pre + f"{l}.{b}.bn2.bias", down, 0)
                if b > 0 or l == 1:
                    mul(pre + f"{l}.{b}.conv1.weight", up, 1)
            if l < 4:
                mul(pre + f"{l + 1}.0.conv1.weight", up, 1)
                mul(pre + f"{l + 1}.0.downsample.0.weight", up, 1)
            else:
                mul("_feature_extractor._image_backbone.1.weight", up, 1)
        if inner:
            for b in range(nb[li]):
                k = ints(f"inner.{l}.{b}", ch)
                down, up = np.exp2(-k).astype(np.float32), np.exp2(k).astype(np.float32)
                mul(pre + f"{l}.{b}.bn1.weight", down, 0)
                mul(pre + f"{l}.{b}.bn1.bias", down, 0)
                mul(pre + f"{l}.{b}.conv2.weight", up, 1)
    return out
