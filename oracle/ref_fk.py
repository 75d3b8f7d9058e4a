"""Oracle (TEST INFRASTRUCTURE ONLY): forward kinematics + linear blend skinning of
the 21 hand landmarks, numpy float32.

Follows lib/common/hand_skinning.py:17-209 (reference) and, for the rotation
exponential it calls at :46, the published algorithm of pytorch3d
`pytorch3d.transforms.so3_exp_map` (dependency pinned only as "@stable",
README.md:10; NOT vendored under /root/reference):
    theta = sqrt(clamp(|v|^2, min=eps)), eps = 1e-4
    R = I + sin(theta)/theta * K + (1-cos(theta))/theta^2 * K^2,  K = hat(v)
Pinned by sample_data/user05/recording_*.npy['gt_keypoints'] (tests/golden/fk_user05.npz).
"""
import numpy as np

N_FRAMES = 17  # root + wrist + 3 per finger (lib/common/hand.py:20)


def so3_exp(v: np.ndarray, eps: float = 1e-4) -> np.ndarray:
    v = v.astype(np.float32)
    n2 = (v * v).sum(-1)
    theta = np.sqrt(np.maximum(n2, np.float32(eps)))
    inv = np.float32(1.0) / theta
    f1 = inv * np.sin(theta)
    f2 = inv * inv * (np.float32(1.0) - np.cos(theta))
    k = np.zeros(v.shape[:-1] + (3, 3), np.float32)
    k[..., 0, 1], k[..., 0, 2] = -v[..., 2], v[..., 1]
    k[..., 1, 0], k[..., 1, 2] = v[..., 2], -v[..., 0]
    k[..., 2, 0], k[..., 2, 1] = -v[..., 1], v[..., 0]
    return f1[..., None, None] * k + f2[..., None, None] * (k @ k) + np.eye(3, dtype=np.float32)


def joint_local_xf(axes, rest, angles) -> np.ndarray:
    # hand_skinning.py:35-53: rotation about the joint's rest position
    r = so3_exp(axes * angles[..., None])
    xf = np.zeros(angles.shape + (4, 4), np.float32)
    xf[..., :3, :3] = r
    xf[..., :3, 3] = rest - (r @ rest[..., None])[..., 0]
    xf[..., 3, 3] = 1
    return xf


def skinning_frames(axes, rest, angles, wrist) -> np.ndarray:
    # hand_skinning.py:17-32,100-127: [B,17,4,4]; per finger keep products after 2,3,4 joints
    loc = joint_local_xf(axes[:, :20], rest[:, :20], angles[:, :20])
    frames = [wrist, wrist]
    for f in range(5):
        t = wrist
        for j in range(4):
            t = t @ loc[:, 4 * f + j]
            if j >= 1:
                frames.append(t)
    return np.stack(frames, 1)


def skin_landmarks(hand_model: dict, joint_angles: np.ndarray, wrist_xf: np.ndarray) -> np.ndarray:
    """hand_model: dict of numpy arrays with the reference HandModel field names; leading
    dims of joint_angles/wrist_xf are arbitrary; model fields either unbatched or with the
    same leading dims (hand_skinning.py:154-209)."""
    lead = joint_angles.shape[:-1]
    b = int(np.prod(lead)) if lead else 1
    ja = joint_angles.reshape(b, -1).astype(np.float32)
    xf = wrist_xf.reshape(b, 4, 4).astype(np.float32)

    def bc(a, tail):
        a = np.asarray(a)
        return np.broadcast_to(a.reshape((-1,) + tail) if a.ndim > len(tail) else a[None], (b,) + tail)

    axes = bc(hand_model["joint_rotation_axes"], (22, 3)).astype(np.float32)
    rest = bc(hand_model["joint_rest_positions"], (22, 3)).astype(np.float32)
    lm = bc(hand_model["landmark_rest_positions"], (21, 3)).astype(np.float32)
    w = bc(hand_model["landmark_rest_bone_weights"], (21, 3)).astype(np.float32)
    idx = bc(hand_model["landmark_rest_bone_indices"], (21, 3)).astype(np.int64)
    frames = skinning_frames(axes, rest, ja, xf)                          # [b,17,4,4]
    # hand_skinning.py:70-97: dense [b,21,17] weights, later non-zero entries overwrite
    dense = np.zeros((b, 21, N_FRAMES), np.float32)
    for k in range(3):
        nz = w[..., k] != 0
        bi, li = np.nonzero(nz)
        dense[bi, li, idx[bi, li, k]] = w[bi, li, k]
    homo = np.concatenate([lm, np.ones((b, 21, 1), np.float32)], -1)       # [b,21,4]
    per_frame = np.einsum("bfij,blj->blfi", frames, homo)                  # [b,21,17,4]
    out = (per_frame * dense[..., None]).sum(2)[..., :3]
    return out.reshape(lead + (21, 3))
