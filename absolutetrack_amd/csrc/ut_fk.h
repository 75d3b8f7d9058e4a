// Device-side forward kinematics + linear blend skinning shared by fk.hip and cropgen.hip.
// See fk.hip for the reference citations (lib/common/hand_skinning.py:17-209, pytorch3d so3_exp_map).
#pragma once
#include <hip/hip_runtime.h>

namespace ut {

struct M34 { float m[12]; };   // rows 0..2 of a 4x4 rigid/affine transform, row major

__device__ inline M34 mul34(const M34& a, const M34& b) {
  M34 c;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = a.m[4 * i] * b.m[j];
      s = fmaf(a.m[4 * i + 1], b.m[4 + j], s);
      s = fmaf(a.m[4 * i + 2], b.m[8 + j], s);
      if (j == 3) s += a.m[4 * i + 3];
      c.m[4 * i + j] = s;
    }
  }
  return c;
}

__device__ inline M34 joint_local(const float* axis, const float* rest, float angle) {
  const float vx = axis[0] * angle, vy = axis[1] * angle, vz = axis[2] * angle;
  const float n2 = vx * vx + vy * vy + vz * vz;
  const float th = sqrtf(fmaxf(n2, 1e-4f));
  const float inv = 1.0f / th;
  const float f1 = inv * sinf(th);
  const float f2 = inv * inv * (1.0f - cosf(th));
  // K = hat(v); K^2 has -(vy^2+vz^2) etc. on the diagonal and vi*vj off it
  float r[9];
  r[0] = 1.0f - f2 * (vy * vy + vz * vz);
  r[1] = -f1 * vz + f2 * (vx * vy);
  r[2] = f1 * vy + f2 * (vx * vz);
  r[3] = f1 * vz + f2 * (vx * vy);
  r[4] = 1.0f - f2 * (vx * vx + vz * vz);
  r[5] = -f1 * vx + f2 * (vy * vz);
  r[6] = -f1 * vy + f2 * (vx * vz);
  r[7] = f1 * vx + f2 * (vy * vz);
  r[8] = 1.0f - f2 * (vx * vx + vy * vy);
  M34 l;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    l.m[4 * i] = r[3 * i]; l.m[4 * i + 1] = r[3 * i + 1]; l.m[4 * i + 2] = r[3 * i + 2];
    l.m[4 * i + 3] = rest[i] - (r[3 * i] * rest[0] + r[3 * i + 1] * rest[1] + r[3 * i + 2] * rest[2]);
  }
  return l;
}


// 21 landmarks [63] of one pose.  hm: packed hand model (321 floats, see include/umetrack_hip.h); a: 22 joint
// angles; wrist: root-to-world transform.
__device__ inline void skin_landmarks_dev(const float* __restrict__ hm, const float* __restrict__ a, const M34& wrist,
                                          float* __restrict__ o) {
  const float* axes = hm;
  const float* rest = hm + 66;
  const float* lm = hm + 132;
  const float* wts = hm + 195;
  const float* idx = hm + 258;
  M34 frames[17];
  frames[0] = wrist;
  frames[1] = wrist;
#pragma unroll
  for (int f = 0; f < 5; ++f) {
    M34 t = wrist;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = 4 * f + j;
      t = mul34(t, joint_local(axes + 3 * q, rest + 3 * q, a[q]));
      if (j >= 1) frames[2 + 3 * f + (j - 1)] = t;
    }
  }
  for (int l = 0; l < 21; ++l) {
    const float px = lm[3 * l], py = lm[3 * l + 1], pz = lm[3 * l + 2];
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int f = 0; f < 17; ++f) {
      // dense skinning weight of frame f: the last non-zero entry naming it wins
      float w = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (wts[3 * l + k] != 0.f && (int)idx[3 * l + k] == f) w = wts[3 * l + k];
      if (w != 0.f) {
        const M34& t = frames[f];
        const float qx = px * w, qy = py * w, qz = pz * w;   // (p,1) * w, as the reference scales first
        ax += t.m[0] * qx + t.m[1] * qy + t.m[2] * qz + t.m[3] * w;
        ay += t.m[4] * qx + t.m[5] * qy + t.m[6] * qz + t.m[7] * w;
        az += t.m[8] * qx + t.m[9] * qy + t.m[10] * qz + t.m[11] * w;
      }
    }
    o[3 * l] = ax; o[3 * l + 1] = ay; o[3 * l + 2] = az;
  }
}

}  // namespace ut
