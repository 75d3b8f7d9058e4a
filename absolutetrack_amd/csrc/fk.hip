// Batched forward kinematics + linear blend skinning of the 21 hand landmarks, fp32.
// Replaces lib/common/hand_skinning.py:17-209 (skin_landmarks) and the rotation exponential it
// takes from pytorch3d (so3_exp_map, eps = 1e-4: theta = sqrt(max(|v|^2, eps)),
// R = I + sin(theta)/theta K + (1-cos(theta))/theta^2 K^2).
// ~4 kFLOP and 0.9 KB per pose - latency bound, not a matrix-core shape.
// 17 skinning frames = [wrist, wrist, then per finger W*L0*L1, W*L0*L1*L2, W*L0*L1*L2*L3]
// (the one-joint product is dropped, hand_skinning.py:32).
#include "ut_fk.h"
#include "ut_kernels.h"

namespace ut {

// Three phases per block of FK_P poses, all operands through LDS so that no thread indexes a private array
// dynamically: (1) one thread per (pose, joint) builds the joint's local transform (sin/cos), (2) one thread per
// (pose, finger) multiplies the chain wrist*L0*L1*L2*L3 and keeps the frames after 2, 3, 4 joints, (3) one thread
// per (pose, landmark) blends.  Same arithmetic, in the same order, as the one-thread-per-pose
// skin_landmarks_dev of ut_fk.h (which cropgen.hip still uses inside its own per-candidate thread).
constexpr int FK_P = 12;

__global__ __launch_bounds__(256) void fk_kernel(const float* __restrict__ hand_model, int n_models,
                                                 const float* __restrict__ ja, int ja_stride,
                                                 const float* __restrict__ xf, int xf_stride,
                                                 const int64_t* __restrict__ mirror, float t_scale, int n,
                                                 float* __restrict__ out) {
  __shared__ float s_local[FK_P][20][12];
  __shared__ float s_frame[FK_P][17][12];
  const int tid = threadIdx.x;
  const int base = blockIdx.x * FK_P;
  // ---- phase 1: joint local transforms (20 per pose) and the wrist frames (slots 0, 1)
  if (tid < FK_P * 20) {
    const int pl = tid / 20, q = tid - pl * 20;
    const int i = base + pl;
    if (i < n) {
      const float* hm = hand_model + (size_t)(n_models == 1 ? 0 : i) * 321;
      const M34 l = joint_local(hm + 3 * q, hm + 66 + 3 * q, ja[(size_t)i * ja_stride + q]);
#pragma unroll
      for (int k = 0; k < 12; ++k) s_local[pl][q][k] = l.m[k];
    }
  } else if (tid < FK_P * 20 + FK_P) {
    const int pl = tid - FK_P * 20;
    const int i = base + pl;
    if (i < n) {
      const float* x = xf + (size_t)i * xf_stride;
      M34 w;
#pragma unroll
      for (int k = 0; k < 12; ++k) w.m[k] = x[k];
      w.m[3] *= t_scale; w.m[7] *= t_scale; w.m[11] *= t_scale;
      if (mirror && mirror[i] == 1) { w.m[0] = -w.m[0]; w.m[4] = -w.m[4]; w.m[8] = -w.m[8]; }
#pragma unroll
      for (int k = 0; k < 12; ++k) { s_frame[pl][0][k] = w.m[k]; s_frame[pl][1][k] = w.m[k]; }
    }
  }
  __syncthreads();
  // ---- phase 2: finger chains
  if (tid < FK_P * 5) {
    const int pl = tid / 5, f = tid - pl * 5;
    if (base + pl < n) {
      M34 t;
#pragma unroll
      for (int k = 0; k < 12; ++k) t.m[k] = s_frame[pl][0][k];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        M34 l;
#pragma unroll
        for (int k = 0; k < 12; ++k) l.m[k] = s_local[pl][4 * f + j][k];
        t = mul34(t, l);
        if (j >= 1) {
#pragma unroll
          for (int k = 0; k < 12; ++k) s_frame[pl][2 + 3 * f + (j - 1)][k] = t.m[k];
        }
      }
    }
  }
  __syncthreads();
  // ---- phase 3: linear blend skinning, frames visited in ascending order like the dense reference sum
  if (tid < FK_P * 21) {
    const int pl = tid / 21, l = tid - pl * 21;
    const int i = base + pl;
    if (i < n) {
      const float* hm = hand_model + (size_t)(n_models == 1 ? 0 : i) * 321;
      const float* lm = hm + 132;
      const float* wts = hm + 195;
      const float* idx = hm + 258;
      const float px = lm[3 * l], py = lm[3 * l + 1], pz = lm[3 * l + 2];
      const float w0 = wts[3 * l], w1 = wts[3 * l + 1], w2 = wts[3 * l + 2];
      const int i0 = (int)idx[3 * l], i1 = (int)idx[3 * l + 1], i2 = (int)idx[3 * l + 2];
      float ax = 0.f, ay = 0.f, az = 0.f;
      for (int f = 0; f < 17; ++f) {
        // dense skinning weight of frame f: the last non-zero entry naming it wins
        float w = 0.f;
        if (w0 != 0.f && i0 == f) w = w0;
        if (w1 != 0.f && i1 == f) w = w1;
        if (w2 != 0.f && i2 == f) w = w2;
        if (w != 0.f) {
          const float* t = s_frame[pl][f];
          const float qx = px * w, qy = py * w, qz = pz * w;   // (p,1) * w, as the reference scales first
          ax += t[0] * qx + t[1] * qy + t[2] * qz + t[3] * w;
          ay += t[4] * qx + t[5] * qy + t[6] * qz + t[7] * w;
          az += t[8] * qx + t[9] * qy + t[10] * qz + t[11] * w;
        }
      }
      float* o = out + (size_t)i * 63 + 3 * l;
      o[0] = ax; o[1] = ay; o[2] = az;
    }
  }
}

hipError_t launch_fk(const float* hand_model, int n_models, const float* ja, int ja_stride, const float* xf,
                     int xf_stride, const int64_t* mirror, float t_scale, int n, float* out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(fk_kernel, dim3((n + FK_P - 1) / FK_P), dim3(256), 0, s, hand_model, n_models, ja, ja_stride, xf,
                     xf_stride, mirror, t_scale, n, out);
  return hipGetLastError();
}

}  // namespace ut
