// Batched forward kinematics + linear blend skinning of the 21 hand landmarks, fp32.
// Replaces lib/common/hand_skinning.py:17-209 (skin_landmarks) and the rotation exponential it
// takes from pytorch3d (so3_exp_map, eps = 1e-4: theta = sqrt(max(|v|^2, eps)),
// R = I + sin(theta)/theta K + (1-cos(theta))/theta^2 K^2).
// One thread per pose: ~4 kFLOP and 0.9 KB each - latency bound, not a matrix-core shape.
// 17 skinning frames = [wrist, wrist, then per finger W*L0*L1, W*L0*L1*L2, W*L0*L1*L2*L3]
// (the one-joint product is dropped, hand_skinning.py:32).
#include "ut_fk.h"
#include "ut_kernels.h"

namespace ut {

__global__ __launch_bounds__(64) void fk_kernel(const float* __restrict__ hand_model, int n_models,
                                                const float* __restrict__ ja, int ja_stride,
                                                const float* __restrict__ xf, int xf_stride,
                                                const int64_t* __restrict__ mirror, float t_scale, int n,
                                                float* __restrict__ out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const float* hm = hand_model + (size_t)(n_models == 1 ? 0 : i) * 321;
  const float* a = ja + (size_t)i * ja_stride;
  const float* x = xf + (size_t)i * xf_stride;
  M34 wrist;
#pragma unroll
  for (int k = 0; k < 12; ++k) wrist.m[k] = x[k];
  wrist.m[3] *= t_scale; wrist.m[7] *= t_scale; wrist.m[11] *= t_scale;
  if (mirror && mirror[i] == 1) { wrist.m[0] = -wrist.m[0]; wrist.m[4] = -wrist.m[4]; wrist.m[8] = -wrist.m[8]; }
  skin_landmarks_dev(hm, a, wrist, out + (size_t)i * 63);
}

hipError_t launch_fk(const float* hand_model, int n_models, const float* ja, int ja_stride, const float* xf,
                     int xf_stride, const int64_t* mirror, float t_scale, int n, float* out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(fk_kernel, dim3((n + 63) / 64), dim3(64), 0, s, hand_model, n_models, ja, ja_stride, xf,
                     xf_stride, mirror, t_scale, n, out);
  return hipGetLastError();
}

}  // namespace ut
