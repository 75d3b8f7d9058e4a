// Per-frame keypoint metrics of the eval scripts, one thread per (hand, frame):
//   keypoint error   = mean over 21 landmarks of |gt - tracked|              load_eval.py:33-34, run_eval_known_skeleton.py:92-93
//   acceleration     = mean over landmarks of |p[t] + p[t+2] - 2 p[t+1]|      load_eval.py:29-31 (tracked and gt)
//   valid_acc        = valid[t] & valid[t+1] & valid[t+2]                     load_eval.py:35-37
// Keypoints are float32 on the device (they come out of ut_fk); the reference holds the same float32 values in
// float64 arrays and does this arithmetic in float64 - so does the kernel.  Latency-bound, 1 KB per thread.
#include "ut_kernels.h"

namespace ut {

__device__ inline double mean_norm21(const float* a0, double wa0, const float* a1, double wa1, const float* a2, double wa2) {
  double s = 0.0;
  for (int l = 0; l < 21; ++l) {
    double d[3];
    for (int k = 0; k < 3; ++k) {
      double v = wa0 * (double)a0[3 * l + k] + wa1 * (double)a1[3 * l + k];
      if (a2) v += wa2 * (double)a2[3 * l + k];
      d[k] = v;
    }
    s += sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  }
  return s / 21.0;
}

__global__ __launch_bounds__(64) void keypoint_metrics_kernel(const float* __restrict__ gt, const float* __restrict__ tracked,
                                                              const uint8_t* __restrict__ valid, int n_hands, int n_frames,
                                                              double* __restrict__ err, double* __restrict__ acc,
                                                              double* __restrict__ gt_acc, uint8_t* __restrict__ valid_acc) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n_hands * n_frames) return;
  const int h = i / n_frames, t = i - h * n_frames;
  const float* g = gt + (size_t)i * 63;
  const float* p = tracked + (size_t)i * 63;
  err[i] = mean_norm21(g, 1.0, p, -1.0, nullptr, 0.0);
  if (t + 2 < n_frames) {
    const size_t o = (size_t)h * (n_frames - 2) + t;
    // pts[t] + pts[t+2] - 2 pts[t+1], summed in that order
    acc[o] = mean_norm21(p, 1.0, p + 126, 1.0, p + 63, -2.0);
    gt_acc[o] = mean_norm21(g, 1.0, g + 126, 1.0, g + 63, -2.0);
    valid_acc[o] = (valid[i] && valid[i + 1] && valid[i + 2]) ? 1 : 0;
  }
}

hipError_t launch_keypoint_metrics(const float* gt, const float* tracked, const uint8_t* valid, int n_hands, int n_frames,
                                   double* err, double* acc, double* gt_acc, uint8_t* valid_acc, hipStream_t s) {
  const int n = n_hands * n_frames;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(keypoint_metrics_kernel, dim3((n + 63) / 64), dim3(64), 0, s, gt, tracked, valid, n_hands, n_frames, err,
                     acc, gt_acc, valid_acc);
  return hipGetLastError();
}

}  // namespace ut
