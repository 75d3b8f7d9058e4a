// Fisheye -> pinhole crop resampler for a batch of crops.
// Replaces lib/tracker/tracker.py:61-89 (_warp_image: coordinate map through
// PinholePlaneCameraModel.window_to_eye -> eye_to_world -> Fisheye62CameraModel.world_to_eye ->
// eye_to_window, lib/common/camera.py:61-85,122-143,296-329, then cv2.remap INTER_LINEAR with
// constant-0 border) and the /255 of tracker.py:332.
// One thread per destination pixel.  The map is evaluated in fp64 and cast to fp32 exactly as the
// reference's numpy code does; the gather of the 4 source taps is what touches HBM (u8 source,
// ~160x160 px footprint per crop) - latency/gather bound, no LDS reuse to exploit.
#include "ut_kernels.h"

namespace ut {

constexpr int CROP_PX = 96 * 96;

__device__ inline int tap(const uint8_t* img, int h, int w, int x, int y) {
  return (x >= 0 && x < w && y >= 0 && y < h) ? (int)img[y * w + x] : 0;
}

// the fp32 source window coordinates of destination pixel (px, py): lib/tracker/tracker.py:69-82
__device__ __forceinline__ void warp_coords(const double* __restrict__ cp, const double* __restrict__ sp, int px, int py,
                                            float& mx, float& my) {
  // crop pinhole: unproject to a unit ray (camera.py:69-75, affine.py:22-24)
  double qx = ((double)px - cp[2]) / cp[0], qy = ((double)py - cp[3]) / cp[1];
  double nrm = fmax(5.43e-20, sqrt(qx * qx + qy * qy + 1.0));
  double vx = qx / nrm, vy = qy / nrm, vz = 1.0 / nrm;
  // eye -> world (camera.py:302-306)
  const double* rc = cp + 4;
  const double* tc = cp + 13;
  double wx = rc[0] * vx + rc[1] * vy + rc[2] * vz + tc[0];
  double wy = rc[3] * vx + rc[4] * vy + rc[5] * vz + tc[1];
  double wz = rc[6] * vx + rc[7] * vy + rc[8] * vz + tc[2];
  // world -> source eye: R^T (w - t) (camera.py:296-300)
  const double* rs = sp + 12;
  const double* ts = sp + 21;
  double dx = wx - ts[0], dy = wy - ts[1], dz = wz - ts[2];
  double ex = rs[0] * dx + rs[3] * dy + rs[6] * dz;
  double ey = rs[1] * dx + rs[4] * dy + rs[7] * dz;
  double ez = rs[2] * dx + rs[5] * dy + rs[8] * dz;
  // arctan projection (camera.py:80-85)
  double r = sqrt(ex * ex + ey * ey);
  double sc = atan2(r, ez) / fmax(r, 2.938735877055719e-39 /* 2^-128 */);
  double ux = ex * sc, uy = ey * sc;
  // Fisheye62 distortion (camera.py:122-143)
  const double k1 = sp[4], k2 = sp[5], k3 = sp[6], k4 = sp[7], p1 = sp[8], p2 = sp[9], k5 = sp[10], k6 = sp[11];
  const double pi2 = 9.869604401089358;
  double r2 = fmin(fmax(ux * ux + uy * uy, -pi2), pi2);
  double r4 = r2 * r2, r6 = r2 * r4;
  double radial = 1 + k1 * r2 + k2 * r4 + k3 * r6 + k4 * (r4 * r4) + k5 * (r4 * r6) + k6 * (r6 * r6);
  double x = ux * radial, y = uy * radial;
  double x2 = x * x, y2 = y * y, xy = x * y, rr = x2 + y2;
  double xd = x + (2 * p2 * xy + p1 * (rr + 2 * x2));
  double yd = y + (2 * p1 * xy + p2 * (rr + 2 * y2));
  double mxd = xd * sp[0] + sp[2], myd = yd * sp[1] + sp[3];
  if (ez < 0) { mxd = -1.0; myd = -1.0; }      // tracker.py:78-80
  mx = (float)mxd; my = (float)myd;            // tracker.py:82
}

// diagnostic entry (ut_warp_map): the coordinate map itself, [n_crops, 96, 96, 2] fp32 (x, y)
__global__ __launch_bounds__(256) void warp_map_kernel(const double* __restrict__ cam, const double* __restrict__ crop,
                                                       const int32_t* __restrict__ src_index, int n_src,
                                                       float* __restrict__ out) {
  const int ci = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int si = src_index[ci];
  float mx = -1.f, my = -1.f;
  if (si >= 0 && si < n_src) warp_coords(crop + (size_t)ci * 24, cam + (size_t)si * 32, pix % 96, pix / 96, mx, my);
  reinterpret_cast<float2*>(out)[(size_t)ci * CROP_PX + pix] = make_float2(mx, my);
}

template <bool U8>
__global__ __launch_bounds__(256) void warp_kernel(const uint8_t* __restrict__ src, int src_h, int src_w,
                                                   const double* __restrict__ cam,
                                                   const double* __restrict__ crop,
                                                   const int32_t* __restrict__ src_index, int n_src, int mode,
                                                   float* __restrict__ out, uint8_t* __restrict__ out_u8,
                                                   int* __restrict__ status) {
  const int ci = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= CROP_PX) return;
  const int si = src_index[ci];
  if (si < 0 || si >= n_src) {      // the reference would raise IndexError on views[cam_idx] (lib/tracker/tracker.py:330)
    if constexpr (U8) out_u8[(size_t)ci * CROP_PX + pix] = 0;
    else out[(size_t)ci * CROP_PX + pix] = 0.f;
    if (pix == 0) atomicOr(status, UT_BAD_SRC_INDEX);
    return;
  }
  float mx, my;
  warp_coords(crop + (size_t)ci * 24, cam + (size_t)si * 32, pix % 96, pix / 96, mx, my);
  const uint8_t* img = src + (size_t)si * src_h * src_w;
  float result;
  if (mode == 0) {
    // OpenCV CV_8U INTER_LINEAR arithmetic: 1/32 px coordinates, 15-bit weights, rounded u8
    float fx32 = fminf(fmaxf(mx * 32.0f, -1.0e9f), 1.0e9f), fy32 = fminf(fmaxf(my * 32.0f, -1.0e9f), 1.0e9f);
    int sx = __float2int_rn(fx32), sy = __float2int_rn(fy32);
    int ix = sx >> 5, iy = sy >> 5, ax = sx & 31, ay = sy & 31;
    ix = min(max(ix, -32768), 32767);
    iy = min(max(iy, -32768), 32767);
    int w00 = (32 - ay) * (32 - ax) * 32, w01 = (32 - ay) * ax * 32, w10 = ay * (32 - ax) * 32, w11 = ay * ax * 32;
    int v = tap(img, src_h, src_w, ix, iy) * w00 + tap(img, src_h, src_w, ix + 1, iy) * w01 +
            tap(img, src_h, src_w, ix, iy + 1) * w10 + tap(img, src_h, src_w, ix + 1, iy + 1) * w11;
    result = (float)((v + (1 << 14)) >> 15);
  } else {
    float cx = fminf(fmaxf(mx, -1.0e6f), 1.0e6f), cy = fminf(fmaxf(my, -1.0e6f), 1.0e6f);
    float x0 = floorf(cx), y0 = floorf(cy);
    float fx = cx - x0, fy = cy - y0;
    int ix = (int)x0, iy = (int)y0;
    double v = (double)tap(img, src_h, src_w, ix, iy) * (double)((1.0f - fx) * (1.0f - fy)) +
               (double)tap(img, src_h, src_w, ix + 1, iy) * (double)(fx * (1.0f - fy)) +
               (double)tap(img, src_h, src_w, ix, iy + 1) * (double)((1.0f - fx) * fy) +
               (double)tap(img, src_h, src_w, ix + 1, iy + 1) * (double)(fx * fy);
    result = (float)v;
  }
  if constexpr (U8) out_u8[(size_t)ci * CROP_PX + pix] = (uint8_t)result;   // mode 0 only: an exact grey level
  else out[(size_t)ci * CROP_PX + pix] = result / 255.0f;
}

hipError_t launch_warp(const uint8_t* src, int n_src, int src_h, int src_w, const double* cam, const double* crop,
                       const int32_t* src_index, int n_crops, int mode, float* out, uint8_t* out_u8, int* status,
                       hipStream_t s) {
  if (out_u8 && mode != 0) return hipErrorInvalidValue;   // only OpenCV's 8-bit arithmetic yields exact grey levels
  for (int done = 0; done < n_crops;) {
    int cnt = n_crops - done < 32768 ? n_crops - done : 32768;
    if (out_u8)
      hipLaunchKernelGGL(warp_kernel<true>, dim3(CROP_PX / 256, cnt), dim3(256), 0, s, src, src_h, src_w, cam,
                         crop + (size_t)done * 24, src_index + done, n_src, mode, (float*)nullptr,
                         out_u8 + (size_t)done * CROP_PX, status);
    else
      hipLaunchKernelGGL(warp_kernel<false>, dim3(CROP_PX / 256, cnt), dim3(256), 0, s, src, src_h, src_w, cam,
                         crop + (size_t)done * 24, src_index + done, n_src, mode, out + (size_t)done * CROP_PX,
                         (uint8_t*)nullptr, status);
    done += cnt;
  }
  return hipGetLastError();
}

hipError_t launch_warp_map(const double* cam, const double* crop, const int32_t* src_index, int n_src, int n_crops,
                           float* out, hipStream_t s) {
  for (int done = 0; done < n_crops;) {
    int cnt = n_crops - done < 32768 ? n_crops - done : 32768;
    hipLaunchKernelGGL(warp_map_kernel, dim3(CROP_PX / 256, cnt), dim3(256), 0, s, cam, crop + (size_t)done * 24,
                       src_index + done, n_src, out + (size_t)done * CROP_PX * 2);
    done += cnt;
  }
  return hipGetLastError();
}

}  // namespace ut
