// Batched crop-camera generation: for every (frame, hand) candidate, in ONE launch,
// what HandTracker.gen_crop_cameras does per frame on the host in the reference:
//   crop points = FK of (label pose, neutral pose, open pose)          lib/tracker/perspective_crop.py:89-133
//   visible-landmark count per fisheye camera                          lib/tracker/perspective_crop.py:54-86
//   first `max_views` eligible cameras in index order                  :157-178 with sort_camera_index=True
//   per selected camera: look-at rotation towards the crop centre, roll by the camera angle, x-mirror for
//   right hands, largest focal that keeps all 63 points in a 96x96 image, times hand_ratio_in_crop
//                                                                      lib/common/crop.py:15-82, lib/common/affine.py:34-76
// and the network-side camera inputs of lib/tracker/tracker.py:333-337 (K, world->eye with t in metres).
// FK in fp32 (the reference runs it in torch fp32), geometry in fp64 (numpy).  One wave per candidate: a few
// kFLOP and ~1.5 KB each - latency bound; the win over the reference is doing all frames in one launch
// instead of ~10 ms of numpy/scipy/torch calls per frame.
#include "ut_fk.h"
#include "ut_kernels.h"
#include "ut_math.h"

namespace ut {

namespace {

// world -> eye of a camera given as cam_params row (R at [12..20], t at [21..23] of camera_to_world)
__device__ inline void world_to_eye_d(const double* cam, const double* w, double* e) {
  const double* r = cam + 12;
  const double* t = cam + 21;
  const double dx = w[0] - t[0], dy = w[1] - t[1], dz = w[2] - t[2];
  e[0] = r[0] * dx + r[3] * dy + r[6] * dz;
  e[1] = r[1] * dx + r[4] * dy + r[7] * dz;
  e[2] = r[2] * dx + r[5] * dy + r[8] * dz;
}

// Fisheye62 eye -> window (lib/common/camera.py:80-85,122-143,308-312)
__device__ inline void fisheye_project_d(const double* cam, const double* e, double* win) {
  const double r = sqrt(e[0] * e[0] + e[1] * e[1]);
  const double sc = atan2(r, e[2]) / fmax(r, 2.938735877055719e-39);
  const double ux = e[0] * sc, uy = e[1] * sc;
  const double k1 = cam[4], k2 = cam[5], k3 = cam[6], k4 = cam[7], p1 = cam[8], p2 = cam[9], k5 = cam[10], k6 = cam[11];
  const double pi2 = 9.869604401089358;
  const double r2 = fmin(fmax(ux * ux + uy * uy, -pi2), pi2);
  const double r4 = r2 * r2, r6 = r2 * r4;
  const double radial = 1 + k1 * r2 + k2 * r4 + k3 * r6 + k4 * (r4 * r4) + k5 * (r4 * r6) + k6 * (r6 * r6);
  const double x = ux * radial, y = uy * radial;
  const double x2 = x * x, y2 = y * y, xy = x * y, rr = x2 + y2;
  win[0] = (x + (2 * p2 * xy + p1 * (rr + 2 * x2))) * cam[0] + cam[2];
  win[1] = (y + (2 * p1 * xy + p2 * (rr + 2 * y2))) * cam[1] + cam[3];
}

__device__ inline void mat3_mul(const double* a, const double* b, double* c) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

// Result of aiming a square pinhole crop camera at a point cloud (lib/common/crop.py:31-82).
struct CropFit {
  double focal, cxy;
  double w2e[16];   // new world->eye (after the optional x mirror)
  double c2w[16];   // its general inverse = the crop camera's camera_to_world_xf
  bool bad;         // the reference raises "Unable to create crop camera" (crop.py:25-26)
};

// c2w0: the original camera's camera_to_world (row major 4x4).  Same operation order as the host code
// (crop.py:57-82, affine.py:47-76): general inverse of camera_to_world, aim +z at `center` in that eye frame,
// inverse back, right-multiply the rotation by aim and roll, inverse again, mirror, fit the focal length.
__device__ inline void fit_begin(const double* c2w0, const double* center, double angle_deg, bool mirror, double* w2e) {
  double w2e0[16], e2w[16];
  inv4(c2w0, w2e0);
  double c_eye[3];
  for (int i = 0; i < 3; ++i)
    c_eye[i] = w2e0[4 * i] * center[0] + w2e0[4 * i + 1] * center[1] + w2e0[4 * i + 2] * center[2] + w2e0[4 * i + 3];
  const double cn = sqrt(c_eye[0] * c_eye[0] + c_eye[1] * c_eye[1] + c_eye[2] * c_eye[2]);
  double b[3] = {c_eye[0] / cn, c_eye[1] / cn, c_eye[2] / cn};
  const double bn = fmax(5.43e-20, sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]));
  b[0] /= bn; b[1] /= bn; b[2] /= bn;
  // from_two_vectors((0,0,1), b): v = a x b, R = I + K + K^2 (1-a.b)/max(|v|^2, 1e-15)   (affine.py:34-44)
  const double v[3] = {-b[1], b[0], 0.0};
  const double sn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const double k[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
  double k2[9], aim[9];
  mat3_mul(k, k, k2);
  const double f = (1 - b[2]) / fmax(sn * sn, 1e-15);
  for (int i = 0; i < 9; ++i) aim[i] = ((i % 4 == 0) ? 1.0 : 0.0) + k[i] + k2[i] * f;
  const double ang = angle_deg * (3.141592653589793 / 180.0);
  const double rz[9] = {cos(ang), -sin(ang), 0, sin(ang), cos(ang), 0, 0, 0, 1};
  inv4(w2e0, e2w);
  double r0[9], r1[9], r2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r0[3 * i + j] = e2w[4 * i + j];
  mat3_mul(r0, aim, r1);
  mat3_mul(r1, rz, r2);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) e2w[4 * i + j] = r2[3 * i + j];
  inv4(e2w, w2e);
  if (mirror)                      // diag(-1,1,1,1) @ w2e (crop.py:63-66)
    for (int j = 0; j < 4; ++j) w2e[j] = -w2e[j];
}

// one bounding point through the new world->eye: largest |x/z|, |y/z| so far and the "behind the camera" flag
// (crop.py:15-28); max is exact, so the points may be visited in any order (or by different lanes)
__device__ inline void fit_point(const double* w2e, const float* pt, double& max_ndc, bool& bad) {
  const double px = (double)pt[0], py = (double)pt[1], pz = (double)pt[2];
  const double ex = w2e[0] * px + w2e[1] * py + w2e[2] * pz + w2e[3];
  const double ey = w2e[4] * px + w2e[5] * py + w2e[6] * pz + w2e[7];
  const double ez = w2e[8] * px + w2e[9] * py + w2e[10] * pz + w2e[11];
  if (ez < 0.0001) bad = true;
  max_ndc = fmax(max_ndc, fmax(fabs(ex / ez), fabs(ey / ez)));
}

__device__ inline void fit_end(double max_ndc, bool bad, int crop_size, double focal_multiplier, CropFit& o) {
  o.cxy = ((double)crop_size - 1.0) / 2.0;
  const double fxy = o.cxy / max_ndc;
  o.bad = bad || fxy < 5.0;
  o.focal = focal_multiplier * fxy;
  inv4(o.w2e, o.c2w);              // crop.py:81
}

__device__ inline void fit_crop_camera(const double* c2w0, const float* pts, int n_pts, const double* center,
                                       double angle_deg, bool mirror, int crop_size, double focal_multiplier,
                                       CropFit& o) {
  fit_begin(c2w0, center, angle_deg, mirror, o.w2e);
  double max_ndc = 0.0;
  bool bad = false;
  for (int q = 0; q < n_pts; ++q) fit_point(o.w2e, pts + 3 * q, max_ndc, bad);
  fit_end(max_ndc, bad, crop_size, focal_multiplier, o);
}

// middle of the bounding box, (pts.min + pts.max) / 2.0 (crop.py:60): in fp32 like numpy on float32 points
// (the tracker path, whose cameras are float64 so that the rest of its chain is float64 in the reference too),
// or exactly (the torch_data path, which is compared with the reference's functions run on float64 copies of
// its all-float32 inputs because their float32 LAPACK chain is not a machine-independent bit pattern)
__device__ inline void bbox_center(const float* pts, int n_pts, double* center, bool exact = false) {
  float lo[3], hi[3];
  for (int d = 0; d < 3; ++d) { lo[d] = 3.0e38f; hi[d] = -3.0e38f; }
  for (int q = 0; q < n_pts; ++q)
    for (int d = 0; d < 3; ++d) {
      lo[d] = fminf(lo[d], pts[3 * q + d]);
      hi[d] = fmaxf(hi[d], pts[3 * q + d]);
    }
  for (int d = 0; d < 3; ++d)
    center[d] = exact ? ((double)lo[d] + (double)hi[d]) / 2.0 : (double)((lo[d] + hi[d]) / 2.0f);
}

}  // namespace

// One WAVE per candidate (one 64-thread workgroup): the pieces of the per-candidate work that are independent run on
// different lanes and meet in LDS -
//   (1) joint transforms of the three crop poses: 3 x 20 lanes            (fk.hip phases 1-3, same arithmetic)
//   (2) finger chains: 3 x 5 lanes   (3) landmarks: 3 x 21 lanes          -> 63 crop points, fp32
//   (4) bounding-box centre: 3 lanes (one per axis; min / max are exact)
//   (5) visible-landmark count: (camera, landmark) pairs on the lanes, three cameras per pass
//   (6) camera selection: lane 0 (first max_views eligible cameras in index order)
//   (7) look-at fit: lane v builds view v's new world->eye; the 63 bounding points go one per lane through it
//       (max / or reductions by wave shuffles: exact in any order); lane v finishes the view and writes its outputs.
// Every value is computed by the same formula, in the same operation order, as the former one-thread-per-candidate
// kernel (253 us per launch however few candidates - the per-frame tracker paid that every frame); the results are
// bit-identical.
constexpr int CG_MAX_CAMS = 16, CG_MAX_VIEWS = 4;

__global__ __launch_bounds__(64) void cropgen_kernel(CropGenArgs g) {
  const int s = blockIdx.x;
  const int lane = threadIdx.x;
  const int frame = g.frame_idx[s];
  const int hand = (int)g.hand_idx[s];
  const float* hm = g.hand_model + (size_t)(g.n_models == 1 ? 0 : s) * 321;
  __shared__ float s_local[3][20][12];
  __shared__ float s_frame[3][17][12];
  __shared__ float s_pts[189];
  __shared__ double s_center[3];
  __shared__ int s_vis[CG_MAX_CAMS];
  __shared__ int s_sel[CG_MAX_VIEWS];
  __shared__ int s_nsel;
  __shared__ double s_w2e[CG_MAX_VIEWS][16];

  // ---- (1) joint local transforms of the label pose, the neutral pose and the open pose; wrist frames
  if (lane < 60) {
    const int pz = lane / 20, q = lane - pz * 20;
    float angle;
    if (pz == 0) angle = g.joint_angles[(size_t)s * 22 + q];
    else if (pz == 1) {                                   // perspective_crop.py:19-24
      const float* lim = g.joint_limits + (size_t)(g.n_models == 1 ? 0 : s) * 44;
      angle = lim[2 * q] * 0.5f + lim[2 * q + 1] * (1.0f - 0.5f);
    } else angle = 0.f;
    const M34 l = joint_local(hm + 3 * q, hm + 66 + 3 * q, angle);
#pragma unroll
    for (int k = 0; k < 12; ++k) s_local[pz][q][k] = l.m[k];
  } else if (lane < 63) {
    const int pz = lane - 60;
    const float* x = g.wrist_xf + (size_t)s * 16;
    M34 w;
#pragma unroll
    for (int k = 0; k < 12; ++k) w.m[k] = x[k];
    if (hand == 1) { w.m[0] = -w.m[0]; w.m[4] = -w.m[4]; w.m[8] = -w.m[8]; }
#pragma unroll
    for (int k = 0; k < 12; ++k) { s_frame[pz][0][k] = w.m[k]; s_frame[pz][1][k] = w.m[k]; }
  }
  if (lane < CG_MAX_CAMS) s_vis[lane] = 0;
  __syncthreads();
  // ---- (2) finger chains
  if (lane < 15) {
    const int pz = lane / 5, f = lane - pz * 5;
    M34 t;
#pragma unroll
    for (int k = 0; k < 12; ++k) t.m[k] = s_frame[pz][0][k];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      M34 l;
#pragma unroll
      for (int k = 0; k < 12; ++k) l.m[k] = s_local[pz][4 * f + j][k];
      t = mul34(t, l);
      if (j >= 1) {
#pragma unroll
        for (int k = 0; k < 12; ++k) s_frame[pz][2 + 3 * f + (j - 1)][k] = t.m[k];
      }
    }
  }
  __syncthreads();
  // ---- (3) linear blend skinning, frames visited in ascending order like the dense reference sum
  if (lane < 63) {
    const int pz = lane / 21, l = lane - pz * 21;
    const float* lm = hm + 132;
    const float* wts = hm + 195;
    const float* idx = hm + 258;
    const float px = lm[3 * l], py = lm[3 * l + 1], pzz = lm[3 * l + 2];
    const float w0 = wts[3 * l], w1 = wts[3 * l + 1], w2 = wts[3 * l + 2];
    const int i0 = (int)idx[3 * l], i1 = (int)idx[3 * l + 1], i2 = (int)idx[3 * l + 2];
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int f = 0; f < 17; ++f) {
      float w = 0.f;
      if (w0 != 0.f && i0 == f) w = w0;
      if (w1 != 0.f && i1 == f) w = w1;
      if (w2 != 0.f && i2 == f) w = w2;
      if (w != 0.f) {
        const float* t = s_frame[pz][f];
        const float qx = px * w, qy = py * w, qz = pzz * w;
        ax += t[0] * qx + t[1] * qy + t[2] * qz + t[3] * w;
        ay += t[4] * qx + t[5] * qy + t[6] * qz + t[7] * w;
        az += t[8] * qx + t[9] * qy + t[10] * qz + t[11] * w;
      }
    }
    s_pts[3 * lane] = ax; s_pts[3 * lane + 1] = ay; s_pts[3 * lane + 2] = az;
    if (pz == 0 && g.landmarks) {   // = landmarks_from_hand_pose(hand_model, pose, hand_idx), same FK as ut_fk
      float* o = g.landmarks + (size_t)s * 63 + 3 * l;
      o[0] = ax; o[1] = ay; o[2] = az;
    }
  }
  __syncthreads();
  // ---- (4) middle of the bounding box in fp32 like (pts.min + pts.max) / 2.0 on float32 points (crop.py:60)
  if (lane < 3) {
    float lo = 3.0e38f, hi = -3.0e38f;
    for (int q = 0; q < 63; ++q) {
      lo = fminf(lo, s_pts[3 * q + lane]);
      hi = fmaxf(hi, s_pts[3 * q + lane]);
    }
    s_center[lane] = (double)((lo + hi) / 2.0f);
  }
  // ---- (5) visibility of the label-pose landmarks in every camera
  for (int c0 = 0; c0 < g.n_cams; c0 += 3) {
    const int ci = c0 + lane / 21, l = lane % 21;
    if (lane < 63 && ci < g.n_cams) {
      const double* cam = g.cam_params + ((size_t)frame * g.n_cams + ci) * 32;
      const double w[3] = {(double)s_pts[3 * l], (double)s_pts[3 * l + 1], (double)s_pts[3 * l + 2]};
      double e[3], win[2];
      world_to_eye_d(cam, w, e);
      fisheye_project_d(cam, e, win);
      if (win[0] >= 0 && win[0] <= g.src_w - 1 && win[1] >= 0 && win[1] <= g.src_h - 1 && e[2] > 0) atomicAdd(&s_vis[ci], 1);
    }
  }
  __syncthreads();
  // ---- (6) the first max_views eligible cameras in index order (sort_camera_index=True)
  if (lane == 0) {
    int n = 0;
    for (int ci = 0; ci < g.n_cams && n < g.max_views; ++ci)
      if (s_vis[ci] >= g.min_vis) s_sel[n++] = ci;
    s_nsel = n;
  }
  __syncthreads();
  const int n_views = s_nsel;
  // ---- (7) look-at fit per selected view
  if (lane < n_views) {
    const int ci = s_sel[lane];
    const double* cam = g.cam_params + ((size_t)frame * g.n_cams + ci) * 32;
    const double* rc = cam + 12;     // camera_to_world rotation (row major), translation at cam+21
    const double* tc = cam + 21;
    const double c2w0[16] = {rc[0], rc[1], rc[2], tc[0], rc[3], rc[4], rc[5], tc[1], rc[6], rc[7], rc[8], tc[2], 0, 0, 0, 1};
    const double center[3] = {s_center[0], s_center[1], s_center[2]};
    double w2e[16];
    fit_begin(c2w0, center, g.camera_angles[ci], hand == 1, w2e);
    for (int k = 0; k < 16; ++k) s_w2e[lane][k] = w2e[k];
  }
  __syncthreads();
  double my_ndc = 0.0;      // of the view this lane will finish (lane v < n_views)
  bool my_bad = false;
  for (int v = 0; v < n_views; ++v) {
    double m = 0.0;
    bool bad = false;
    if (lane < 63) fit_point(s_w2e[v], s_pts + 3 * lane, m, bad);
    int badi = bad ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      m = fmax(m, __shfl_xor(m, off));
      badi |= __shfl_xor(badi, off);
    }
    if (lane == v) { my_ndc = m; my_bad = badi != 0; }
  }
  int my_status = 0;
  if (lane < n_views) {
    CropFit fit;
    for (int k = 0; k < 16; ++k) fit.w2e[k] = s_w2e[lane][k];
    fit_end(my_ndc, my_bad, g.crop_size, g.focal_multiplier, fit);
    if (fit.bad) my_status = 1;
    const double focal = fit.focal, cxy = fit.cxy;
    const double* c2w = fit.c2w;
    // ---- outputs
    double* cp = g.crop_params + ((size_t)s * g.max_views + lane) * 24;
    cp[0] = focal; cp[1] = focal; cp[2] = cxy; cp[3] = cxy;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) cp[4 + 3 * i + j] = c2w[4 * i + j];
      cp[13 + i] = c2w[4 * i + 3];
    }
    for (int i = 16; i < 24; ++i) cp[i] = 0.0;
    float* kk = g.intrinsics + ((size_t)s * g.max_views + lane) * 9;
    kk[0] = (float)focal; kk[1] = 0.f; kk[2] = (float)cxy; kk[3] = 0.f; kk[4] = (float)focal; kk[5] = (float)cxy;
    kk[6] = 0.f; kk[7] = 0.f; kk[8] = 1.f;
    // extrinsics = inv(crop camera_to_world) with the translation in metres (tracker.py:335-337)
    double ext[16];
    inv4(c2w, ext);
    float* ex = g.extrinsics + ((size_t)s * g.max_views + lane) * 16;
    for (int i = 0; i < 16; ++i) ex[i] = (float)((i % 4 == 3 && i < 12) ? ext[i] * 0.001 : ext[i]);
    g.cam_index[(size_t)s * g.max_views + lane] = s_sel[lane];
  } else if (lane < g.max_views) {
    g.cam_index[(size_t)s * g.max_views + lane] = -1;
  }
  // "Unable to create crop camera" on any selected view
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) my_status |= __shfl_xor(my_status, off);
  if (lane == 0) {
    g.n_views[s] = n_views;
    g.status[s] = my_status;
  }
}

// torch_data path: _gen_crop_matrices of lib/batched_dataset/data_transform.py:147-212
// for every (frame, view) of a batch in one launch.  The original cameras are pinholes given as world->eye
// extrinsics + K; the crop camera is the same look-at fit as above with camera_angle 0; outputs are the network's
// extrinsics/intrinsics and the pixel homography of data_transform.py:57-76
//   resample_xf = K_orig44 @ world_to_eye_orig @ eye_to_world_new @ K_new44^-1      (crop pixel -> source pixel).
// The reference runs this chain in float32 (numpy keeps the dtype of the float32 sample through every
// np.linalg.inv, i.e. OpenBLAS sgesv); here it is float64 and rounded once at the end, which reproduces the
// reference's own functions fed float64 copies of the same values (tests/golden/torch_data.npz, *_f64chain).
__global__ __launch_bounds__(64) void cropmat_kernel(CropMatArgs g) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= g.n_frames * g.n_views) return;
  const int frame = i / g.n_views;
  const float* pts = g.crop_points + (size_t)frame * g.n_pts * 3;
  double center[3];
  bbox_center(pts, g.n_pts, center, true);
  double w2e_orig[16], c2w0[16];
  for (int k = 0; k < 16; ++k) w2e_orig[k] = (double)g.orig_extrinsics[(size_t)i * 16 + k];
  inv4(w2e_orig, c2w0);            // camera_to_world_xf=np.linalg.inv(world_to_eye_xf)  (data_transform.py:192)
  CropFit fit;
  fit_crop_camera(c2w0, pts, g.n_pts, center, 0.0, g.hand_idx[frame] == 1, g.crop_size, g.focal_multiplier, fit);
  g.status[i] = fit.bad ? 1 : 0;
  double ext[16];
  inv4(fit.c2w, ext);              // new_world_to_eye_xf = inv(camera_new.camera_to_world_xf)  (:203)
  for (int k = 0; k < 16; ++k) g.extrinsics_xf[(size_t)i * 16 + k] = (float)ext[k];
  float* kn = g.new_intrinsics + (size_t)i * 9;
  kn[0] = (float)fit.focal; kn[1] = 0.f; kn[2] = (float)fit.cxy; kn[3] = 0.f; kn[4] = (float)fit.focal;
  kn[5] = (float)fit.cxy; kn[6] = 0.f; kn[7] = 0.f; kn[8] = 1.f;
  // resample matrix (data_transform.py:57-76)
  const float* ko = g.orig_intrinsics + (size_t)i * 9;
  double k_orig[16] = {0}, k_inv[16] = {0}, w2e0[16], t0[16], t1[16], r[16];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) k_orig[4 * a + b] = (double)ko[3 * a + b];
  k_orig[15] = 1.0;
  // only fx, fy, cx, cy of the original K enter the reference's camera_orig (data_transform.py:179-191)
  k_orig[1] = 0.0; k_orig[4] = 0.0; k_orig[8] = 0.0; k_orig[9] = 0.0; k_orig[10] = 1.0;
  const double fo = fit.focal, co = fit.cxy;
  k_inv[0] = 1.0 / fo; k_inv[2] = -co / fo; k_inv[5] = 1.0 / fo; k_inv[6] = -co / fo; k_inv[10] = 1.0; k_inv[15] = 1.0;
  inv4(c2w0, w2e0);                // world_to_eye_orig = inv(camera_orig.camera_to_world_xf)  (:72)
  mul4(k_orig, w2e0, t0);
  mul4(t0, fit.c2w, t1);
  mul4(t1, k_inv, r);
  for (int k = 0; k < 16; ++k) g.resample_xf[(size_t)i * 16 + k] = (float)r[k];
}

hipError_t launch_cropmat(const CropMatArgs& g, hipStream_t s) {
  const int n = g.n_frames * g.n_views;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(cropmat_kernel, dim3((n + 63) / 64), dim3(64), 0, s, g);
  return hipGetLastError();
}

hipError_t launch_cropgen(const CropGenArgs& g, hipStream_t s) {
  if (g.n <= 0) return hipSuccess;
  if (g.n_cams > CG_MAX_CAMS || g.max_views > CG_MAX_VIEWS) return hipErrorInvalidValue;
  hipLaunchKernelGGL(cropgen_kernel, dim3(g.n), dim3(64), 0, s, g);
  return hipGetLastError();
}

}  // namespace ut
