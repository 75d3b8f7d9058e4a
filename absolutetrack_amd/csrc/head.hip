// Glue kernels of the regression head (everything between the backbone and the pose record that
// is not a convolution).  The 1x1 / 3x3 convolutions of the head run on conv_igemm.hip.
// Feature maps are 6x6; in the workspace they are NHWC [S,36,C].
//
//  ftl_in_kernel              lib/models/model_utils.py:57-104,166-192, lib/models/feature_extractor.py:61-94,96-129
//  ftl_out_temporal_in_kernel lib/models/feature_extractor.py:135-139, lib/models/temporal.py:51-91
//  temporal_out_kernel        lib/models/temporal.py:87-91,133-137, lib/models/umetrack_model.py:198-210
//  skeleton_kernel            lib/models/skeleton_encoder.py:36-53
//  pool_matvec + decode       lib/models/model_utils.py:205-207,17-54, lib/models/regressor.py:76-121,
//                             lib/models/umetrack_model.py:77-97
#include "ut_kernels.h"
#include "ut_math.h"

namespace ut {

constexpr int PIX = 36, FC = 72, MEMC = 18, TC = 92;

// ---------------------------------------------------------------- descriptor checks
// The reference indexes with these tensors in Python and raises IndexError / asserts on bad ones
// (lib/models/temporal.py:101-137, lib/models/umetrack_model.py:149-166,224-229); here a bad entry sets a bit in
// the status word and every kernel below returns before touching memory when an error bit is set.
__global__ __launch_bounds__(256) void validate_desc_kernel(HeadArgs a) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= a.n_samples) return;
  int bits = 0;
  const long long r0 = a.sample_range[2 * s], r1 = a.sample_range[2 * s + 1];
  const long long nv = r1 - r0;
  if (r0 < 0 || r1 > a.n_crops || nv < 1 || nv > 2) bits |= UT_BAD_SAMPLE_RANGE;
  else if (nv == 1) bits |= UT_SINGLE_VIEW;
  const long long slot = a.memory_idx[s];
  if (slot < 0 || slot >= a.n_slots) bits |= UT_BAD_MEMORY_IDX;
  else if (atomicAdd(a.slot_seen + slot, 1) != 0) bits |= UT_DUP_MEMORY_IDX;
  const long long hand = a.hand_idx[s];
  if (hand != 0 && hand != 1) bits |= UT_BAD_HAND_IDX;
  if (bits & UT_STATUS_ERRORS) atomicOr(a.status, bits & UT_STATUS_ERRORS);
  if (bits & ~UT_STATUS_ERRORS) atomicOr(a.status + 1, bits & ~UT_STATUS_ERRORS);
}
#define UT_RETURN_IF_INVALID(a)                                                              \
  if ((((volatile const int*)(a).status)[0] & UT_STATUS_ERRORS) ||                           \
      (((volatile const int*)(a).status)[1] & (a).call_error_mask)) return

// ---------------------------------------------------------------- FTL in (+ view concat)
// One workgroup per sample.  Two-view samples: both views are moved to the canonical space with
// A_v = S_0^-1 X_0 X_v^-1 S_v and written side by side (144 channels).  One-view samples: FTL
// with S_v only, written straight to the fused buffer (the fusion convs are skipped for them).
__global__ __launch_bounds__(256) void ftl_in_kernel(HeadArgs a, HeadBuffers b) {
  UT_RETURN_IF_INVALID(a);
  const int s = blockIdx.x;
  const int r0 = (int)a.sample_range[2 * s], r1 = (int)a.sample_range[2 * s + 1];
  const int nv = r1 - r0;
  __shared__ float xf[2][12];
  if (threadIdx.x < nv) {
    const int v = threadIdx.x;
    const double sv = (double)(a.intrinsics[(size_t)(r0 + v) * 9] / 200.0f);
    double out[16];
    if (nv == 1) {
      for (int i = 0; i < 16; ++i) out[i] = (i % 5 == 0) ? 1.0 : 0.0;
      out[10] = sv;
    } else {
      const double s0 = (double)(a.intrinsics[(size_t)r0 * 9] / 200.0f);
      double x0[16], xv[16], xvi[16], t[16];
      load4(a.extrinsics + (size_t)r0 * 16, x0);
      load4(a.extrinsics + (size_t)(r0 + v) * 16, xv);
      inv4(xv, xvi);
      for (int i = 0; i < 4; ++i) xvi[4 * i + 2] *= sv;     // X_v^-1 S_v  (S scales column 2)
      mul4(x0, xvi, t);
      for (int j = 0; j < 4; ++j) t[8 + j] /= s0;           // S_0^-1 (...) (scales row 2)
      for (int i = 0; i < 16; ++i) out[i] = t[i];
    }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) xf[v][4 * i + j] = (float)out[4 * i + j];
  }
  __syncthreads();
  // 24 channel triples x 36 pixels per view
  for (int it = threadIdx.x; it < 2 * 24 * PIX; it += 256) {
    const int v = it / (24 * PIX);
    const int rem = it - v * 24 * PIX;
    const int c = rem / PIX, p = rem - c * PIX;
    if (nv == 2) {
      const float* f = a.feat + (size_t)(r0 + v) * FC * PIX;
      const float x = f[c * PIX + p], y = f[(24 + c) * PIX + p], z = f[(48 + c) * PIX + p];
      const float* m = xf[v];
      float* o = b.cat144 + ((size_t)s * PIX + p) * 144 + v * FC;
      o[c] = m[0] * x + m[1] * y + m[2] * z + m[3];
      o[24 + c] = m[4] * x + m[5] * y + m[6] * z + m[7];
      o[48 + c] = m[8] * x + m[9] * y + m[10] * z + m[11];
    } else {
      float* o144 = b.cat144 + ((size_t)s * PIX + p) * 144 + v * FC;
      o144[c] = 0.f; o144[24 + c] = 0.f; o144[48 + c] = 0.f;   // keep the unused GEMM rows finite
      if (v == 0) {
        const float* f = a.feat + (size_t)r0 * FC * PIX;
        const float* m = xf[0];
        const float x = f[c * PIX + p], y = f[(24 + c) * PIX + p], z = f[(48 + c) * PIX + p];
        float* o = b.fused + ((size_t)s * PIX + p) * FC;
        o[c] = m[0] * x + m[1] * y + m[2] * z + m[3];
        o[24 + c] = m[4] * x + m[5] * y + m[6] * z + m[7];
        o[48 + c] = m[8] * x + m[9] * y + m[10] * z + m[11];
      }
    }
  }
}

// ---------------------------------------------------------------- FTL out + temporal input
// Two-view samples: fused canonical features back to cam0 space with S_0.  Then the memory of the
// sample's slot is warped by cur_ext * prev_ext^-1 (or zeroed) and concatenated in front:
// t92a[s][p] = [mem'(18) | fused(72) | 0 0].
__global__ __launch_bounds__(256) void ftl_out_temporal_in_kernel(HeadArgs a, HeadBuffers b) {
  UT_RETURN_IF_INVALID(a);
  const int s = blockIdx.x;
  const int r0 = (int)a.sample_range[2 * s], r1 = (int)a.sample_range[2 * s + 1];
  const int nv = r1 - r0;
  const int slot = (int)a.memory_idx[s];
  const bool use_mem = a.use_memory[s] != 0;
  __shared__ float rel[12];
  if (threadIdx.x == 0) {
    double cur[16];
    load4(a.extrinsics + (size_t)r0 * 16, cur);
    if (use_mem) {
      double prev[16], pinv[16], r[16];
      load4(a.prev_ext + (size_t)slot * 16, prev);
      inv4(prev, pinv);
      mul4(cur, pinv, r);
      for (int i = 0; i < 12; ++i) rel[i] = (float)r[i];
    }
    for (int i = 0; i < 16; ++i) a.prev_ext[(size_t)slot * 16 + i] = (float)cur[i];
  }
  __syncthreads();
  const float s0 = a.intrinsics[(size_t)r0 * 9] / 200.0f;
  for (int it = threadIdx.x; it < PIX * TC; it += 256) {
    const int p = it / TC, c = it - p * TC;
    float v = 0.f;
    if (c >= MEMC && c < MEMC + FC) {
      const int fc = c - MEMC;
      const size_t o = ((size_t)s * PIX + p) * FC + fc;
      if (nv == 2) {
        v = b.f72b[o];
        if (fc >= 48) v *= s0;
        b.fused[o] = v;
      } else {
        v = b.fused[o];
      }
    } else if (c < MEMC && use_mem) {
      const float* m = a.mem + ((size_t)slot * PIX + p) * MEMC;
      const int k = c % 6, row = c / 6;
      v = rel[4 * row] * m[k] + rel[4 * row + 1] * m[6 + k] + rel[4 * row + 2] * m[12 + k] + rel[4 * row + 3];
    }
    b.t92a[((size_t)s * PIX + p) * TC + c] = v;
  }
}

// ---------------------------------------------------------------- temporal output split
// regin: [S,36,reg_stride], channels 0 .. reg_c - 1 written, reg_c .. reg_stride - 1 zeroed (reg_stride > reg_c: the regressor's
// tensors padded to the channel count its split-fp16 convolutions run on); out_max (optional): receives the bits of max |regin|
__global__ __launch_bounds__(256) void temporal_out_kernel(HeadArgs a, const float* __restrict__ t_out,
                                                           const float* __restrict__ skel, int n_skel,
                                                           float* __restrict__ regin, int reg_c, int reg_stride, unsigned* out_max) {
  UT_RETURN_IF_INVALID(a);
  const int s = blockIdx.x;
  const int slot = (int)a.memory_idx[s];
  unsigned mx = 0;
  for (int it = threadIdx.x; it < PIX * TC; it += 256) {
    const int p = it / TC, c = it - p * TC;
    if (c >= MEMC + FC) continue;
    const float v = t_out[((size_t)s * PIX + p) * TC + c];
    if (c < MEMC) a.mem[((size_t)slot * PIX + p) * MEMC + c] = v;
    else {
      regin[((size_t)s * PIX + p) * reg_stride + (c - MEMC)] = v;
      mx = max(mx, abs_bits(v));
    }
  }
  if (reg_c > FC) {
    const float* sk = skel + (size_t)(n_skel == 1 ? 0 : s) * PIX * 4;
    for (int it = threadIdx.x; it < PIX * 4; it += 256) {
      const int p = it >> 2, c = it & 3;
      regin[((size_t)s * PIX + p) * reg_stride + FC + c] = sk[it];
      mx = max(mx, abs_bits(sk[it]));
    }
  }
  const int npad = reg_stride - reg_c;
  for (int it = threadIdx.x; it < PIX * npad; it += 256) {
    const int p = it / npad, c = it - p * npad;
    regin[((size_t)s * PIX + p) * reg_stride + reg_c + c] = 0.f;
  }
  if (out_max) publish_abs_max(out_max, mx);
}

// ---------------------------------------------------------------- skeleton encoder
__global__ __launch_bounds__(192) void skeleton_kernel(const float* __restrict__ skel_in,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ bn_scale,
                                                       const float* __restrict__ bn_shift,
                                                       float* __restrict__ out) {
  const int k = blockIdx.x;
  __shared__ float x[132];
  const float* axes = skel_in + (size_t)k * 132;
  const float* rest = axes + 66;
  if (threadIdx.x < 132) {
    const int j = threadIdx.x / 6, e = threadIdx.x % 6;
    x[threadIdx.x] = e < 3 ? axes[3 * j + e] : rest[3 * j + e - 3];
  }
  __syncthreads();
  const int o = threadIdx.x;
  if (o < 144) {
    float acc = 0.f;
    for (int i = 0; i < 132; ++i) acc = fmaf(w[o * 132 + i], x[i], acc);
    acc += bias[o];
    const int c = o / PIX, p = o % PIX;
    out[((size_t)k * PIX + p) * 4 + c] = fmaxf(acc * bn_scale[c] + bn_shift[c], 0.f);
  }
}

// ---------------------------------------------------------------- pool + output conv + decode
// Global average pool + the regressor's final 1x1 convolution (commuted: the pool is linear), one block per
// sample: raw [S,64] (d valid entries, rest 0).
__global__ __launch_bounds__(128) void pool_matvec_kernel(const float* __restrict__ reg_feat, int reg_c, int reg_stride,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          int d, float* __restrict__ raw_out) {
  const int s = blockIdx.x;
  __shared__ float pooled[80];
  const float* f = reg_feat + (size_t)s * PIX * reg_stride;
  if (threadIdx.x < reg_c) {
    float acc = 0.f;
    for (int p = 0; p < PIX; ++p) acc += f[p * reg_stride + threadIdx.x];
    pooled[threadIdx.x] = acc * (1.0f / 36.0f);
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    float v = 0.f;
    if (threadIdx.x < d) {
      float acc = 0.f;
      for (int c = 0; c < reg_c; ++c) acc = fmaf(w[threadIdx.x * reg_c + c], pooled[c], acc);
      v = acc + bias[threadIdx.x];
    }
    raw_out[(size_t)s * 64 + threadIdx.x] = v;
  }
}

// Decoders (lib/models/regressor.py:76-121) + Procrustes alignment (lib/models/model_utils.py:17-54) + world
// transform (lib/models/umetrack_model.py:77-97), one THREAD per sample: the float64 Kabsch chain is serial, so
// 64 samples share a wave instead of one lane of a wave each.
__global__ __launch_bounds__(64) void decode_kernel(HeadArgs a, const float* __restrict__ raw_in, int d,
                                                    float* __restrict__ out_pose) {
  UT_RETURN_IF_INVALID(a);
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= a.n_samples) return;
  const float* raw = raw_in + (size_t)s * 64;
  float* o = out_pose + (size_t)s * 60;
  // joint angles: 20 finger DoF + 2 zero wrist angles
  for (int k = 0; k < 22; ++k) o[k] = k < 20 ? raw[k] : 0.f;
  // sigmas: clamp(softplus(x), 1e-5)
  const int sig0 = (d == 63) ? 42 : 41;
  for (int k = 0; k < 21; ++k) {
    const float x = raw[sig0 + k];
    const float sp = x > 20.f ? x : log1pf(expf(x));
    o[39 + k] = fmaxf(sp, 1e-5f);
  }
  o[38] = (d == 63) ? expf(raw[41]) : 0.f;
  {
    // fixed source points (lib/models/regressor.py:19-47)
    const double k = 0.1, q = 0.1 / 1.4142135623730951;
    const double src[7][3] = {{0, 0, 0}, {k, 0, 0}, {0, k, 0}, {0, 0, k}, {-q, -q, 0}, {-q, 0, -q}, {0, -q, -q}};
    double dst[7][3], ms[3] = {0, 0, 0}, md[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        dst[i][j] = (double)raw[20 + 3 * i + j];
        // the reference holds the source points in fp32
        ms[j] += (double)(float)src[i][j];
        md[j] += dst[i][j];
      }
    for (int j = 0; j < 3; ++j) { ms[j] /= 7.0; md[j] /= 7.0; }
    double h[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) h[r][c] += ((double)(float)src[i][r] - ms[r]) * (dst[i][c] - md[c]);
    double rot[3][3];
    kabsch_rotation(h, rot);
    double xf[16] = {0};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) xf[4 * i + j] = rot[i][j];
      xf[4 * i + 3] = md[i] - (rot[i][0] * ms[0] + rot[i][1] * ms[1] + rot[i][2] * ms[2]);
    }
    xf[15] = 1.0;
    // to world: inverse(cam0 extrinsics) @ xf, mirror x for right hands
    const int r0 = (int)a.sample_range[2 * s];
    double ext[16], einv[16], wxf[16];
    load4(a.extrinsics + (size_t)r0 * 16, ext);
    inv4(ext, einv);
    mul4(einv, xf, wxf);
    if (a.hand_idx[s] == 1)
      for (int i = 0; i < 4; ++i) wxf[4 * i] = -wxf[4 * i];
    for (int i = 0; i < 16; ++i) o[22 + i] = (float)wxf[i];
  }
}

// ---------------------------------------------------------------- launchers
hipError_t launch_validate_desc(const HeadArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(validate_desc_kernel, dim3((a.n_samples + 255) / 256), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_ftl_in(const HeadArgs& a, const HeadBuffers& b, hipStream_t s) {
  hipLaunchKernelGGL(ftl_in_kernel, dim3(a.n_samples), dim3(256), 0, s, a, b);
  return hipGetLastError();
}
hipError_t launch_ftl_out_temporal_in(const HeadArgs& a, const HeadBuffers& b, hipStream_t s) {
  hipLaunchKernelGGL(ftl_out_temporal_in_kernel, dim3(a.n_samples), dim3(256), 0, s, a, b);
  return hipGetLastError();
}
hipError_t launch_temporal_out(const HeadArgs& a, const float* t_out, const float* skel, int n_skel,
                               float* regin, int reg_c, int reg_stride, unsigned* out_max, hipStream_t s) {
  hipLaunchKernelGGL(temporal_out_kernel, dim3(a.n_samples), dim3(256), 0, s, a, t_out, skel, n_skel, regin, reg_c, reg_stride, out_max);
  return hipGetLastError();
}
hipError_t launch_skeleton(const float* skel_in, const float* w, const float* bias, const float* bn_scale,
                           const float* bn_shift, float* out, int n_skel, hipStream_t s) {
  hipLaunchKernelGGL(skeleton_kernel, dim3(n_skel), dim3(192), 0, s, skel_in, w, bias, bn_scale, bn_shift, out);
  return hipGetLastError();
}
hipError_t launch_pool_decode(const HeadArgs& a, const float* reg_feat, int reg_c, int reg_stride, const float* w,
                              const float* bias, int d, float* out_pose, float* out_raw, float* raw_ws, hipStream_t s) {
  float* raw = out_raw ? out_raw : raw_ws;     // [S,64]
  hipLaunchKernelGGL(pool_matvec_kernel, dim3(a.n_samples), dim3(128), 0, s, reg_feat, reg_c, reg_stride, w, bias, d, raw);
  hipLaunchKernelGGL(decode_kernel, dim3((a.n_samples + 63) / 64), dim3(64), 0, s, a, raw, d, out_pose);
  return hipGetLastError();
}

}  // namespace ut

namespace ut {
// temporal memory [slots,36,18] (NHWC) -> [slots,18,36] (the reference's NCHW view), for inspection
__global__ void mem_export_kernel(const float* __restrict__ mem, float* __restrict__ out, int total) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int slot = i / (PIX * MEMC), rem = i - slot * PIX * MEMC;
  const int c = rem / PIX, p = rem - c * PIX;
  out[i] = mem[((size_t)slot * PIX + p) * MEMC + c];
}
hipError_t launch_mem_export(const float* mem, float* out, int slots, hipStream_t s) {
  const int total = slots * PIX * MEMC;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(mem_export_kernel, dim3((total + 255) / 256), dim3(256), 0, s, mem, out, total);
  return hipGetLastError();
}
}  // namespace ut
