// Small dense fp64 helpers shared by the head kernels; host+device so that they can be unit
// tested (and sanitised) in a plain CPU build (tests/cpu_math_harness.cpp).
#pragma once
#include <math.h>
#ifndef __HIPCC__
#define UT_HD
#else
#define UT_HD __host__ __device__
#endif

namespace ut {

// ---------------------------------------------------------------- small dense helpers (fp64)
UT_HD inline bool inv4(const double* a, double* out) {
  // Gauss-Jordan with partial pivoting on [a | I].  Every index is a compile-time constant after unrolling (the
  // pivot row is swapped in by predicated exchanges), so on the GPU the 4x8 tableau lives in registers and not
  // in scratch memory.
  double m[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { m[i][j] = a[4 * i + j]; m[i][4 + j] = (i == j) ? 1.0 : 0.0; }
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    double best = fabs(m[c][c]);
#pragma unroll
    for (int r = c + 1; r < 4; ++r)
      if (fabs(m[r][c]) > best) { best = fabs(m[r][c]); piv = r; }
    if (best == 0.0) ok = false;
#pragma unroll
    for (int r = c + 1; r < 4; ++r) {
      const bool sw = (piv == r);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const double x = m[c][j], y = m[r][j];
        m[c][j] = sw ? y : x;
        m[r][j] = sw ? x : y;
      }
    }
    const double d = 1.0 / m[c][c];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[c][j] *= d;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r != c) {
        const double f = m[r][c];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j];
      }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) out[4 * i + j] = m[i][4 + j];
  return ok;
}

UT_HD inline void mul4(const double* a, const double* b, double* c) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * k + j];
      c[4 * i + j] = s;
    }
}

UT_HD inline void load4(const float* p, double* m) {
  for (int i = 0; i < 16; ++i) m[i] = (double)p[i];
}

// ---------------------------------------------------------------- Procrustes (fp64)
UT_HD inline void jacobi_eig3(double a[3][3], double v[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) v[i][j] = (i == j);
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    // converged: the off-diagonal mass is below 1e-40 of the diagonal's, further rotations are exact identities
    // in double precision (Jacobi converges quadratically: 1e-3 -> 1e-6 -> 1e-12 -> 1e-24 -> 1e-48)
    if (off < 1e-300 || off < 1e-40 * (fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]))) break;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
        for (int k = 0; k < 3; ++k) {           // A <- A J
          double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - sn * akq;
          a[k][q] = sn * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {           // A <- J^T A
          double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - sn * aqk;
          a[q][k] = sn * apk + c * aqk;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - sn * vkq;
          v[k][q] = sn * vkp + c * vkq;
        }
      }
  }
}

// R = V diag(1,1,det(V U^T)) U^T for H = U S V^T (lib/models/model_utils.py:40-49).  Only the two
// leading left singular vectors are needed: det * v3 u3^T does not depend on the sign of u3, so
// u3 := u1 x u2 (det U = +1) and the sign is det V.
UT_HD inline void kabsch_rotation(const double h[3][3], double r[3][3]) {
  double ata[3][3], v[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += h[k][i] * h[k][j];
      ata[i][j] = s;
    }
  jacobi_eig3(ata, v);
  // columns sorted by descending eigenvalue: a three-element compare-exchange network on (value, column) pairs,
  // stable like the index sort it replaces (exchange only on strictly greater), all indices static
  double ev[3] = {ata[0][0], ata[1][1], ata[2][2]};
  double vs[3][3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int k = 0; k < 3; ++k) vs[k][c] = v[k][c];
#define UT_CMPX(I, J)                                                         \
  if (ev[J] > ev[I]) {                                                        \
    double t_ = ev[I]; ev[I] = ev[J]; ev[J] = t_;                             \
    for (int k = 0; k < 3; ++k) { t_ = vs[k][I]; vs[k][I] = vs[k][J]; vs[k][J] = t_; } \
  }
  UT_CMPX(0, 1) UT_CMPX(0, 2) UT_CMPX(1, 2)
#undef UT_CMPX
  double u[3][3];
  for (int c = 0; c < 2; ++c) {
    double n2 = 0;
    for (int i = 0; i < 3; ++i) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += h[i][k] * vs[k][c];
      u[i][c] = s;
      n2 += s * s;
    }
    if (c == 1) {   // re-orthogonalise against u1
      double d = u[0][0] * u[0][1] + u[1][0] * u[1][1] + u[2][0] * u[2][1];
      n2 = 0;
      for (int i = 0; i < 3; ++i) { u[i][1] -= d * u[i][0]; n2 += u[i][1] * u[i][1]; }
    }
    double inv = n2 > 0 ? 1.0 / sqrt(n2) : 0.0;
    for (int i = 0; i < 3; ++i) u[i][c] *= inv;
  }
  u[0][2] = u[1][0] * u[2][1] - u[2][0] * u[1][1];
  u[1][2] = u[2][0] * u[0][1] - u[0][0] * u[2][1];
  u[2][2] = u[0][0] * u[1][1] - u[1][0] * u[0][1];
  double detv = vs[0][0] * (vs[1][1] * vs[2][2] - vs[1][2] * vs[2][1])
              - vs[0][1] * (vs[1][0] * vs[2][2] - vs[1][2] * vs[2][0])
              + vs[0][2] * (vs[1][0] * vs[2][1] - vs[1][1] * vs[2][0]);
  double sgn = detv >= 0 ? 1.0 : -1.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      r[i][j] = vs[i][0] * u[j][0] + vs[i][1] * u[j][1] + sgn * vs[i][2] * u[j][2];
}


}  // namespace ut
