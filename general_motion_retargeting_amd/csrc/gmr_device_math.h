// gmr_device_math.h -- FP64 quaternion / SE(3) helpers for the gfx950 IK kernels.
// Quaternions wxyz, tangent order [v; w].  Formulas follow SURVEY.md Appendix A (mink / MuJoCo
// semantics, restated); small-angle series keep every coefficient accurate to ~1e-16.
#pragma once
#include <hip/hip_runtime.h>

namespace gmr {

struct d3 { double x, y, z; };
struct d4 { double w, x, y, z; };

__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 operator*(double s, d3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ d3 cross(d3 a, d3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

__device__ __forceinline__ d4 qmul(d4 a, d4 b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
          a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
          a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
// 1/sqrt(x) for a normal-range positive x (pivots of an SPD matrix, squared quaternion norms):
// v_rsq_f64 seed (24 bits) + ONE third-order step y (1 + e/2 + 3 e^2 / 8), e = 1 - x y^2: 1.7e-16 relative error measured
// over 2^20 inputs (two Newton steps: 2.4e-16; tools/micro/rsq_check.hip) with four instead of six dependent
// operations -- the pivots of the factorisations wait on this chain.  No range scaling / special cases: NaN and
// non-positive inputs stay NaN (callers test).
__device__ __forceinline__ double fast_rsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-x, y * y, 1.0);
  return fma(y * e, fma(0.375, e, 0.5), y);
}

__device__ __forceinline__ d4 qconj(d4 a) { return {a.w, -a.x, -a.y, -a.z}; }
__device__ __forceinline__ d4 qnormalize(d4 a) {
  double n2 = a.w * a.w + a.x * a.x + a.y * a.y + a.z * a.z;
  double s = fast_rsqrt(n2);
  return {a.w * s, a.x * s, a.y * s, a.z * s};
}
// rotate v by unit quaternion q: v + 2 w (u x v) + 2 u x (u x v)
__device__ __forceinline__ d3 qrot(d4 q, d3 v) {
  d3 u = {q.x, q.y, q.z};
  d3 t = 2.0 * cross(u, v);
  return v + q.w * t + cross(u, t);
}
// rotate v by the inverse of unit quaternion q
__device__ __forceinline__ d3 qrot_inv(d4 q, d3 v) {
  d3 u = {-q.x, -q.y, -q.z};
  d3 t = 2.0 * cross(u, v);
  return v + q.w * t + cross(u, t);
}
__device__ __forceinline__ d4 axis_angle(d3 axis, double angle) {
  double s, c;
  sincos(0.5 * angle, &s, &c);
  return {c, axis.x * s, axis.y * s, axis.z * s};
}

// SO3 log of a unit quaternion (|w| <= pi), mink/jaxlie branch structure (App. A.5)
__device__ __forceinline__ d3 so3_log(d4 q) {
  double n2 = q.x * q.x + q.y * q.y + q.z * q.z;
  double f;
  if (n2 < 1e-10) {
    f = 2.0 / q.w - 2.0 / 3.0 * n2 / (q.w * q.w * q.w);
  } else {
    double n = sqrt(n2);
    if (fabs(q.w) < 1e-10) f = (q.w > 0.0 ? 1.0 : -1.0) * 3.14159265358979323846 / n;
    else f = 2.0 * atan2(q.w < 0 ? -n : n, fabs(q.w)) / n;
  }
  return {f * q.x, f * q.y, f * q.z};
}

// coefficient a of K^2 in  I - K/2 + a K^2  (V^-1 of SE3.log and Jl^-1 of SO3 share it)
__device__ __forceinline__ double vinv_coef(double t2) {
  if (t2 < 1e-2)
    return 1.0 / 12.0 + t2 * (1.0 / 720.0 + t2 * (1.0 / 30240.0 + t2 * (1.0 / 1209600.0 + t2 / 47900160.0)));
  double t = sqrt(t2), h = 0.5 * t, s, c;
  sincos(h, &s, &c);
  return (1.0 - h * c / s) / t2;
}

// same coefficient, also handing back sin t and cos t (t = |w|) for the Q coefficients of Jl^-1;
// below the series switch they are not needed (set to 0 / 1)
__device__ __forceinline__ double vinv_coef_sc(double t2, double& sin_t, double& cos_t) {
  if (t2 < 1e-2) {
    sin_t = 0.0; cos_t = 1.0;
    return 1.0 / 12.0 + t2 * (1.0 / 720.0 + t2 * (1.0 / 30240.0 + t2 * (1.0 / 1209600.0 + t2 / 47900160.0)));
  }
  double t = sqrt(t2), h = 0.5 * t, s, c;
  sincos(h, &s, &c);
  sin_t = 2.0 * s * c;
  cos_t = 1.0 - 2.0 * s * s;
  return (1.0 - h * c / s) / t2;
}

// 3x3 row-major helpers
struct m3 { double a[9]; };
__device__ __forceinline__ m3 skew(d3 w) {
  m3 K;
  K.a[0] = 0; K.a[1] = -w.z; K.a[2] = w.y;
  K.a[3] = w.z; K.a[4] = 0; K.a[5] = -w.x;
  K.a[6] = -w.y; K.a[7] = w.x; K.a[8] = 0;
  return K;
}
__device__ __forceinline__ m3 mmul(const m3& A, const m3& B) {
  m3 C;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      C.a[3 * i + j] = A.a[3 * i] * B.a[j] + A.a[3 * i + 1] * B.a[3 + j] + A.a[3 * i + 2] * B.a[6 + j];
  return C;
}
__device__ __forceinline__ d3 mvec(const m3& A, d3 v) {
  return {A.a[0] * v.x + A.a[1] * v.y + A.a[2] * v.z, A.a[3] * v.x + A.a[4] * v.y + A.a[5] * v.z,
          A.a[6] * v.x + A.a[7] * v.y + A.a[8] * v.z};
}

// e = log(T_wb^-1 T_wt) = [V^-1(w) p_bt ; w]; aux = {a, sin|w|, cos|w|} is reused by se3_jlinv_aux
__device__ __forceinline__ void se3_log_rel(d3 pb, d4 qb, d3 pt, d4 qt, double e[6], double aux[3]) {
  d4 qbt = qmul(qconj(qb), qt);
  d3 pbt = qrot_inv(qb, pt - pb);
  d3 w = so3_log(qbt);
  double t2 = dot(w, w);
  double a = vinv_coef_sc(t2, aux[1], aux[2]);
  aux[0] = a;
  // V^-1 p = p - 0.5 w x p + a w x (w x p)
  d3 wp = cross(w, pbt);
  d3 wwp = cross(w, wp);
  d3 v = pbt - 0.5 * wp + a * wwp;
  e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = w.x; e[4] = w.y; e[5] = w.z;
}

// Jl^-1(e) = [[A, B], [0, A]], B = -A Q A (Barfoot 7.86b; identity when |w|^2 < 1e-10, mink) with the
// skew products of Q collapsed by [a]x[b]x = b a^T - (a.b) I
// (s = w.rho, n = w x rho, m = w x n):
//   Q = 1/2 [rho]x + c1 (rho w^T + w rho^T) - 2 c1 s I - (c1 + c2) s [w]x - c2 [m]x - 2 c4 s (w w^T - t^2 I)
//   A = (1 - a t^2) I - 1/2 [w]x + a w w^T
// two 3x3 products instead of nine; aux = {a, sin t, cos t} from se3_log_rel.
__device__ __forceinline__ void se3_jlinv_aux(const double e[6], const double aux[3], m3& A, m3& B) {
  d3 rho = {e[0], e[1], e[2]}, w = {e[3], e[4], e[5]};
  double t2 = dot(w, w);
#pragma unroll
  for (int i = 0; i < 9; i++) { A.a[i] = 0.0; B.a[i] = 0.0; }
  A.a[0] = A.a[4] = A.a[8] = 1.0;
  if (t2 < 1e-10) return;
  const double a = aux[0];
  double c1, c2, c3;
  if (t2 < 1e-2) {
    c1 = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0 - t2 * t2 * t2 / 362880.0;
    c2 = -1.0 / 24.0 + t2 / 720.0 - t2 * t2 / 40320.0 + t2 * t2 * t2 / 3628800.0;
    c3 = -1.0 / 120.0 + t2 / 5040.0 - t2 * t2 / 362880.0 + t2 * t2 * t2 / 39916800.0;
  } else {
    double t = sqrt(t2), sn = aux[1], cs = aux[2];
    double it2 = 1.0 / t2, it = 1.0 / t;
    c1 = (t - sn) * it2 * it;
    c2 = (1.0 - 0.5 * t2 - cs) * it2 * it2;
    c3 = (t - sn - t2 * t / 6.0) * it2 * it2 * it;
  }
  const double c4 = -0.5 * (c2 - 3.0 * c3);
  const double s = dot(w, rho);
  d3 n = cross(w, rho), m = cross(w, n);
  const double wv[3] = {w.x, w.y, w.z}, rv[3] = {rho.x, rho.y, rho.z};
  // skew part of Q: 1/2 rho - (c1 + c2) s w - c2 m
  const double k1 = (c1 + c2) * s;
  d3 sq = {0.5 * rho.x - k1 * w.x - c2 * m.x, 0.5 * rho.y - k1 * w.y - c2 * m.y, 0.5 * rho.z - k1 * w.z - c2 * m.z};
  m3 Q = skew(sq);
  const double dq = -2.0 * c1 * s + 2.0 * c4 * s * t2, kw = -2.0 * c4 * s;
  m3 Am = skew(d3{-0.5 * w.x, -0.5 * w.y, -0.5 * w.z});
  const double da = 1.0 - a * t2;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      Q.a[3 * i + j] += c1 * (rv[i] * wv[j] + wv[i] * rv[j]) + kw * wv[i] * wv[j] + (i == j ? dq : 0.0);
      Am.a[3 * i + j] += a * wv[i] * wv[j] + (i == j ? da : 0.0);
    }
  m3 AQA = mmul(mmul(Am, Q), Am);
#pragma unroll
  for (int i = 0; i < 9; i++) { A.a[i] = Am.a[i]; B.a[i] = -AQA.a[i]; }
}

// ---- lean variants for the hot loop -----------------------------------------------------------------------------
// The residual of one task needs log(q) of a UNIT quaternion q = (w, v): with n = |v| the half angle is
// h = atan2(n, |w|), so sin h = n and cos h = |w| are already there -- the closed forms of V^-1 and Jl^-1 need no
// further sin / cos / sqrt, and one reciprocal (1 / t) serves all divisions.  Same functions and the same series
// switches as se3_log_rel / se3_jlinv_aux above (mink's formulas, App. A.5); values agree to a few ulp.

// 1 / x for a normal-range x: v_rcp_f64 seed + two Newton steps (~1 ulp)
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
}

// sin and cos for |x| up to a few hundred (joint angles, half angles): Cody-Waite reduction by pi/2 in two pieces,
// fdlibm's kernel polynomials on [-pi/4, pi/4] (< 1 ulp there); no large-argument path, no tables
__device__ __forceinline__ void sincos_small(double x, double* sn, double* cs) {
  const double k = rint(x * 6.36619772367581382433e-01);              // 2 / pi
  double r = fma(-k, 1.57079632673412561417e+00, x);                  // first 33 bits of pi / 2: exact product
  r = fma(-k, 6.07710050650619224932e-11, r);
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                                 2.75573137070700676789e-06), -1.98412698298579493134e-04),
                               8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                                 -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                               -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double s = fma(r * z, ps, r);
  const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int q = (int)k;
  const double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
  *sn = (q & 2) ? -ss : ss;
  *cs = ((q + 1) & 2) ? -cc : cc;
}

// atan2(y, x) for y >= 0, x >= 0, not both zero (sin and cos of a half angle): the quotient of the smaller by the
// larger is reduced once more by tan(pi/8) -- (a - b) / (a + b) is the tangent of (angle - pi/4) -- so that ONE
// division feeds fdlibm's atan kernel for |z| < 7/16 (two parallel Horner chains, < 1 ulp there).  ~35 instructions
// and a short dependent chain instead of the library's ~110.
__device__ __forceinline__ double atan2_q1(double y, double x) {
  const bool swap = y > x;
  const double a = swap ? x : y, b = swap ? y : x;            // a <= b: angle' = atan(a / b) in [0, pi/4]
  const bool red = a > 0.41421356237309503 * b;
  const double num = red ? a - b : a, den = red ? a + b : b;
  const double z = num * fast_rcp(den);
  const double z2 = z * z, w = z2 * z2;
  const double s1 = z2 * fma(w, fma(w, fma(w, fma(w, fma(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02),
                                                  6.66107313738753120669e-02), 9.09088713343650656196e-02),
                                   1.42857142725034663711e-01), 3.33333333333329318027e-01);
  const double s2 = w * fma(w, fma(w, fma(w, fma(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02),
                                          -7.69187620504482999495e-02), -1.11111104054623557880e-01),
                            -1.99999999998764832476e-01);
  double r = z - z * (s1 + s2);
  if (red) r += 0.78539816339744830962;
  return swap ? 1.57079632679489661923 - r : r;
}

// e = log(T_wb^-1 T_wt) = [V^-1(w) p_bt ; w]; aux = {a, sin t, cos t, t, 1/t} (t = |w|) for se3_jlinv_aux5
__device__ __forceinline__ void se3_log_rel5(d3 pb, d4 qb, d3 pt, d4 qt, double e[6], double aux[5]) {
  const d4 q = qmul(qconj(qb), qt);
  const d3 pbt = qrot_inv(qb, pt - pb);
  // SO3 log, mink / jaxlie branch structure (App. A.5)
  const double n2 = q.x * q.x + q.y * q.y + q.z * q.z;
  const double cw = fabs(q.w);
  double f, sh = 0.0, ch = 1.0, inv_sh = 0.0, half = 0.0;   // half angle h, sin h, cos h, 1 / sin h
  if (n2 < 1e-10) {
    f = 2.0 / q.w - 2.0 / 3.0 * n2 / (q.w * q.w * q.w);
  } else {
    const double inv_n = fast_rsqrt(n2);
    sh = n2 * inv_n; ch = cw; inv_sh = inv_n;
    half = atan2_q1(sh, ch);
    bool neg = q.w < 0.0;
    if (cw < 1e-10) {           // mink sets the angle to pi exactly here: sin(pi/2), cos(pi/2) as libm returns them
      half = 1.57079632679489661923; sh = 1.0; ch = 6.123233995736766e-17; inv_sh = 1.0;
      neg = !(q.w > 0.0);
    }
    f = 2.0 * half * inv_n;
    if (neg) f = -f;
  }
  const d3 w = {f * q.x, f * q.y, f * q.z};
  const double t2 = dot(w, w);
  double a;
  if (t2 < 1e-2) {
    aux[1] = 0.0; aux[2] = 1.0; aux[3] = 0.0; aux[4] = 0.0;
    a = 1.0 / 12.0 + t2 * (1.0 / 720.0 + t2 * (1.0 / 30240.0 + t2 * (1.0 / 1209600.0 + t2 / 47900160.0)));
  } else {                      // t >= 0.1: the main branch above was taken
    const double t = 2.0 * half, inv_t = fast_rcp(t);
    aux[1] = 2.0 * sh * ch;
    aux[2] = 1.0 - 2.0 * sh * sh;
    aux[3] = t; aux[4] = inv_t;
    a = (sh - half * ch) * inv_sh * (inv_t * inv_t);      // (1 - (t/2) cot(t/2)) / t^2
  }
  aux[0] = a;
  // V^-1 p = p - 0.5 w x p + a w x (w x p)
  const d3 wp = cross(w, pbt);
  const d3 wwp = cross(w, wp);
  const d3 v = pbt - 0.5 * wp + a * wwp;
  e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = w.x; e[4] = w.y; e[5] = w.z;
}

// Jl^-1(e) = [[A, B], [0, A]] as se3_jlinv_aux, with t and 1/t handed over by se3_log_rel5 (no sqrt, no division)
__device__ __forceinline__ void se3_jlinv_aux5(const double e[6], const double aux[5], m3& A, m3& B) {
  d3 rho = {e[0], e[1], e[2]}, w = {e[3], e[4], e[5]};
  double t2 = dot(w, w);
#pragma unroll
  for (int i = 0; i < 9; i++) { A.a[i] = 0.0; B.a[i] = 0.0; }
  A.a[0] = A.a[4] = A.a[8] = 1.0;
  if (t2 < 1e-10) return;
  const double a = aux[0];
  double c1, c2, c3;
  if (t2 < 1e-2) {
    c1 = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0 - t2 * t2 * t2 / 362880.0;
    c2 = -1.0 / 24.0 + t2 / 720.0 - t2 * t2 / 40320.0 + t2 * t2 * t2 / 3628800.0;
    c3 = -1.0 / 120.0 + t2 / 5040.0 - t2 * t2 / 362880.0 + t2 * t2 * t2 / 39916800.0;
  } else {
    const double t = aux[3], it = aux[4], sn = aux[1], cs = aux[2];
    const double it2 = it * it;
    c1 = (t - sn) * it2 * it;
    c2 = (1.0 - 0.5 * t2 - cs) * it2 * it2;
    c3 = (t - sn - t2 * t / 6.0) * it2 * it2 * it;
  }
  const double c4 = -0.5 * (c2 - 3.0 * c3);
  const double s = dot(w, rho);
  d3 n = cross(w, rho), m = cross(w, n);
  const double wv[3] = {w.x, w.y, w.z}, rv[3] = {rho.x, rho.y, rho.z};
  const double k1 = (c1 + c2) * s;
  d3 sq = {0.5 * rho.x - k1 * w.x - c2 * m.x, 0.5 * rho.y - k1 * w.y - c2 * m.y, 0.5 * rho.z - k1 * w.z - c2 * m.z};
  m3 Q = skew(sq);
  const double dq = -2.0 * c1 * s + 2.0 * c4 * s * t2, kw = -2.0 * c4 * s;
  m3 Am = skew(d3{-0.5 * w.x, -0.5 * w.y, -0.5 * w.z});
  const double da = 1.0 - a * t2;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      Q.a[3 * i + j] += c1 * (rv[i] * wv[j] + wv[i] * rv[j]) + kw * wv[i] * wv[j] + (i == j ? dq : 0.0);
      Am.a[3 * i + j] += a * wv[i] * wv[j] + (i == j ? da : 0.0);
    }
  m3 AQA = mmul(mmul(Am, Q), Am);
#pragma unroll
  for (int i = 0; i < 9; i++) { A.a[i] = Am.a[i]; B.a[i] = -AQA.a[i]; }
}

// Column j of the same blocks, for three lanes per task (lane = 16 j + task): Ac = A e_j, Bc = B e_j with
// A = skew(-w/2) + a w w^T + (1 - a t^2) I and B = -A Q A, Q = skew(sq) + c1 (rho w^T + w rho^T) + kw w w^T + dq I as in
// se3_jlinv_aux5 -- but no matrix is ever formed: skew(u) x = u x x and (u v^T) x = u (v . x), so a column of B is
// three matrix-vector products written as cross and dot products.  About half the instructions of the full blocks per
// lane; values agree with them to rounding.
__device__ __forceinline__ void se3_jlinv_col5(const double e[6], const double aux[5], int j, double Ac[3], double Bc[3]) {
  const d3 rho = {e[0], e[1], e[2]}, w = {e[3], e[4], e[5]};
  const d3 ej = {j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0};
  const double t2 = dot(w, w);
  Ac[0] = ej.x; Ac[1] = ej.y; Ac[2] = ej.z;
  Bc[0] = Bc[1] = Bc[2] = 0.0;
  if (t2 < 1e-10) return;
  const double a = aux[0];
  double c1, c2, c3;
  if (t2 < 1e-2) {
    c1 = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0 - t2 * t2 * t2 / 362880.0;
    c2 = -1.0 / 24.0 + t2 / 720.0 - t2 * t2 / 40320.0 + t2 * t2 * t2 / 3628800.0;
    c3 = -1.0 / 120.0 + t2 / 5040.0 - t2 * t2 / 362880.0 + t2 * t2 * t2 / 39916800.0;
  } else {
    const double t = aux[3], it = aux[4], sn = aux[1], cs = aux[2];
    const double it2 = it * it;
    c1 = (t - sn) * it2 * it;
    c2 = (1.0 - 0.5 * t2 - cs) * it2 * it2;
    c3 = (t - sn - t2 * t / 6.0) * it2 * it2 * it;
  }
  const double c4 = -0.5 * (c2 - 3.0 * c3);
  const double s = dot(w, rho);
  const d3 n = cross(w, rho), m = cross(w, n);
  const double k1 = (c1 + c2) * s;
  const d3 sq = {0.5 * rho.x - k1 * w.x - c2 * m.x, 0.5 * rho.y - k1 * w.y - c2 * m.y, 0.5 * rho.z - k1 * w.z - c2 * m.z};
  const double dq = -2.0 * c1 * s + 2.0 * c4 * s * t2, kw = -2.0 * c4 * s;
  const double da = 1.0 - a * t2;
  const d3 hw = {-0.5 * w.x, -0.5 * w.y, -0.5 * w.z};
  // v = A e_j
  const double awj = a * dot(w, ej);
  const d3 hx = cross(hw, ej);
  const d3 v = {hx.x + awj * w.x + da * ej.x, hx.y + awj * w.y + da * ej.y, hx.z + awj * w.z + da * ej.z};
  // y = Q v
  const double wv = dot(w, v), rv = dot(rho, v);
  const double kr = c1 * wv, kq = c1 * rv + kw * wv;
  const d3 sx = cross(sq, v);
  const d3 y = {sx.x + kr * rho.x + kq * w.x + dq * v.x, sx.y + kr * rho.y + kq * w.y + dq * v.y, sx.z + kr * rho.z + kq * w.z + dq * v.z};
  // z = A y
  const double awy = a * dot(w, y);
  const d3 hy = cross(hw, y);
  Ac[0] = v.x; Ac[1] = v.y; Ac[2] = v.z;
  Bc[0] = -(hy.x + awy * w.x + da * y.x); Bc[1] = -(hy.y + awy * w.y + da * y.y); Bc[2] = -(hy.z + awy * w.z + da * y.z);
}

// The lane id as a value the optimiser cannot see through.  Every predicate on the lane id (lane < nb, lane == pivot, ...)
// is invariant for the whole kernel, so LLVM computes each ONCE at kernel entry and keeps its 64-bit mask in an SGPR
// pair; a fully inlined frame loop has more than a hundred of them, they spill to VGPR lanes (v_writelane) and every
// use pays two v_readlane -- a fifth of the vector instructions of the latency kernel's solve loop.  A phase that
// starts from fresh_lane() recomputes the few predicates it needs (one v_cmp each) and lets them die at its end.
// GMR_NO_FRESH_LANE restores the hoisted form (A/B builds).
__device__ __forceinline__ int fresh_lane(int lane) {
#ifndef GMR_NO_FRESH_LANE
  asm volatile("" : "+v"(lane));
#endif
  return lane;
}

// wave64 butterfly reductions (deterministic, every lane gets the result)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}

// Reductions over the 16-lane DPP rows (v_mov_b32 row_shr: a few cycles per step instead of a 70-cycle
// ds_bpermute round trip).  The IK kernel reduces over tasks (<= 16: row 0) and over dofs (<= 48: rows 0..2).
__device__ __forceinline__ double dpp_row_shr(double v, double fill, int n) {   // lane i <- lane i - n of its row, `fill` if none
  union { double d; int i[2]; } a, f, r;
  a.d = v; f.d = fill;
  // dpp_ctrl 0x110 | n = row_shr:n; bound_ctrl = false keeps `old` (= fill) where the source lane does not exist
  switch (n) {
    case 1: r.i[0] = __builtin_amdgcn_update_dpp(f.i[0], a.i[0], 0x111, 0xf, 0xf, false); r.i[1] = __builtin_amdgcn_update_dpp(f.i[1], a.i[1], 0x111, 0xf, 0xf, false); break;
    case 2: r.i[0] = __builtin_amdgcn_update_dpp(f.i[0], a.i[0], 0x112, 0xf, 0xf, false); r.i[1] = __builtin_amdgcn_update_dpp(f.i[1], a.i[1], 0x112, 0xf, 0xf, false); break;
    case 4: r.i[0] = __builtin_amdgcn_update_dpp(f.i[0], a.i[0], 0x114, 0xf, 0xf, false); r.i[1] = __builtin_amdgcn_update_dpp(f.i[1], a.i[1], 0x114, 0xf, 0xf, false); break;
    default: r.i[0] = __builtin_amdgcn_update_dpp(f.i[0], a.i[0], 0x118, 0xf, 0xf, false); r.i[1] = __builtin_amdgcn_update_dpp(f.i[1], a.i[1], 0x118, 0xf, 0xf, false); break;
  }
  return r.d;
}
// value of the neighbouring lane (lane ^ 1): DPP quad_perm [1, 0, 3, 2] (every lane has a source: no `old` operand,
// so no register copy in front of the DPP move)
__device__ __forceinline__ double dpp_swap_pairs(double v) {
  union { double d; int i[2]; } a, r;
  a.d = v;
  r.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], 0xB1, 0xf, 0xf, true);
  r.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], 0xB1, 0xf, 0xf, true);
  return r.d;
}
// value of lane k of the own 16-lane row, in every lane of that row: DPP row_newbcast:k.  gfx90a+ executes this ONE
// control on 64-bit operands (v_mov_b64_dpp), so a double is broadcast by a single instruction: four independent
// broadcasts (one per row), the result stays in a VGPR.  (As two 32-bit update_dpp with a tied `old` operand it was two
// register copies + two DPP moves: 45 % of the tree solver's vector instructions were moves.)
#define GMR_ROW_BCAST_CASE(K) case K: y = __builtin_amdgcn_update_dpp(x, x, 0x150 + K, 0xf, 0xf, true); break;
__device__ __forceinline__ double row_bcast_d(double v, int k) {
  const long long x = __builtin_bit_cast(long long, v);
  long long y = x;
  switch (k) {
    GMR_ROW_BCAST_CASE(0) GMR_ROW_BCAST_CASE(1) GMR_ROW_BCAST_CASE(2) GMR_ROW_BCAST_CASE(3)
    GMR_ROW_BCAST_CASE(4) GMR_ROW_BCAST_CASE(5) GMR_ROW_BCAST_CASE(6) GMR_ROW_BCAST_CASE(7)
    GMR_ROW_BCAST_CASE(8) GMR_ROW_BCAST_CASE(9) GMR_ROW_BCAST_CASE(10) GMR_ROW_BCAST_CASE(11)
    GMR_ROW_BCAST_CASE(12) GMR_ROW_BCAST_CASE(13) GMR_ROW_BCAST_CASE(14) GMR_ROW_BCAST_CASE(15)
  }
  return __builtin_bit_cast(double, y);
}
#undef GMR_ROW_BCAST_CASE
// sum of lanes 0..15 (lanes 16..63 must hold 0 or are ignored), the same value in every lane
__device__ __forceinline__ double row0_sum(double v) {
  v += dpp_row_shr(v, 0.0, 1);
  v += dpp_row_shr(v, 0.0, 2);
  v += dpp_row_shr(v, 0.0, 4);
  v += dpp_row_shr(v, 0.0, 8);
  union { double d; int i[2]; } a, r;
  a.d = v;
  r.i[0] = __builtin_amdgcn_readlane(a.i[0], 15);
  r.i[1] = __builtin_amdgcn_readlane(a.i[1], 15);
  return r.d;
}
// min over lanes 0..15
__device__ __forceinline__ double row0_min(double v) {
  v = fmin(v, dpp_row_shr(v, v, 1));
  v = fmin(v, dpp_row_shr(v, v, 2));
  v = fmin(v, dpp_row_shr(v, v, 4));
  v = fmin(v, dpp_row_shr(v, v, 8));
  union { double d; int i[2]; } a, r;
  a.d = v;
  r.i[0] = __builtin_amdgcn_readlane(a.i[0], 15);
  r.i[1] = __builtin_amdgcn_readlane(a.i[1], 15);
  return r.d;
}
// max over lanes 0..47
__device__ __forceinline__ double rows3_max(double v) {
  v = fmax(v, dpp_row_shr(v, v, 1));
  v = fmax(v, dpp_row_shr(v, v, 2));
  v = fmax(v, dpp_row_shr(v, v, 4));
  v = fmax(v, dpp_row_shr(v, v, 8));
  union { double d; int i[2]; } a, r0, r1, r2;
  a.d = v;
  r0.i[0] = __builtin_amdgcn_readlane(a.i[0], 15); r0.i[1] = __builtin_amdgcn_readlane(a.i[1], 15);
  r1.i[0] = __builtin_amdgcn_readlane(a.i[0], 31); r1.i[1] = __builtin_amdgcn_readlane(a.i[1], 31);
  r2.i[0] = __builtin_amdgcn_readlane(a.i[0], 47); r2.i[1] = __builtin_amdgcn_readlane(a.i[1], 47);
  return fmax(fmax(r0.d, r1.d), r2.d);
}

}  // namespace gmr
