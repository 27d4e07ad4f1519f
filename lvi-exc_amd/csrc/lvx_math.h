// lvx_math.h — split R3 + SO3 uniform cubic B-spline evaluation with ANALYTIC Jacobians.
//
// Device math of the MI355X evaluator (FP64).  The functions are __host__ __device__ only so that
// tests/native can check them on the CPU against the oracle's dual numbers; the product library
// (liblvx.so) only ever runs them inside HIP kernels.
//
// Reference behaviour restated (citations into /root/reference/src/lvi_exc/thirdparty/Kontiki/include/kontiki/):
//   basis matrices                      trajectories/spline_base.h:19-29
//   index / interpolation amount        trajectories/spline_base.h:153-157
//   segment dispatch, t-1e-5 retry      trajectories/spline_base.h:194-222 (segment built at :398-424)
//   R3 evaluation                       trajectories/uniform_r3_spline_trajectory.h:36-103
//   SO3 cumulative evaluation           trajectories/uniform_so3_spline_trajectory.h:46-125
//   logq / expq (eps = 1e-16)           math/quaternion_math.h:16-89
// The reference differentiates with ceres::Jet; here derivatives are closed form on the group:
// control point k is perturbed as c_k <- Exp(2 delta_k) (x) c_k, which is exactly
// ceres::EigenQuaternionParameterization::Plus (delta is a half-angle vector, left-multiplied).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define LVX_HD __host__ __device__ __forceinline__
#else
#define LVX_HD inline
#endif

namespace lvx {

struct v3 { double x, y, z; };
struct quat { double x, y, z, w; };   // Eigen coefficient order
struct m3 { double a[9]; };           // row-major

LVX_HD v3 mk(double x, double y, double z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
LVX_HD v3 operator+(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
LVX_HD v3 operator-(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
LVX_HD v3 operator-(v3 a) { return mk(-a.x, -a.y, -a.z); }
LVX_HD v3 operator*(double s, v3 a) { return mk(s * a.x, s * a.y, s * a.z); }
LVX_HD double dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LVX_HD v3 cross(v3 a, v3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
LVX_HD double comp(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

LVX_HD quat mkq(double w, double x, double y, double z) { quat q; q.x = x; q.y = y; q.z = z; q.w = w; return q; }
LVX_HD quat qconj(quat q) { return mkq(q.w, -q.x, -q.y, -q.z); }
LVX_HD quat qmul(quat a, quat b) {  // Hamilton product, Eigen's scalar order
  return mkq(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
             a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
             a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
             a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x);
}
LVX_HD v3 qrot(quat q, v3 v) {  // Eigen _transformVector
  v3 qv = mk(q.x, q.y, q.z);
  v3 uv = cross(qv, v);
  uv = uv + uv;
  return v + q.w * uv + cross(qv, uv);
}
LVX_HD v3 qrot_inv(quat q, v3 v) { return qrot(qconj(q), v); }

LVX_HD m3 m3_identity() { m3 r; for (int i = 0; i < 9; ++i) r.a[i] = 0.0; r.a[0] = r.a[4] = r.a[8] = 1.0; return r; }
LVX_HD m3 m3_zero() { m3 r; for (int i = 0; i < 9; ++i) r.a[i] = 0.0; return r; }
LVX_HD m3 operator*(const m3& A, const m3& B) {
  m3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[3 * i + j] = A.a[3 * i] * B.a[j] + A.a[3 * i + 1] * B.a[3 + j] + A.a[3 * i + 2] * B.a[6 + j];
  return r;
}
LVX_HD m3 mul_t(const m3& A, const m3& B) {  // A * B^T
  m3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[3 * i + j] = A.a[3 * i] * B.a[3 * j] + A.a[3 * i + 1] * B.a[3 * j + 1] + A.a[3 * i + 2] * B.a[3 * j + 2];
  return r;
}
LVX_HD m3 tmul(const m3& A, const m3& B) {  // A^T * B
  m3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[3 * i + j] = A.a[i] * B.a[j] + A.a[3 + i] * B.a[3 + j] + A.a[6 + i] * B.a[6 + j];
  return r;
}
LVX_HD m3 operator+(const m3& A, const m3& B) { m3 r; for (int i = 0; i < 9; ++i) r.a[i] = A.a[i] + B.a[i]; return r; }
LVX_HD m3 operator-(const m3& A, const m3& B) { m3 r; for (int i = 0; i < 9; ++i) r.a[i] = A.a[i] - B.a[i]; return r; }
LVX_HD m3 operator*(double s, const m3& A) { m3 r; for (int i = 0; i < 9; ++i) r.a[i] = s * A.a[i]; return r; }
LVX_HD m3 transpose(const m3& A) { m3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.a[3 * i + j] = A.a[3 * j + i]; return r; }
LVX_HD v3 operator*(const m3& A, v3 v) {
  return mk(A.a[0] * v.x + A.a[1] * v.y + A.a[2] * v.z, A.a[3] * v.x + A.a[4] * v.y + A.a[5] * v.z, A.a[6] * v.x + A.a[7] * v.y + A.a[8] * v.z);
}
LVX_HD v3 tmulv(const m3& A, v3 v) {  // A^T v  (== row vector v^T A, transposed)
  return mk(A.a[0] * v.x + A.a[3] * v.y + A.a[6] * v.z, A.a[1] * v.x + A.a[4] * v.y + A.a[7] * v.z, A.a[2] * v.x + A.a[5] * v.y + A.a[8] * v.z);
}
LVX_HD m3 skew(v3 v) { m3 r; r.a[0] = 0; r.a[1] = -v.z; r.a[2] = v.y; r.a[3] = v.z; r.a[4] = 0; r.a[5] = -v.x; r.a[6] = -v.y; r.a[7] = v.x; r.a[8] = 0; return r; }
// rotation matrix of a (unit) quaternion
LVX_HD m3 rotmat(quat q) {
  const double xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z, xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z, wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
  m3 r;
  r.a[0] = 1.0 - 2.0 * (yy + zz); r.a[1] = 2.0 * (xy - wz); r.a[2] = 2.0 * (xz + wy);
  r.a[3] = 2.0 * (xy + wz); r.a[4] = 1.0 - 2.0 * (xx + zz); r.a[5] = 2.0 * (yz - wx);
  r.a[6] = 2.0 * (xz - wy); r.a[7] = 2.0 * (yz + wx); r.a[8] = 1.0 - 2.0 * (xx + yy);
  return r;
}

// ---------------------------------------------------------------------------------------------
// SO(3)/S^3 Jacobians for a rotation vector phi (angle th = |phi|, may reach (pi, 2pi) because the
// reference's logq does no hemisphere handling: quaternion_math.h:46-52).
//   Jr(phi)    = I - c1 [phi]x + c2 [phi]x^2,  c1 = (1-cos th)/th^2, c2 = (th - sin th)/th^3
//   Jr^-1(phi) = I + 1/2 [phi]x + c3 [phi]x^2, c3 = 1/th^2 - (1+cos th)/(2 th sin th)
// Series below |phi| < 0.05 (truncation < 1e-17).
// ---------------------------------------------------------------------------------------------
LVX_HD m3 so3_Jr(v3 phi) {
  const double t2 = dot(phi, phi);
  double c1, c2;
  if (t2 < 2.5e-3) {
    c1 = 0.5 - t2 * (1.0 / 24.0 - t2 * (1.0 / 720.0 - t2 / 40320.0));
    c2 = 1.0 / 6.0 - t2 * (1.0 / 120.0 - t2 * (1.0 / 5040.0 - t2 / 362880.0));
  } else {
    const double th = sqrt(t2);
    c1 = (1.0 - cos(th)) / t2;
    c2 = (th - sin(th)) / (t2 * th);
  }
  const m3 K = skew(phi);
  return m3_identity() - c1 * K + c2 * (K * K);
}
LVX_HD m3 so3_Jr_inv(v3 phi) {
  const double t2 = dot(phi, phi);
  double c3;
  if (t2 < 2.5e-3) {
    c3 = 1.0 / 12.0 + t2 * (1.0 / 720.0 + t2 * (1.0 / 30240.0 + t2 / 1209600.0));
  } else {
    const double th = sqrt(t2);
    const double h = 0.5 * th;
    c3 = 1.0 / t2 - cos(h) / (2.0 * th * sin(h));   // (1+cos th)/sin th = cot(th/2)
  }
  const m3 K = skew(phi);
  return m3_identity() + 0.5 * K + c3 * (K * K);
}

// quaternion_math.h:16-59 — returns the half-angle vector Omega (pure quaternion part); *ok=false if non-unit
LVX_HD v3 logq_half(quat q, bool* ok) {
  const double qn = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  if (fabs(qn - 1.0) > 1e-5) *ok = false;
  const double v2 = q.x * q.x + q.y * q.y + q.z * q.z;
  double k;
  if (v2 > 1e-16) { const double vn = sqrt(v2); k = atan2(vn, q.w) / vn; }
  else k = 1.0;
  return mk(q.x * k, q.y * k, q.z * k);
}
// quaternion_math.h:62-89 with a pure-quaternion argument (w = 0 => exp(w) = 1)
LVX_HD quat expq_half(v3 v) {
  const double v2 = v.x * v.x + v.y * v.y + v.z * v.z;
  double ka, kv;
  if (v2 > 1e-16) { const double vn = sqrt(v2); ka = cos(vn); kv = sin(vn) / vn; }
  else { ka = 1.0; kv = 1.0; }
  return mkq(ka, kv * v.x, kv * v.y, kv * v.z);
}

// ---------------------------------------------------------------------------------------------
// Knot lookup with the reference's segment semantics.  A residual's spline segment starts at the master
// knot i1 = floor((t_span - t0)/dt) (spline_base.h:387-401) and has t0_seg = t0 + dt*i1; evaluation checks
// t in [t0_seg, t0_seg + (n-3) dt) and otherwise retries t - 1e-5 (:196-203), then recomputes u against the
// SEGMENT origin (:153-157).  n_seg = 4 for single-time spans.  Returns false on std::range_error.
// ---------------------------------------------------------------------------------------------
// c + a * b in TWO roundings.  The evaluator is compiled with FP contraction on, but time arithmetic must round like the reference's separately
// rounded expressions: the segment origin t0_seg = t0 + dt * i1 (spline_base.h:399) and the range bounds t0 + (n - 3) * dt (:48-56) decide which knot
// interval a measurement falls into and its interpolation amount u — fused, t0_seg moves by an ulp of t (6e-14 s at t = 500 s), u by 3e-12, and a
// gyroscope row of a 500 s trajectory by 1e-11 of the family's scale (found by tests/test_gpu_fullsize_oracle.py; invisible at 10 s).
// (An empty asm keeps the product in a register of its own: under -ffp-contract=fast the backend fuses any fmul + fadd it sees, whatever the
// source-level contraction pragmas say.)
LVX_HD double madd_2r(double a, double b, double c) {
  double p = a * b;
#if defined(__AMDGCN__)
  asm volatile("" : "+v"(p));
#elif defined(__x86_64__)
  asm volatile("" : "+x"(p));
#else
  asm volatile("" : "+m"(p));   // any other host (the oracle / host checks build this header with plain g++): through memory, portable
#endif
  return c + p;
}

struct KnotRef { int i0; double u; };   // master index of the first of the four control points; interpolation amount

// x / dt for the knot lookups, whose floor() classifies a measurement into its knot interval and must agree with the division's.
// q = x * (1/dt) is within 2 ulp of the exact quotient, so floor(q) = floor(RN(x/dt)) unless q lies that close to an integer; only those
// lanes divide (FP64 division is a ~12-instruction dependent chain; 1/dt is wave-uniform and hoisted).  The interpolation amount
// u = q - floor(q) may differ from the divided one in its last bit — deterministic, and the same function serves every kernel.
LVX_HD double quot_dt(double x, double dt) {
  const double inv = 1.0 / dt;
  double q = x * inv;
  const double n = rint(q);
  if (fabs(q - n) <= 1e-9 * fmax(1.0, fabs(q))) q = x / dt;
  return q;
}

LVX_HD bool knot_lookup_seg(double t0_seg, double dt, int n_seg, int i1, double t, KnotRef* out) {
  const double tmin = t0_seg, tmax = madd_2r((double)(n_seg - 3), dt, t0_seg);
  double te = t;
  if (!((te >= tmin) && (te < tmax))) {
    te = t - 0.00001;
    if (!((te >= tmin) && (te < tmax))) return false;
  }
  const double s = quot_dt(te - t0_seg, dt);
  const int il = (int)floor(s);
  if ((n_seg < 4) || (il < 0) || (il > (n_seg - 4))) return false;
  out->i0 = i1 + il;
  out->u = s - (double)il;
  return true;
}
// single-time span {{t_span, t_span}} evaluated at t_eval (= t_span + time offset): 4-knot segment
LVX_HD bool knot_lookup(double t0, double dt, int n_knots, double t_span, double t_eval, KnotRef* out) {
  const double tmax_master = madd_2r((double)(n_knots - 3), dt, t0);
  if (n_knots < 4 || t_span < t0 || t_span >= tmax_master) return false;   // CheckTimeSpans, trajectory_estimator.h:102-127
  const int i1 = (int)floor(quot_dt(t_span - t0, dt));
  return knot_lookup_seg(madd_2r(dt, (double)i1, t0), dt, 4, i1, t_eval, out);
}

// ---------------------------------------------------------------------------------------------
// R3: basis weights (uniform_r3_spline_trajectory.h:51-94). p = sum B[j] c_j etc.; d(p)/d(c_j) = B[j] I.
// ---------------------------------------------------------------------------------------------
struct R3Basis { double Bp[4], Bv[4], Ba[4]; };
LVX_HD void r3_basis(double u, double dt, R3Basis* b) {
  const double u2 = u * u, u3 = u2 * u;
  const double di = 1.0 / dt, di2 = di * di;
  // [1 u u2 u3] * M,  M from spline_base.h:19-23
  b->Bp[0] = 1.0 / 6.0 + u * (-3.0 / 6.0) + u2 * (3.0 / 6.0) + u3 * (-1.0 / 6.0);
  b->Bp[1] = 4.0 / 6.0 + u2 * (-6.0 / 6.0) + u3 * (3.0 / 6.0);
  b->Bp[2] = 1.0 / 6.0 + u * (3.0 / 6.0) + u2 * (3.0 / 6.0) + u3 * (-3.0 / 6.0);
  b->Bp[3] = u3 * (1.0 / 6.0);
  const double U1 = di, U2 = di * (2.0 * u), U3 = di * (3.0 * u2);
  b->Bv[0] = U1 * (-3.0 / 6.0) + U2 * (3.0 / 6.0) + U3 * (-1.0 / 6.0);
  b->Bv[1] = U2 * (-6.0 / 6.0) + U3 * (3.0 / 6.0);
  b->Bv[2] = U1 * (3.0 / 6.0) + U2 * (3.0 / 6.0) + U3 * (-3.0 / 6.0);
  b->Bv[3] = U3 * (1.0 / 6.0);
  const double A2 = di2 * 2.0, A3 = di2 * (6.0 * u);
  b->Ba[0] = A2 * (3.0 / 6.0) + A3 * (-1.0 / 6.0);
  b->Ba[1] = A2 * (-6.0 / 6.0) + A3 * (3.0 / 6.0);
  b->Ba[2] = A2 * (3.0 / 6.0) + A3 * (-3.0 / 6.0);
  b->Ba[3] = A3 * (1.0 / 6.0);
}

// ---------------------------------------------------------------------------------------------
// SO3 cumulative spline: value + analytic Jacobians.
//   q = c0 E1 E2 E3,  E_j = expq(B_j Omega_j),  Omega_j = logq(c_{j-1}^* c_j)        (:75-104)
//   w_body = q^* (2 vec(qdot q^*)) q = R3^T R2^T dB1 d1 + R3^T dB2 d2 + dB3 d3,  d_j = 2 Omega_j  (:92-121)
// Jacobians w.r.t. the ceres tangent delta_k of control point k (k = 0..3):
//   q(delta) = q (x) Exp(xi),  xi = sum_k dxi[k] delta_k      (body-frame rotation-vector perturbation)
//   w_body(delta) = w_body + sum_k dw[k] delta_k
// ---------------------------------------------------------------------------------------------
struct So3Eval {
  quat q;
  v3 w_body;
  m3 dxi[4];
  m3 dw[4];
};

template <bool NEED_W, bool NEED_J, bool NEED_DW = (NEED_W && NEED_J)>
LVX_HD bool so3_eval(const quat c[4], double u, double dt, So3Eval* out) {
  const double u2 = u * u, u3 = u2 * u;
  // [1 u u2 u3] * M_cumul (spline_base.h:25-29); B[0] = 1
  double B[4], dB[4];
  B[1] = 5.0 / 6.0 + u * (3.0 / 6.0) + u2 * (-3.0 / 6.0) + u3 * (1.0 / 6.0);
  B[2] = 1.0 / 6.0 + u * (3.0 / 6.0) + u2 * (3.0 / 6.0) + u3 * (-2.0 / 6.0);
  B[3] = u3 * (1.0 / 6.0);
  if (NEED_W) {
    const double di = 1.0 / dt;
    const double U1 = di, U2 = di * (2.0 * u), U3 = di * (3.0 * u2);
    dB[1] = U1 * (3.0 / 6.0) + U2 * (-3.0 / 6.0) + U3 * (1.0 / 6.0);
    dB[2] = U1 * (3.0 / 6.0) + U2 * (3.0 / 6.0) + U3 * (-2.0 / 6.0);
    dB[3] = U3 * (1.0 / 6.0);
  }
  bool ok = true;
  v3 d[4];        // rotation vectors d_j = 2 Omega_j
  quat E[4];
  quat q = c[0];
#pragma unroll
  for (int j = 1; j < 4; ++j) {
    const v3 Om = logq_half(qmul(qconj(c[j - 1]), c[j]), &ok);
    d[j] = 2.0 * Om;
    E[j] = expq_half(B[j] * Om);
    q = qmul(q, E[j]);
  }
  out->q = q;
  if (!NEED_W && !NEED_J) return ok;
  const m3 R2 = rotmat(E[2]), R3 = rotmat(E[3]);
  v3 w1, w2r, w2, w3r;
  if (NEED_W) {
    w1 = dB[1] * d[1];
    w2r = tmulv(R2, w1);            // R2^T w1
    w2 = w2r + dB[2] * d[2];
    w3r = tmulv(R3, w2);            // R3^T w2
    out->w_body = w3r + dB[3] * d[3];
  }
  if (!NEED_J) return ok;
  const m3 R1 = rotmat(E[1]);
  m3 Jri[4], P[4];
#pragma unroll
  for (int j = 1; j < 4; ++j) {
    Jri[j] = so3_Jr_inv(d[j]);
    P[j] = B[j] * so3_Jr(B[j] * d[j]);
  }
  const m3 R3t = transpose(R3);
  const m3 R32t = tmul(R3, transpose(R2));      // R3^T R2^T
  // coefficient of delta d_j in xi
  const m3 T1 = R32t * P[1], T2 = R3t * P[2], T3 = P[3];
  // d xi / d eta_k  (eta_k: right perturbation of control point k)
  m3 Xe[4];
  Xe[0] = tmul(R1 * (R2 * R3), m3_identity()) - mul_t(T1, Jri[1]);
  Xe[1] = T1 * Jri[1] - mul_t(T2, Jri[2]);
  Xe[2] = T2 * Jri[2] - mul_t(T3, Jri[3]);
  Xe[3] = T3 * Jri[3];
  m3 We[4];
  if (NEED_DW) {
    const m3 W1 = dB[1] * R32t;
    const m3 W2 = R3t * (skew(w2r) * P[2] + dB[2] * m3_identity());
    const m3 W3 = skew(w3r) * P[3] + dB[3] * m3_identity();
    We[0] = -1.0 * mul_t(W1, Jri[1]);
    We[1] = W1 * Jri[1] - mul_t(W2, Jri[2]);
    We[2] = W2 * Jri[2] - mul_t(W3, Jri[3]);
    We[3] = W3 * Jri[3];
  }
  // eta_k = R(c_k)^T (2 delta_k)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const m3 Rk = rotmat(c[k]);
    out->dxi[k] = 2.0 * mul_t(Xe[k], Rk);
    if (NEED_DW) out->dw[k] = 2.0 * mul_t(We[k], Rk);
  }
  return ok;
}


// ---------------------------------------------------------------------------------------------
// The same evaluation with the u-independent part hoisted.  Omega_j = logq(c_{j-1}^* c_j) and J_r^-1(2 Omega_j) depend only on two
// neighbouring control points, not on the evaluation time: every measurement of a knot interval (tens of rows) shares them, so the fused
// kernels compute them once per control-point pair into LDS (So3Pre) and each row is left with three sincos:
//   E_j = (cos a, sinc(a) B_j Omega_j), a = B_j |Omega_j|;   J_r(B_j d_j) = I - c1 K + c2 K^2 with theta = 2 a:
//   c1 = (1 - cos theta)/theta^2 = sinc(a)^2 / 2,   c2 = (theta - sin theta)/theta^3 = (1 - sinc(a) cos a) / (4 a^2)   (double-angle forms)
// Same series switches as so3_Jr / expq_half; results agree with so3_eval to rounding (tests/test_host_math.py).
// ---------------------------------------------------------------------------------------------
constexpr double SO3_SMALL_A = 0.8;   // largest |Omega| served by the small-angle polynomials (so3_small_coeffs)
struct So3Pre { v3 Om; double on; m3 Jri; double c3; int ok; };   // Omega, |Omega|, J_r^-1(2 Omega) and its coefficient c3 (J_r^-1 = I + K/2 + c3 K^2), unit-norm check of logq
LVX_HD void so3_pre(quat ca, quat cb, So3Pre* o) {
  bool ok = true;
  o->Om = logq_half(qmul(qconj(ca), cb), &ok);
  o->on = sqrt(dot(o->Om, o->Om));
  o->Jri = so3_Jr_inv(2.0 * o->Om);
  { const double t2 = 4.0 * o->on * o->on;   // same switch as so3_Jr_inv
    if (t2 < 2.5e-3) o->c3 = 1.0 / 12.0 + t2 * (1.0 / 720.0 + t2 * (1.0 / 30240.0 + t2 / 1209600.0));
    else { const double th = sqrt(t2), h = 0.5 * th; o->c3 = 1.0 / t2 - cos(h) / (2.0 * th * sin(h)); } }
  o->ok = ok ? (o->on <= SO3_SMALL_A ? 1 : 2) : 0;   // 2: valid, but beyond the small-angle polynomials of so3_value_pre
}
// The factor angles a = B_j |Omega_j| <= |Omega_j| are small (Omega_j is the half-angle vector between neighbouring control points), so
// Taylor polynomials in a^2 give sin(a)/a, cos(a) and c2 = (theta - sin theta)/theta^3 (theta = 2a) to double precision for |a| <= 0.8
// (first dropped terms: 1.5e-19, 3e-21, 8e-22) — no range reduction, no division, no series / closed-form switch.  Pairs with
// |Omega| > SO3_SMALL_A are marked in the table (So3Pre::ok == 2) and their rows take the exact fallback kernel.
LVX_HD void so3_small_coeffs(double a2, double* kv, double* ka, double* c2) {
  *kv = 1.0 + a2 * (-1.0 / 6.0 + a2 * (1.0 / 120.0 + a2 * (-1.0 / 5040.0 + a2 * (1.0 / 362880.0 + a2 * (-1.0 / 39916800.0 + a2 * (1.0 / 6227020800.0 +
        a2 * (-1.0 / 1307674368000.0 + a2 * (1.0 / 355687428096000.0))))))));
  *ka = 1.0 + a2 * (-0.5 + a2 * (1.0 / 24.0 + a2 * (-1.0 / 720.0 + a2 * (1.0 / 40320.0 + a2 * (-1.0 / 3628800.0 + a2 * (1.0 / 479001600.0 +
        a2 * (-1.0 / 87178291200.0 + a2 * (1.0 / 20922789888000.0 + a2 * (-1.0 / 6402373705728000.0)))))))));
  const double t2 = 4.0 * a2;
  *c2 = 1.0 / 6.0 + t2 * (-1.0 / 120.0 + t2 * (1.0 / 5040.0 + t2 * (-1.0 / 362880.0 + t2 * (1.0 / 39916800.0 + t2 * (-1.0 / 6227020800.0 +
        t2 * (1.0 / 1307674368000.0 + t2 * (-1.0 / 355687428096000.0 + t2 * (1.0 / 121645100408832000.0 + t2 * (-1.0 / 51090942171709440000.0)))))))));
}
// returns 0, or 1 (a pair failed logq's unit-norm check), or 2 (a pair's angle is beyond the small-angle polynomials: exact fallback)
// NEED_XI = false (with NEED_J): only the angular-velocity derivatives dw[] (the gyroscope rows of k_imu_rot, which evaluates twice to halve its live registers)
template <bool NEED_W, bool NEED_J, bool NEED_DW = (NEED_W && NEED_J), bool NEED_XI = NEED_J>
LVX_HD int so3_eval_pre(const quat c[4], const So3Pre* pre, double u, double dt, So3Eval* out) {
  const double u2 = u * u, u3 = u2 * u;
  double B[4], dB[4];
  B[1] = 5.0 / 6.0 + u * (3.0 / 6.0) + u2 * (-3.0 / 6.0) + u3 * (1.0 / 6.0);
  B[2] = 1.0 / 6.0 + u * (3.0 / 6.0) + u2 * (3.0 / 6.0) + u3 * (-2.0 / 6.0);
  B[3] = u3 * (1.0 / 6.0);
  if (NEED_W) {
    const double di = 1.0 / dt;
    const double U1 = di, U2 = di * (2.0 * u), U3 = di * (3.0 * u2);
    dB[1] = U1 * (3.0 / 6.0) + U2 * (-3.0 / 6.0) + U3 * (1.0 / 6.0);
    dB[2] = U1 * (3.0 / 6.0) + U2 * (3.0 / 6.0) + U3 * (-2.0 / 6.0);
    dB[3] = U3 * (1.0 / 6.0);
  }
  int bad = 0;
  v3 d[4];
  quat E[4];
  m3 P[4];
  quat q = c[0];
#pragma unroll
  for (int j = 1; j < 4; ++j) {
    const So3Pre& pj = pre[j - 1];
    bad |= pj.ok == 1 ? 0 : (pj.ok == 0 ? 1 : 2);
    d[j] = 2.0 * pj.Om;
    const v3 v = B[j] * pj.Om;
    const double a = B[j] * pj.on, a2 = a * a;
    double ka, kv, c2;
    so3_small_coeffs(a2, &kv, &ka, &c2);
    E[j] = mkq(ka, kv * v.x, kv * v.y, kv * v.z);
    q = qmul(q, E[j]);
    if (NEED_J) {
      const double c1 = 0.5 * kv * kv;     // (1 - cos theta)/theta^2 with theta = 2 a
      const m3 K = skew(B[j] * d[j]);
      P[j] = B[j] * (m3_identity() - c1 * K + c2 * (K * K));
    }
  }
  out->q = q;
  if (!NEED_W && !NEED_J) return bad;
  const m3 R2 = rotmat(E[2]), R3 = rotmat(E[3]);
  v3 w1, w2r, w2, w3r;
  if (NEED_W) {
    w1 = dB[1] * d[1];
    w2r = tmulv(R2, w1);
    w2 = w2r + dB[2] * d[2];
    w3r = tmulv(R3, w2);
    out->w_body = w3r + dB[3] * d[3];
  }
  if (!NEED_J) return bad;
  const m3 R1 = rotmat(E[1]);
  const m3 R3t = transpose(R3);
  const m3 R32t = tmul(R3, transpose(R2));
  const m3 T1 = R32t * P[1], T2 = R3t * P[2], T3 = P[3];
  m3 Xe[4];
  if (NEED_XI) {
    Xe[0] = tmul(R1 * (R2 * R3), m3_identity()) - mul_t(T1, pre[0].Jri);
    Xe[1] = T1 * pre[0].Jri - mul_t(T2, pre[1].Jri);
    Xe[2] = T2 * pre[1].Jri - mul_t(T3, pre[2].Jri);
    Xe[3] = T3 * pre[2].Jri;
  }
  m3 We[4];
  if (NEED_DW) {
    const m3 W1 = dB[1] * R32t;
    const m3 W2 = R3t * (skew(w2r) * P[2] + dB[2] * m3_identity());
    const m3 W3 = skew(w3r) * P[3] + dB[3] * m3_identity();
    We[0] = -1.0 * mul_t(W1, pre[0].Jri);
    We[1] = W1 * pre[0].Jri - mul_t(W2, pre[1].Jri);
    We[2] = W2 * pre[1].Jri - mul_t(W3, pre[2].Jri);
    We[3] = W3 * pre[2].Jri;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const m3 Rk = rotmat(c[k]);
    if (NEED_XI) out->dxi[k] = 2.0 * mul_t(Xe[k], Rk);
    if (NEED_DW) out->dw[k] = 2.0 * mul_t(We[k], Rk);
  }
  return bad;
}


// ---------------------------------------------------------------------------------------------
// Reverse mode for single-row residuals.  A scalar residual needs, per control point k, only the VECTOR dxi[k]^T g (g = d r / d xi): with
//   u3 = g, u2 = R3 g, u1 = R2 u2, u0 = R1 u1,   a_j = P_j^T u_j,
//   y0 = u0 - Jri1 a1,  y1 = Jri1^T a1 - Jri2 a2,  y2 = Jri2^T a2 - Jri3 a3,  y3 = Jri3^T a3,     dxi[k]^T g = 2 R(c_k) y_k
// (the transposes of Xe[k] in so3_eval), and every product is a rotation by a quaternion or a pair of cross products:
//   J_r(phi)^T v = v + c1 phi x v + c2 phi x (phi x v),   J_r^-1(d) v = v + d x v / 2 + c3 d x (d x v),   (J_r^-1)^T v = v - d x v / 2 + c3 d x (d x v).
// 16 matrix-vector-sized operations instead of ~14 3x3 matrix products, and no 3x3 matrix is kept in registers.
// ---------------------------------------------------------------------------------------------
struct So3Val { quat q; quat E[4]; v3 phi[4]; double c1[4], c2[4]; };   // value, factors E_j, phi_j = B_j d_j and the J_r coefficients (j = 1..3)
// returns 0, or 1 when a control-point pair failed logq's unit-norm check, or 2 when a pair's angle is beyond the small-angle polynomials
LVX_HD int so3_value_pre(const quat c[4], const So3Pre* pre, double u, So3Val* o) {
  const double u2 = u * u, u3 = u2 * u;
  double B[4];
  B[1] = 5.0 / 6.0 + u * (3.0 / 6.0) + u2 * (-3.0 / 6.0) + u3 * (1.0 / 6.0);
  B[2] = 1.0 / 6.0 + u * (3.0 / 6.0) + u2 * (3.0 / 6.0) + u3 * (-2.0 / 6.0);
  B[3] = u3 * (1.0 / 6.0);
  int bad = 0;
  quat q = c[0];
#pragma unroll
  for (int j = 1; j < 4; ++j) {
    const So3Pre& pj = pre[j - 1];
    bad |= pj.ok == 1 ? 0 : (pj.ok == 0 ? 1 : 2);
    const v3 v = B[j] * pj.Om;
    const double a = B[j] * pj.on, a2 = a * a;
    double ka, kv, c2;
    so3_small_coeffs(a2, &kv, &ka, &c2);
    o->E[j] = mkq(ka, kv * v.x, kv * v.y, kv * v.z);
    q = qmul(q, o->E[j]);
    o->phi[j] = (2.0 * B[j]) * pj.Om;
    o->c1[j] = B[j] * (0.5 * kv * kv); o->c2[j] = B[j] * c2;       // P_j = B_j J_r(phi_j) = B_j (I - c1 K + c2 K^2), c1 = (1 - cos theta)/theta^2 = sinc(a)^2 / 2
  }
  o->phi[0] = mk(B[1], B[2], B[3]);           // slot 0 carries the cumulative basis values
  o->q = q;
  return bad;
}
// y[k] = dxi[k]^T g for the four control points
LVX_HD void so3_pullback_pre(const quat c[4], const So3Pre* pre, const So3Val& s, v3 g, v3 y[4]) {
  const double B[4] = {0.0, s.phi[0].x, s.phi[0].y, s.phi[0].z};
  v3 uv[4];
  uv[3] = g; uv[2] = qrot(s.E[3], uv[3]); uv[1] = qrot(s.E[2], uv[2]); uv[0] = qrot(s.E[1], uv[1]);
  v3 a[4], fw[4], bw[4];   // a_j = P_j^T u_j ; fw_j = Jri_j a_j ; bw_j = Jri_j^T a_j
#pragma unroll
  for (int j = 1; j < 4; ++j) {
    const v3 p1 = cross(s.phi[j], uv[j]);
    a[j] = B[j] * uv[j] + s.c1[j] * p1 + s.c2[j] * cross(s.phi[j], p1);
    const v3 d = 2.0 * pre[j - 1].Om;
    const v3 k1 = cross(d, a[j]);
    const v3 k2 = pre[j - 1].c3 * cross(d, k1);
    fw[j] = a[j] + 0.5 * k1 + k2;
    bw[j] = a[j] - 0.5 * k1 + k2;
  }
  const v3 y0 = uv[0] - fw[1], y1 = bw[1] - fw[2], y2 = bw[2] - fw[3], y3 = bw[3];
  y[0] = 2.0 * qrot(c[0], y0); y[1] = 2.0 * qrot(c[1], y1); y[2] = 2.0 * qrot(c[2], y2); y[3] = 2.0 * qrot(c[3], y3);
}

// ---------------------------------------------------------------------------------------------
// Reverse mode for the body angular velocity (the gyroscope rows of k_imu_rot): z[k] = dw[k]^T g without the 3 x 3 blocks W_j / We[k] of so3_eval_pre.
//   w1 = dB1 d1,  w2r = R2^T w1,  w2 = w2r + dB2 d2,  w3r = R3^T w2,  w_body = w3r + dB3 d3                       (value; d_j = 2 Omega_j)
//   b1 = W1^T g = dB1 R2 R3 g,   b2 = W2^T g = P2^T ((R3 g) x w2r) + dB2 R3 g,   b3 = W3^T g = P3^T (g x w3r) + dB3 g      (skew(w)^T v = v x w)
//   z0 = -Jri1 b1,  z1 = Jri1^T b1 - Jri2 b2,  z2 = Jri2^T b2 - Jri3 b3,  z3 = Jri3^T b3,      dw[k]^T g = 2 R(c_k) z_k
// with the same vector forms of P_j^T, J_r^-1 and its transpose as so3_pullback_pre.
// ---------------------------------------------------------------------------------------------
struct So3ValW { So3Val s; double dB[4]; v3 w2r, w3r, w_body; };
LVX_HD int so3_value_w_pre(const quat c[4], const So3Pre* pre, double u, double dt, So3ValW* o) {
  const int bad = so3_value_pre(c, pre, u, &o->s);
  const double u2 = u * u, di = 1.0 / dt;
  const double U1 = di, U2 = di * (2.0 * u), U3 = di * (3.0 * u2);
  o->dB[0] = 0.0;
  o->dB[1] = U1 * (3.0 / 6.0) + U2 * (-3.0 / 6.0) + U3 * (1.0 / 6.0);
  o->dB[2] = U1 * (3.0 / 6.0) + U2 * (3.0 / 6.0) + U3 * (-2.0 / 6.0);
  o->dB[3] = U3 * (1.0 / 6.0);
  const v3 w1 = (2.0 * o->dB[1]) * pre[0].Om;
  o->w2r = qrot_inv(o->s.E[2], w1);
  const v3 w2 = o->w2r + (2.0 * o->dB[2]) * pre[1].Om;
  o->w3r = qrot_inv(o->s.E[3], w2);
  o->w_body = o->w3r + (2.0 * o->dB[3]) * pre[2].Om;
  return bad;
}
LVX_HD void so3_pullback_w_pre(const quat c[4], const So3Pre* pre, const So3ValW& w, v3 g, v3 z[4]) {
  const So3Val& s = w.s;
  const double B[4] = {0.0, s.phi[0].x, s.phi[0].y, s.phi[0].z};
  auto PT = [&](int j, v3 v) { const v3 p1 = cross(s.phi[j], v); return B[j] * v + s.c1[j] * p1 + s.c2[j] * cross(s.phi[j], p1); };
  const v3 h2 = qrot(s.E[3], g);                       // R3 g
  v3 b[4];
  b[1] = w.dB[1] * qrot(s.E[2], h2);
  b[2] = PT(2, cross(h2, w.w2r)) + w.dB[2] * h2;
  b[3] = PT(3, cross(g, w.w3r)) + w.dB[3] * g;
  v3 fw[4], bw[4];
#pragma unroll
  for (int j = 1; j < 4; ++j) {
    const v3 d = 2.0 * pre[j - 1].Om;
    const v3 k1 = cross(d, b[j]);
    const v3 k2 = pre[j - 1].c3 * cross(d, k1);
    fw[j] = b[j] + 0.5 * k1 + k2;
    bw[j] = b[j] - 0.5 * k1 + k2;
  }
  z[0] = 2.0 * qrot(c[0], -1.0 * fw[1]); z[1] = 2.0 * qrot(c[1], bw[1] - fw[2]); z[2] = 2.0 * qrot(c[2], bw[2] - fw[3]); z[3] = 2.0 * qrot(c[3], bw[3]);
}

}  // namespace lvx
