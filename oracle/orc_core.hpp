// oracle/orc_core.hpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the LVI-ExC continuous-time hot path (Kontiki split R3+SO3
// uniform cubic B-spline, sensors, measurement functors) in the reference's own
// arithmetic order, templated on the scalar T (double or a forward-mode dual
// number that plays the role of ceres::Jet).  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may use anything under oracle/.
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
// this path and cannot be compiled here (Eigen, Ceres, PCL, ROS absent), so this
// restatement is pinned only by known-answer identities, derivative checks and
// cross-implementation agreement (tests/).  Third-party pieces (Eigen quaternion
// algebra, ceres::Jet, HuberLoss, EigenQuaternionParameterization) are restated
// from their public semantics.
//
// Citations are into /root/reference/src/lvi_exc/thirdparty/Kontiki/include/ (K/).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace orc {

// ---------------------------------------------------------------------------
// Forward-mode dual number (stands in for ceres::Jet<double, N>).
// ---------------------------------------------------------------------------
template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0) { for (int i = 0; i < N; ++i) v[i] = 0; }
  Jet(double x) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0; }  // NOLINT (implicit like ceres)
  static Jet seed(double x, int k) { Jet j(x); j.v[k] = 1.0; return j; }
};
template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f) { Jet<N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; const double gi = 1.0 / g.a; const double fg = f.a * gi; h.a = fg;
  for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - fg * g.v[i]) * gi; return h; }
template <int N> inline Jet<N>& operator+=(Jet<N>& f, const Jet<N>& g) { f = f + g; return f; }
template <int N> inline Jet<N>& operator-=(Jet<N>& f, const Jet<N>& g) { f = f - g; return f; }
template <int N> inline Jet<N>& operator*=(Jet<N>& f, const Jet<N>& g) { f = f * g; return f; }
template <int N> inline Jet<N> operator+(const Jet<N>& f, double s) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator+(double s, const Jet<N>& f) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, double s) { Jet<N> h = f; h.a -= s; return h; }
template <int N> inline Jet<N> operator-(double s, const Jet<N>& f) { Jet<N> h = -f; h.a += s; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, double s) { Jet<N> h; h.a = f.a * s; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s; return h; }
template <int N> inline Jet<N> operator*(double s, const Jet<N>& f) { return f * s; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, double s) { return f * (1.0 / s); }
template <int N> inline Jet<N> operator/(double s, const Jet<N>& g) { return Jet<N>(s) / g; }
template <int N> inline bool operator>(const Jet<N>& f, double s) { return f.a > s; }
template <int N> inline bool operator<(const Jet<N>& f, double s) { return f.a < s; }
template <int N> inline bool operator>=(const Jet<N>& f, double s) { return f.a >= s; }
template <int N> inline bool operator<=(const Jet<N>& f, double s) { return f.a <= s; }
template <int N> inline bool operator>(const Jet<N>& f, const Jet<N>& g) { return f.a > g.a; }
template <int N> inline bool operator<(const Jet<N>& f, const Jet<N>& g) { return f.a < g.a; }
template <int N> inline bool operator>=(const Jet<N>& f, const Jet<N>& g) { return f.a >= g.a; }

inline double value_of(double x) { return x; }
template <int N> inline double value_of(const Jet<N>& x) { return x.a; }

// scalar functions in the ceres:: namespace the reference uses
inline double orc_sqrt(double x) { return std::sqrt(x); }
inline double orc_sin(double x) { return std::sin(x); }
inline double orc_cos(double x) { return std::cos(x); }
inline double orc_exp(double x) { return std::exp(x); }
inline double orc_abs(double x) { return std::fabs(x); }
inline double orc_atan2(double y, double x) { return std::atan2(y, x); }
inline double orc_pow(double x, int p) { return std::pow(x, static_cast<double>(p)); }
template <int N> inline Jet<N> orc_sqrt(const Jet<N>& f) { Jet<N> h; h.a = std::sqrt(f.a); const double s = 1.0 / (2.0 * h.a); for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s; return h; }
template <int N> inline Jet<N> orc_sin(const Jet<N>& f) { Jet<N> h; h.a = std::sin(f.a); const double c = std::cos(f.a); for (int i = 0; i < N; ++i) h.v[i] = c * f.v[i]; return h; }
template <int N> inline Jet<N> orc_cos(const Jet<N>& f) { Jet<N> h; h.a = std::cos(f.a); const double s = -std::sin(f.a); for (int i = 0; i < N; ++i) h.v[i] = s * f.v[i]; return h; }
template <int N> inline Jet<N> orc_exp(const Jet<N>& f) { Jet<N> h; h.a = std::exp(f.a); for (int i = 0; i < N; ++i) h.v[i] = h.a * f.v[i]; return h; }
template <int N> inline Jet<N> orc_abs(const Jet<N>& f) { return f.a < 0.0 ? -f : f; }
template <int N> inline Jet<N> orc_atan2(const Jet<N>& g, const Jet<N>& f) {  // atan2(g, f)
  Jet<N> h; h.a = std::atan2(g.a, f.a); const double t = 1.0 / (f.a * f.a + g.a * g.a);
  for (int i = 0; i < N; ++i) h.v[i] = t * (-g.a * f.v[i] + f.a * g.v[i]); return h; }
template <int N> inline Jet<N> orc_pow(const Jet<N>& f, int p) {  // ceres pow(Jet, double)
  Jet<N> h; h.a = std::pow(f.a, static_cast<double>(p)); const double d = p * std::pow(f.a, static_cast<double>(p - 1));
  for (int i = 0; i < N; ++i) h.v[i] = d * f.v[i]; return h; }

// ---------------------------------------------------------------------------
// Eigen-semantics 3-vector and quaternion (coefficient storage x,y,z,w).
// ---------------------------------------------------------------------------
template <class T> struct V3 {
  T x, y, z;
  V3() : x(T(0)), y(T(0)), z(T(0)) {}
  V3(T a, T b, T c) : x(a), y(b), z(c) {}
  T& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  const T& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
template <class T> inline V3<T> operator+(const V3<T>& a, const V3<T>& b) { return V3<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class T> inline V3<T> operator-(const V3<T>& a, const V3<T>& b) { return V3<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class T> inline V3<T> operator-(const V3<T>& a) { return V3<T>(-a.x, -a.y, -a.z); }
template <class T> inline V3<T> operator*(const T& s, const V3<T>& a) { return V3<T>(s * a.x, s * a.y, s * a.z); }
template <class T> inline V3<T> operator*(const V3<T>& a, const T& s) { return V3<T>(a.x * s, a.y * s, a.z * s); }
template <class T> inline V3<T> operator/(const V3<T>& a, const T& s) { return V3<T>(a.x / s, a.y / s, a.z / s); }
template <class T> inline V3<T> cross(const V3<T>& a, const V3<T>& b) {
  return V3<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
template <class T> inline T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

template <class T> struct Quat {
  T x, y, z, w;
  Quat() : x(T(0)), y(T(0)), z(T(0)), w(T(1)) {}
  // Eigen constructor order (w, x, y, z)
  Quat(T w_, T x_, T y_, T z_) : x(x_), y(y_), z(z_), w(w_) {}
  static Quat from_coeffs(const T* c) { return Quat(c[3], c[0], c[1], c[2]); }  // Eigen::Map order x,y,z,w
  V3<T> vec() const { return V3<T>(x, y, z); }
  Quat conjugate() const { return Quat(w, -x, -y, -z); }
  T squaredNorm() const { return x * x + y * y + z * z + w * w; }
  T norm() const { return orc_sqrt(squaredNorm()); }
};
// Eigen quat_product (generic, non-vectorised)
template <class T> inline Quat<T> operator*(const Quat<T>& a, const Quat<T>& b) {
  return Quat<T>(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
                 a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                 a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
                 a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x); }
// Eigen QuaternionBase::_transformVector: v + w*uv + vec x uv, uv = 2 * vec x v
template <class T> inline V3<T> operator*(const Quat<T>& q, const V3<T>& v) {
  V3<T> uv = cross(q.vec(), v);
  uv = uv + uv;
  return v + q.w * uv + cross(q.vec(), uv); }
template <class T> inline Quat<T> scaled(const Quat<T>& q, const T& s) { return Quat<T>(q.w * s, q.x * s, q.y * s, q.z * s); }

// ---------------------------------------------------------------------------
// K/kontiki/math/quaternion_math.h
// ---------------------------------------------------------------------------
static const double kEps = 1e-16;            // quaternion_math.h:10
static const double kEpsUnitCheck = 1e-5;    // quaternion_math.h:11

struct nonunit_quat_error : std::runtime_error { using std::runtime_error::runtime_error; };

template <class T> inline Quat<T> logq(const Quat<T>& q) {  // quaternion_math.h:16-59
  T qn = q.norm();
  if (orc_abs(qn - T(1.0)) > kEpsUnitCheck) throw nonunit_quat_error("logq: Only implemented for unit quaternions");
  T k;
  T v_squared = q.x * q.x + q.y * q.y + q.z * q.z;
  if (v_squared > kEps) { T vn = orc_sqrt(v_squared); k = orc_atan2(vn, q.w) / vn; }
  else { k = T(1.0); }
  return Quat<T>(T(0), q.x * k, q.y * k, q.z * k);
}
template <class T> inline Quat<T> expq(const Quat<T>& q) {  // quaternion_math.h:62-89
  T v_squared = q.x * q.x + q.y * q.y + q.z * q.z;
  T ea = orc_exp(q.w);
  T ka, kv;
  if (v_squared > kEps) { T v_norm = orc_sqrt(v_squared); ka = ea * orc_cos(v_norm); kv = ea * orc_sin(v_norm) / v_norm; }
  else { ka = ea; kv = ea; }
  return Quat<T>(ka, kv * q.x, kv * q.y, kv * q.z);
}
template <class T> inline V3<T> angular_velocity(const Quat<T>& q, const Quat<T>& dq) {  // quaternion_math.h:92-95
  Quat<T> w = scaled(dq * q.conjugate(), T(2));
  return w.vec();
}

// ---------------------------------------------------------------------------
// K/kontiki/trajectories/spline_base.h:19-29 — basis matrices
// ---------------------------------------------------------------------------
static const double M_[4][4] = {{1. / 6., 4. / 6., 1. / 6., 0}, {-3. / 6., 0, 3. / 6., 0},
                                {3. / 6., -6. / 6, 3. / 6., 0}, {-1. / 6., 3. / 6., -3. / 6., 1. / 6.}};
static const double M_cumul_[4][4] = {{6. / 6., 5. / 6., 1. / 6., 0}, {0. / 6., 3. / 6., 3. / 6., 0},
                                      {0. / 6., -3. / 6., 3. / 6., 0}, {0. / 6., 1. / 6., -2. / 6., 1. / 6.}};

enum EvalFlags { EvalPosition = 1, EvalVelocity = 2, EvalAcceleration = 4, EvalOrientation = 8, EvalAngularVelocity = 16 };  // trajectory.h:17-23

struct range_error : std::range_error { using std::range_error::range_error; };

// spline_base.h:31-39 SplineSegmentMeta
struct SegMeta { double t0, dt; int n; double MinTime() const { return t0; } double MaxTime() const { return t0 + (n - 3) * dt; } };

template <class T> struct Eval {  // trajectory.h:26-35 TrajectoryEvaluation
  V3<T> position, velocity, acceleration, angular_velocity; Quat<T> orientation; };

// U^T * M  (Eigen row-vector * matrix: B_j = sum_i U_i M_ij, i ascending)
template <class T> inline void rowvec_times(const T U[4], const double Mx[4][4], T B[4]) {
  for (int j = 0; j < 4; ++j) B[j] = U[0] * T(Mx[0][j]) + U[1] * T(Mx[1][j]) + U[2] * T(Mx[2][j]) + U[3] * T(Mx[3][j]);
}

// spline_base.h:153-157 CalculateIndexAndInterpolationAmount
template <class T> inline void index_and_u(const SegMeta& m, const T& t, int& i0, T& u) {
  T s = (t - T(m.t0)) / T(m.dt);
  i0 = static_cast<int>(std::floor(value_of(s)));
  u = s - T(double(i0));
}

// K/kontiki/trajectories/uniform_r3_spline_trajectory.h:36-103
// cps[i] points at the 3 scalars of control point i of this segment.
template <class T> inline void r3_segment_evaluate(const SegMeta& m, const T* const* cps, const T& t, int flags, Eval<T>& out) {
  int i0; T u; index_and_u(m, t, i0, u);
  const int N = m.n;
  if ((N < 4) || (i0 < 0) || (i0 > (N - 4))) throw range_error("r3: t out of range for spline segment");
  T Up[4], Uv[4], Ua[4], Bp[4], Bv[4], Ba[4];
  T u2 = T(0), u3 = T(0);
  T dt_inv = T(1) / T(m.dt);
  if ((flags & EvalPosition) || (flags & EvalVelocity)) u2 = orc_pow(u, 2);
  if (flags & EvalPosition) u3 = orc_pow(u, 3);
  if (flags & EvalPosition) { Up[0] = T(1); Up[1] = u; Up[2] = u2; Up[3] = u3; rowvec_times(Up, M_, Bp); out.position = V3<T>(); }
  if (flags & EvalVelocity) { Uv[0] = dt_inv * T(0); Uv[1] = dt_inv * T(1); Uv[2] = dt_inv * (T(2) * u); Uv[3] = dt_inv * (T(3) * u2); rowvec_times(Uv, M_, Bv); out.velocity = V3<T>(); }
  if (flags & EvalAcceleration) { T d2 = orc_pow(dt_inv, 2); Ua[0] = d2 * T(0); Ua[1] = d2 * T(0); Ua[2] = d2 * T(2); Ua[3] = d2 * (T(6) * u); rowvec_times(Ua, M_, Ba); out.acceleration = V3<T>(); }
  for (int i = i0; i < i0 + 4; ++i) {
    V3<T> cp(cps[i][0], cps[i][1], cps[i][2]);
    if (flags & EvalPosition) out.position = out.position + Bp[i - i0] * cp;
    if (flags & EvalVelocity) out.velocity = out.velocity + Bv[i - i0] * cp;
    if (flags & EvalAcceleration) out.acceleration = out.acceleration + Ba[i - i0] * cp;
  }
}

// K/kontiki/trajectories/uniform_so3_spline_trajectory.h:46-125
template <class T> inline void so3_segment_evaluate(const SegMeta& m, const T* const* cps, const T& t, int flags, Eval<T>& out) {
  if (!(flags & (EvalOrientation | EvalAngularVelocity))) return;
  int i0; T u; index_and_u(m, t, i0, u);
  const int N = m.n;
  if ((N < 4) || (i0 < 0) || (i0 > (N - 4))) throw range_error("so3: t out of range for spline segment");
  T U[4], dU[4], B[4], dB[4];
  T u2 = orc_pow(u, 2);
  T u3 = orc_pow(u, 3);
  T dt_inv = T(1) / T(m.dt);
  U[0] = T(1); U[1] = u; U[2] = u2; U[3] = u3;
  rowvec_times(U, M_cumul_, B);
  const bool need_w = (flags & EvalAngularVelocity) != 0;
  if (need_w) { dU[0] = dt_inv * T(0); dU[1] = dt_inv * T(1); dU[2] = dt_inv * (T(2) * u); dU[3] = dt_inv * (T(3) * u2); rowvec_times(dU, M_cumul_, dB); }
  Quat<T> dq_parts[3] = {Quat<T>(T(1), T(0), T(0), T(0)), Quat<T>(T(1), T(0), T(0), T(0)), Quat<T>(T(1), T(0), T(0), T(0))};
  Quat<T> q = Quat<T>::from_coeffs(cps[i0]);
  const int K = i0 + 4;
  for (int i = i0 + 1; i < K; ++i) {
    Quat<T> qa = Quat<T>::from_coeffs(cps[i - 1]);
    Quat<T> qb = Quat<T>::from_coeffs(cps[i]);
    Quat<T> omega = logq(qa.conjugate() * qb);
    Quat<T> eomegab = expq(scaled(omega, B[i - i0]));
    q = q * eomegab;
    if (need_w) {
      for (int j = i0 + 1; j < K; ++j) {
        const int mm = j - i0 - 1;
        if (i == j) dq_parts[mm] = dq_parts[mm] * scaled(omega, dB[i - i0]);
        dq_parts[mm] = dq_parts[mm] * eomegab;
      }
    }
  }
  out.orientation = q;
  if (need_w) {
    Quat<T> sum(dq_parts[0].w + dq_parts[1].w + dq_parts[2].w, dq_parts[0].x + dq_parts[1].x + dq_parts[2].x,
                dq_parts[0].y + dq_parts[1].y + dq_parts[2].y, dq_parts[0].z + dq_parts[1].z + dq_parts[2].z);
    Quat<T> dq = Quat<T>::from_coeffs(cps[i0]) * sum;
    out.angular_velocity = angular_velocity(q, dq);
  }
}

// spline_base.h:63-94 SplineMeta + :194-222 SplineView::Evaluate (segment dispatch with t-1e-5 retry)
struct SplineMeta { std::vector<SegMeta> segments; int NumParameters() const { int n = 0; for (auto& s : segments) n += s.n; return n; } };

template <class T, class SegFn>
inline void spline_view_evaluate(const SplineMeta& meta, const T* const* params, const T& t, int flags, Eval<T>& out, SegFn seg_eval) {
  int offset = 0;
  for (const auto& seg : meta.segments) {
    if ((value_of(t) >= seg.MinTime()) && (value_of(t) < seg.MaxTime())) { seg_eval(seg, params + offset, t, flags, out); return; }
    else {
      T t_temp = t - T(0.00001);
      if ((value_of(t_temp) >= seg.MinTime()) && (value_of(t_temp) < seg.MaxTime())) { seg_eval(seg, params + offset, t_temp, flags, out); return; }
    }
    offset += seg.n;
  }
  throw range_error("No segment found for time t");
}

// split_trajectory.h:15-25 SplitMeta, :41-58 SplitView::Evaluate.  params: r3 control points first, then so3.
struct SplitMeta { SplineMeta r3, so3; int NumParameters() const { return r3.NumParameters() + so3.NumParameters(); } };

// A trajectory "view" over a parameter-block array; kind selects Split (R3+SO3) or SO3-only (Solve #0 estimator).
template <class T> struct TrajView {
  const SplitMeta* meta; const T* const* params; bool so3_only;
  void Evaluate(const T& t, int flags, Eval<T>& out) const {
    if (so3_only) {  // type::Trajectory<UniformSO3SplineTrajectory>: SplineView::Evaluate directly
      spline_view_evaluate<T>(meta->so3, params, t, flags, out, so3_segment_evaluate<T>);
      return;
    }
    const int lin = flags & (EvalPosition | EvalVelocity | EvalAcceleration);
    const int rot = flags & (EvalOrientation | EvalAngularVelocity);
    if (lin) spline_view_evaluate<T>(meta->r3, params, t, lin, out, r3_segment_evaluate<T>);
    if (rot) spline_view_evaluate<T>(meta->so3, params + meta->r3.NumParameters(), t, rot, out, so3_segment_evaluate<T>);
  }
};

// spline_base.h:380-426 SplineEntity::AddToProblem — which master knots a residual block owns, and its segment metas.
// Returns the master knot indices pushed (in push order) and fills meta.segments.
inline void spline_add_to_problem(double master_t0, double master_dt, const std::vector<std::pair<double, double>>& times,
                                  SplineMeta& meta, std::vector<int>& knots) {
  int current_segment_start = 0, current_segment_end = -1;
  SegMeta master{master_t0, master_dt, 0};
  for (auto tt : times) {
    int i1, i2; double u_notused;
    index_and_u<double>(master, tt.first, i1, u_notused);
    index_and_u<double>(master, tt.second, i2, u_notused);
    if (i1 > current_segment_end) {
      double segment_t0 = master_t0 + master_dt * i1;
      meta.segments.push_back(SegMeta{segment_t0, master_dt, 0});
      current_segment_start = i1;
    } else {
      i1 = current_segment_end + 1;
    }
    SegMeta& cur = meta.segments.back();
    for (int i = i1; i < (i2 + 4); ++i) { knots.push_back(i); cur.n += 1; }
    current_segment_end = current_segment_start + cur.n - 1;
  }
}

// ---------------------------------------------------------------------------
// Sensors.  K/kontiki/sensors/sensors.h:36-85 (q_rel[4], p_rel[3], time_offset[1])
// ---------------------------------------------------------------------------
template <class T> struct SensorView {
  const T* const* p;  // blocks: 0 q_rel(x,y,z,w) 1 p_rel 2 time_offset
  Quat<T> relative_orientation() const { return Quat<T>::from_coeffs(p[0]); }
  V3<T> relative_position() const { return V3<T>(p[1][0], p[1][1], p[1][2]); }
  T time_offset() const { return p[2][0]; }
};

static const double STANDARD_GRAVITY = -9.79;  // imu.h:25

// imu.h:36-101 + constant_bias_imu.h:33-61: blocks 3 roll, 4 pitch, 5 acc bias, 6 gyro bias
template <class T> struct ImuView : SensorView<T> {
  V3<T> refined_gravity() const {  // imu.h:61-70
    T cosRoll = orc_cos(this->p[3][0]); T sinRoll = orc_sin(this->p[3][0]);
    T cosPitch = orc_cos(this->p[4][0]); T sinPitch = orc_sin(this->p[4][0]);
    return V3<T>(-sinPitch * cosRoll * T(STANDARD_GRAVITY), sinRoll * T(STANDARD_GRAVITY), -cosRoll * cosPitch * T(STANDARD_GRAVITY));
  }
  V3<T> accelerometer_bias() const { return V3<T>(this->p[5][0], this->p[5][1], this->p[5][2]); }
  V3<T> gyroscope_bias() const { return V3<T>(this->p[6][0], this->p[6][1], this->p[6][2]); }
  V3<T> Gyroscope(const TrajView<T>& traj, const T& t) const {  // imu.h:87-91, constant_bias_imu.h:57-61
    Eval<T> r; traj.Evaluate(t + this->time_offset(), EvalOrientation | EvalAngularVelocity, r);
    V3<T> base = r.orientation.conjugate() * r.angular_velocity;
    return base + gyroscope_bias();
  }
  V3<T> Accelerometer(const TrajView<T>& traj, const T& t) const {  // imu.h:95-101, constant_bias_imu.h:51-55
    Eval<T> r; traj.Evaluate(t + this->time_offset(), EvalOrientation | EvalAcceleration, r);
    V3<T> base = r.orientation.conjugate() * (r.acceleration + refined_gravity());
    return base + accelerometer_bias();
  }
};

// pinhole_camera.h:20-41 PinholeMeta (+ camera.h:25-29)
struct PinholeMeta {
  double readout = 0; int rows = 0, cols = 0;
  double fx = 1, fy = 1, cx = 0, cy = 0, k1 = 0, k2 = 0, p1 = 0, p2 = 0, k3 = 0;
  bool do_distortion = false;
  double inv_K11 = 1, inv_K13 = 0, inv_K22 = 1, inv_K23 = 0;
  void finalize() {  // pinhole_camera.h:59-82
    inv_K11 = 1.0 / fx; inv_K13 = -cx / fx; inv_K22 = 1.0 / fy; inv_K23 = -cy / fy;
    do_distortion = std::fabs(k1) > 1e-5 || std::fabs(k2) > 1e-5 || std::fabs(p1) > 1e-5 || std::fabs(p1) > 1e-5;  // sic: p1 twice (:78)
  }
};

template <class T> struct CameraView : SensorView<T> {
  const PinholeMeta* meta;
  void distortion(const T pu[2], T du[2]) const {  // pinhole_camera.h:199-215
    T k1 = T(meta->k1), k2 = T(meta->k2), p1 = T(meta->p1), p2 = T(meta->p2), k3 = T(meta->k3);
    T mx2_u = pu[0] * pu[0], my2_u = pu[1] * pu[1], mxy_u = pu[0] * pu[1];
    T rho2_u = mx2_u + my2_u;
    T rad_dist_u = k1 * rho2_u + k2 * rho2_u * rho2_u + k3 * rho2_u * rho2_u * rho2_u;
    du[0] = pu[0] * rad_dist_u + T(2.0) * p1 * mxy_u + p2 * (rho2_u + T(2.0) * mx2_u);
    du[1] = pu[1] * rad_dist_u + T(2.0) * p2 * mxy_u + p1 * (rho2_u + T(2.0) * my2_u);
  }
  V3<T> Unproject(const T y[2]) const {  // pinhole_camera.h:113-124, :131-191
    if (meta->do_distortion) {
      T mx_d = T(meta->inv_K11) * y[0] + T(meta->inv_K13);
      T my_d = T(meta->inv_K22) * y[1] + T(meta->inv_K23);
      T du[2]; T pu[2] = {mx_d, my_d};
      distortion(pu, du);
      T mx_u = mx_d - du[0], my_u = my_d - du[1];
      for (int i = 1; i < 8; ++i) { T q[2] = {mx_u, my_u}; distortion(q, du); mx_u = mx_d - du[0]; my_u = my_d - du[1]; }
      return V3<T>(mx_u, my_u, T(1));
    }
    // camera_matrix().inverse() * (u, v, 1) for K = [fx 0 cx; 0 fy cy; 0 0 1] (Eigen 3x3 inverse via cofactors/det):
    // inverse of this upper-triangular K is [1/fx 0 -cx/fx; 0 1/fy -cy/fy; 0 0 1]; restated in closed form.
    const double fx = meta->fx, fy = meta->fy, cx = meta->cx, cy = meta->cy;
    const double det = fx * fy;  // Eigen computes cofactor/det; entries below are cofactor * (1/det)
    const double invdet = 1.0 / det;
    const double i00 = fy * invdet, i02 = (-cx * fy) * invdet, i11 = fx * invdet, i12 = (-(fx * cy)) * invdet, i22 = (fx * fy) * invdet;
    return V3<T>(T(i00) * y[0] + T(i02), T(i11) * y[1] + T(i12), T(i22) * T(1));
  }
  void Project(const V3<T>& P, T p[2]) const {  // pinhole_camera.h:96-110 -> spaceToPlane :217-238
    const T eps = T(1e-32);
    T pu[2] = {P.x / (eps + P.z), P.y / (eps + P.z)};
    T pd[2];
    if (!meta->do_distortion) { pd[0] = pu[0]; pd[1] = pu[1]; }
    else { T du[2]; distortion(pu, du); pd[0] = pu[0] + du[0]; pd[1] = pu[1] + du[1]; }
    p[0] = T(meta->fx) * pd[0] + T(meta->cx);
    p[1] = T(meta->fy) * pd[1] + T(meta->cy);
  }
};

// ---------------------------------------------------------------------------
// Measurement residuals (Error<T>()) in reference order.
// ---------------------------------------------------------------------------
// gyroscope_measurement.h:36-38
template <class T> inline void gyro_error(const ImuView<T>& imu, const TrajView<T>& traj, double t, const double w[3], double weight, T r[3]) {
  V3<T> m = imu.Gyroscope(traj, T(t));
  r[0] = T(weight) * (T(w[0]) - m.x); r[1] = T(weight) * (T(w[1]) - m.y); r[2] = T(weight) * (T(w[2]) - m.z);
}
// accelerometer_measurement.h:39-41
template <class T> inline void accel_error(const ImuView<T>& imu, const TrajView<T>& traj, double t, const double a[3], double weight, T r[3]) {
  V3<T> m = imu.Accelerometer(traj, T(t));
  r[0] = T(weight) * (T(a[0]) - m.x); r[1] = T(weight) * (T(a[1]) - m.y); r[2] = T(weight) * (T(a[2]) - m.z);
}
// point-to-plane tail shared by lidar_surfel_point.h:54-69 and camera_surfel_landmark.h:77-91
template <class T> inline T plane_dist(const T* plane_cp, const V3<T>& p_M) {
  V3<T> Pi(plane_cp[0], plane_cp[1], plane_cp[2]);
  T plane_d = orc_sqrt(Pi.x * Pi.x + Pi.y * Pi.y + Pi.z * Pi.z);
  T n0 = Pi.x / plane_d, n1 = Pi.y / plane_d, n2 = Pi.z / plane_d;
  return (n0 * p_M.x + n1 * p_M.y + n2 * p_M.z) - plane_d;  // ceres::DotProduct order
}
// lidar_surfel_point.h:31-82
template <class T> inline T surfel_error(const TrajView<T>& traj, const SensorView<T>& lidar, const T* plane_cp,
                                         const double pt[3], double timestamp, double map_time, double weight) {
  const int flags = EvalPosition | EvalOrientation;
  Eval<T> T0, Tk;
  traj.Evaluate(T(map_time) + lidar.time_offset(), flags, T0);
  traj.Evaluate(T(timestamp) + lidar.time_offset(), flags, Tk);
  const V3<T> p_LinI = lidar.relative_position();
  const Quat<T> q_LtoI = lidar.relative_orientation();
  V3<T> p_Lk{T(pt[0]), T(pt[1]), T(pt[2])};
  V3<T> p_I = q_LtoI * p_Lk + p_LinI;
  V3<T> p_temp = T0.orientation.conjugate() * (Tk.orientation * p_I + Tk.position - T0.position);
  V3<T> p_M = q_LtoI.conjugate() * (p_temp - p_LinI);
  return T(weight) * plane_dist(plane_cp, p_M);
}
// static_rscamera_measurement.h:20-60 + Error :93-99
template <class T> inline void reproj_error(const TrajView<T>& traj, const CameraView<T>& cam, const T& inverse_depth,
                                            const double uv_ref[2], double t0_ref, const double uv_obs[2], double t0_obs,
                                            double weight, T r[2]) {
  T time_offset = cam.time_offset();
  T row_delta = T(cam.meta->readout) / T(double(cam.meta->rows));
  T t_ref = T(t0_ref) + time_offset + T(uv_ref[1]) * row_delta;
  T t_obs = T(t0_obs) + time_offset + T(uv_obs[1]) * row_delta;
  const int flags = EvalPosition | EvalOrientation;
  Eval<T> er, eo;
  traj.Evaluate(t_ref, flags, er);
  traj.Evaluate(t_obs, flags, eo);
  const V3<T> p_CinI = cam.relative_position();
  const Quat<T> q_CinI = cam.relative_orientation();
  const V3<T> p_ct = q_CinI.conjugate() * (-p_CinI);
  const Quat<T> q_ct = q_CinI.conjugate();
  T y[2] = {T(uv_ref[0]), T(uv_ref[1])};
  V3<T> yh = cam.Unproject(y);
  V3<T> X_ref = q_ct.conjugate() * (yh - inverse_depth * p_ct);
  V3<T> X = er.orientation * X_ref + er.position * inverse_depth;
  V3<T> X_obs = eo.orientation.conjugate() * (X - inverse_depth * eo.position);
  V3<T> X_camera = q_ct * X_obs + p_ct * inverse_depth;
  T yhat[2]; cam.Project(X_camera, yhat);
  r[0] = T(weight) * (T(uv_obs[0]) - yhat[0]);
  r[1] = T(weight) * (T(uv_obs[1]) - yhat[1]);
}
// camera_surfel_landmark.h:29-103 (inverse depth enters as a constant: :159-161)
template <class T> inline T camsurf_error(const TrajView<T>& traj, const CameraView<T>& cam, const SensorView<T>& lidar, const T* plane_cp,
                                          double inverse_depth_const, const double uv_ref[2], double timestamp, double map_time, double weight) {
  const T eps = T(1e-8);
  const int flags = EvalPosition | EvalOrientation;
  Eval<T> T0, Tk;
  traj.Evaluate(T(map_time) + cam.time_offset(), flags, T0);
  traj.Evaluate(T(timestamp) + cam.time_offset(), flags, Tk);
  const V3<T> p_CinI = cam.relative_position();
  const V3<T> p_LinI = lidar.relative_position();
  const Quat<T> q_CtoI = cam.relative_orientation();
  const Quat<T> q_LtoI = lidar.relative_orientation();
  T y[2] = {T(uv_ref[0]), T(uv_ref[1])};
  V3<T> yh = cam.Unproject(y) / (T(inverse_depth_const) + eps);
  V3<T> p_I = q_CtoI * yh + p_CinI;
  V3<T> p_temp = T0.orientation.conjugate() * (Tk.orientation * p_I + Tk.position - T0.position);
  V3<T> p_M = q_LtoI.conjugate() * (p_temp - p_LinI);
  return T(weight) * plane_dist(plane_cp, p_M);
}
// orientation_measurement.h:30-33; Eigen angularDistance: 2*atan2(|vec(d)|, |w(d)|), d = q * other.conjugate()
template <class T> inline T orientation_error(const TrajView<T>& traj, double t, const double q_wxyz[4], double weight) {
  Eval<T> r; traj.Evaluate(T(t), EvalOrientation, r);
  Quat<T> q{T(q_wxyz[0]), T(q_wxyz[1]), T(q_wxyz[2]), T(q_wxyz[3])};
  Quat<T> d = q * r.orientation.conjugate();
  T vn = orc_sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
  return T(weight) * (T(2) * orc_atan2(vn, orc_abs(d.w)));
}

}  // namespace orc
