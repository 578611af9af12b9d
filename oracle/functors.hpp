// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it, and only as the
// checker.  PARITY UNPINNED: the reference ships no golden vectors or tests for this path (SURVEY.md §4,
// §8c); this restatement is pinned by the mpmath-derived fixtures in tests/golden/ instead.
//
// functors.hpp — CPU restatement of the three ACTIVE cost functors of the reference and of the pieces of
// Eigen / Ceres they call, written so that each statement can be read against the reference line it follows.
//
//   SixDOFError::operator()                          reference src/CeresResidues.h:32-69   (ctor :22-28)
//   SixDOFErrorWithSwitchingConstraints::operator()  reference src/CeresResidues.h:158-201 (ctor :148-154)
//   NodePoseRegularization::operator()               reference src/CeresResidues.h:104-127
//
// Third-party behaviour restated from the libraries' published algorithms (neither is vendored in
// /root/reference, versions unpinned: CMakeLists.txt:22-23 `find_package(Eigen3)`, `find_package(Ceres)`):
//   Eigen 3.x  Quaternion product, conjugate, `_transformVector`, `toRotationMatrix`,
//              rotation-matrix -> quaternion assignment (branch on trace / largest diagonal)
//   Ceres 1.12-2.1  `Jet<double,N>` forward-mode dual numbers (AutoDiffCostFunction),
//              `EigenQuaternionParameterization::{Plus,ComputeJacobian}`
#pragma once
#include <cmath>

namespace orc {

// ------------------------------------------------------------------------------------------------
// ceres::Jet<double,N> — value + N partials, the arithmetic AutoDiffCostFunction differentiates with.
// ------------------------------------------------------------------------------------------------
template <int N>
struct Jet {
    double a;
    double v[N];
    Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
    explicit Jet(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
    Jet(double s, int k) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; v[k] = 1.0; }
};
template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f) { Jet<N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
    // ceres/jet.h: h = f/g, dh = (df - f/g dg)/g
    Jet<N> h; const double gi = 1.0 / g.a; const double fg = f.a * gi; h.a = fg;
    for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - fg * g.v[i]) * gi; return h;
}
template <int N> inline Jet<N> operator+(const Jet<N>& f, double s) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator+(double s, const Jet<N>& f) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, double s) { Jet<N> h = f; h.a -= s; return h; }
template <int N> inline Jet<N> operator-(double s, const Jet<N>& f) { Jet<N> h = -f; h.a += s; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, double s) { Jet<N> h; h.a = f.a * s; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s; return h; }
template <int N> inline Jet<N> operator*(double s, const Jet<N>& f) { return f * s; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, double s) { return f * (1.0 / s); }
template <int N> inline Jet<N> operator/(double s, const Jet<N>& g) {
    Jet<N> h; const double gi = 1.0 / g.a; h.a = s * gi; const double m = -s * gi * gi;
    for (int i = 0; i < N; ++i) h.v[i] = m * g.v[i]; return h;
}
template <int N> inline Jet<N>& operator+=(Jet<N>& f, const Jet<N>& g) { f = f + g; return f; }
template <int N> inline Jet<N>& operator*=(Jet<N>& f, const Jet<N>& g) { f = f * g; return f; }
template <int N> inline bool operator>(const Jet<N>& f, const Jet<N>& g) { return f.a > g.a; }  // ceres compares the scalar part
template <int N> inline bool operator>(const Jet<N>& f, double s) { return f.a > s; }
template <int N> inline Jet<N> sqrt(const Jet<N>& f) { Jet<N> h; h.a = std::sqrt(f.a); const double t = 1.0 / (2.0 * h.a); for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * t; return h; }
inline double sqrt(double x) { return std::sqrt(x); }

template <typename T> inline double scalar_part(const T& x) { return x.a; }
template <> inline double scalar_part<double>(const double& x) { return x; }

// ------------------------------------------------------------------------------------------------
// The slice of Eigen the functors use.  Coefficients stored x,y,z,w (Eigen::Quaternion::coeffs()).
// ------------------------------------------------------------------------------------------------
template <typename T> struct Vec3 { T x, y, z; };
template <typename T> struct Quat { T x, y, z, w; };
template <typename T> struct Mat3 { T m[3][3]; };  // m[row][col]

template <typename T> inline Vec3<T> operator-(const Vec3<T>& a, const Vec3<T>& b) { return Vec3<T>{a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> inline Vec3<T> operator+(const Vec3<T>& a, const Vec3<T>& b) { return Vec3<T>{a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> inline Vec3<T> cross(const Vec3<T>& a, const Vec3<T>& b) {
    return Vec3<T>{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Eigen::QuaternionBase::conjugate()
template <typename T> inline Quat<T> conjugate(const Quat<T>& q) { return Quat<T>{-q.x, -q.y, -q.z, q.w}; }
// Eigen internal::quat_product (generic, non-SIMD form)
template <typename T> inline Quat<T> operator*(const Quat<T>& a, const Quat<T>& b) {
    Quat<T> r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}
// Eigen::QuaternionBase::_transformVector:  uv = 2 (u x v);  v + w uv + u x uv   (assumes |q| = 1; never normalises)
template <typename T> inline Vec3<T> operator*(const Quat<T>& q, const Vec3<T>& v) {
    const Vec3<T> u{q.x, q.y, q.z};
    Vec3<T> uv = cross(u, v);
    uv = uv + uv;
    const Vec3<T> wuv{q.w * uv.x, q.w * uv.y, q.w * uv.z};
    return v + wuv + cross(u, uv);
}
// Eigen::QuaternionBase::toRotationMatrix
template <typename T> inline Mat3<T> toRotationMatrix(const Quat<T>& q) {
    const T tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
    const T twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const T txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const T tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    Mat3<T> R;
    R.m[0][0] = 1.0 - (tyy + tzz); R.m[0][1] = txy - twz;         R.m[0][2] = txz + twy;
    R.m[1][0] = txy + twz;         R.m[1][1] = 1.0 - (txx + tzz); R.m[1][2] = tyz - twx;
    R.m[2][0] = txz - twy;         R.m[2][1] = tyz + twx;         R.m[2][2] = 1.0 - (txx + tyy);
    return R;
}
// Eigen internal::quaternionbase_assign_impl<Matrix3,3,3>::run — `Quaterniond(Matrix3d)` / `Quaternion<T> q(R)`.
template <typename T> inline Quat<T> fromRotationMatrix(const Mat3<T>& mat) {
    T q[4];  // x,y,z,w
    T t = mat.m[0][0] + mat.m[1][1] + mat.m[2][2];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (mat.m[2][1] - mat.m[1][2]) * t;
        q[1] = (mat.m[0][2] - mat.m[2][0]) * t;
        q[2] = (mat.m[1][0] - mat.m[0][1]) * t;
    } else {
        int i = 0;
        if (mat.m[1][1] > mat.m[0][0]) i = 1;
        if (mat.m[2][2] > mat.m[i][i]) i = 2;
        const int j = (i + 1) % 3;
        const int k = (j + 1) % 3;
        t = sqrt(mat.m[i][i] - mat.m[j][j] - mat.m[k][k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (mat.m[k][j] - mat.m[j][k]) * t;
        q[j] = (mat.m[j][i] + mat.m[i][j]) * t;
        q[k] = (mat.m[k][i] + mat.m[i][k]) * t;
    }
    return Quat<T>{q[0], q[1], q[2], q[3]};
}

// A Matrix4d as the reference passes it around: 16 doubles, column-major (Eigen default).
struct Mat4d {
    double d[16];
    double operator()(int r, int c) const { return d[c * 4 + r]; }
    double& operator()(int r, int c) { return d[c * 4 + r]; }
};

// General 4x4 inverse by cofactors (what Eigen's `Matrix<T,4,4>::inverse()` computes, CeresResidues.h:117).
inline Mat4d inverse4(const Mat4d& A) {
    const double* m = A.d;  // column-major, but the adjugate formula below is layout-agnostic up to transposition,
    double inv[16];         // and (A^T)^-1 = (A^-1)^T, so applying it to the raw array is consistent.
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    Mat4d R;
    for (int i = 0; i < 16; ++i) R.d[i] = inv[i] / det;
    return R;
}

// ------------------------------------------------------------------------------------------------
// SixDOFError — reference src/CeresResidues.h:19-90
// ------------------------------------------------------------------------------------------------
struct SixDOFError {
    Quat<double> observed_c1_q_c2;
    Vec3<double> observed_c1_t_c2;
    double weight;
    // ctor, CeresResidues.h:22-28
    SixDOFError(const Mat4d& observed__c1_T_c2, double _weight) {
        Mat3<double> R;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R.m[r][c] = observed__c1_T_c2(r, c);
        observed_c1_q_c2 = fromRotationMatrix(R);                                                         // :24
        observed_c1_t_c2 = Vec3<double>{observed__c1_T_c2(0, 3), observed__c1_T_c2(1, 3), observed__c1_T_c2(2, 3)};  // :25
        weight = _weight;                                                                                 // :27
    }
    template <typename T> static Quat<T> castq(const Quat<double>& q) { return Quat<T>{T(q.x), T(q.y), T(q.z), T(q.w)}; }
    template <typename T> static Vec3<T> castv(const Vec3<double>& v) { return Vec3<T>{T(v.x), T(v.y), T(v.z)}; }

    // operator(), CeresResidues.h:32-69.   q stored x,y,z,w (Eigen::Map<const Quaternion<T>>, :41,:45)
    template <typename T>
    bool operator()(const T* const q1, const T* const t1, const T* const q2, const T* const t2, T* residue_ptr) const {
        const Vec3<T> p_1{t1[0], t1[1], t1[2]};                                   // :40
        const Quat<T> q_1{q1[0], q1[1], q1[2], q1[3]};                            // :41
        const Vec3<T> p_2{t2[0], t2[1], t2[2]};                                   // :44
        const Quat<T> q_2{q2[0], q2[1], q2[2], q2[3]};                            // :45
        const Quat<T> q_1_inverse = conjugate(q_1);                               // :48
        const Quat<T> q_12_estimated = q_1_inverse * q_2;                         // :49
        const Vec3<T> p_12_estimated = q_1_inverse * (p_2 - p_1);                 // :50
        const Quat<T> delta_q = conjugate(q_12_estimated) * castq<T>(observed_c1_q_c2);                     // :53
        const Vec3<T> delta_t = conjugate(q_12_estimated) * (castv<T>(observed_c1_t_c2) - p_12_estimated);  // :54
        residue_ptr[0] = delta_t.x; residue_ptr[1] = delta_t.y; residue_ptr[2] = delta_t.z;                 // :58
        residue_ptr[3] = T(2.0) * delta_q.x; residue_ptr[4] = T(2.0) * delta_q.y; residue_ptr[5] = T(2.0) * delta_q.z;  // :59
        const T s = T(1.0);                                                       // :65 (DCS disabled, :64 commented)
        const T sw = s * T(weight);
        for (int i = 0; i < 6; ++i) residue_ptr[i] = residue_ptr[i] * sw;         // :66
        return true;
    }
};

// ------------------------------------------------------------------------------------------------
// SixDOFErrorWithSwitchingConstraints — reference src/CeresResidues.h:145-222
// ------------------------------------------------------------------------------------------------
struct SixDOFErrorWithSwitchingConstraints {
    Quat<double> observed_c1_q_c2;
    Vec3<double> observed_c1_t_c2;
    double weight;  // stored (:153) and never used (:198 `//* T(weight)`)
    SixDOFErrorWithSwitchingConstraints(const Mat4d& observed__c1_T_c2, double _weight) {
        Mat3<double> R;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R.m[r][c] = observed__c1_T_c2(r, c);
        observed_c1_q_c2 = fromRotationMatrix(R);                                                         // :150
        observed_c1_t_c2 = Vec3<double>{observed__c1_T_c2(0, 3), observed__c1_T_c2(1, 3), observed__c1_T_c2(2, 3)};  // :151
        weight = _weight;                                                                                 // :153
    }
    // operator(), CeresResidues.h:158-201
    template <typename T>
    bool operator()(const T* const q1, const T* const t1, const T* const q2, const T* const t2,
                    const T* const switching_var, T* residue_ptr) const {
        const Vec3<T> p_1{t1[0], t1[1], t1[2]};                                   // :169
        const Quat<T> q_1{q1[0], q1[1], q1[2], q1[3]};                            // :170
        const Vec3<T> p_2{t2[0], t2[1], t2[2]};                                   // :173
        const Quat<T> q_2{q2[0], q2[1], q2[2], q2[3]};                            // :174
        const Quat<T> q_1_inverse = conjugate(q_1);                               // :177
        const Quat<T> q_12_estimated = q_1_inverse * q_2;                         // :178
        const Vec3<T> p_12_estimated = q_1_inverse * (p_2 - p_1);                 // :179
        const Quat<T> delta_q = conjugate(q_12_estimated) * SixDOFError::castq<T>(observed_c1_q_c2);                     // :182
        const Vec3<T> delta_t = conjugate(q_12_estimated) * (SixDOFError::castv<T>(observed_c1_t_c2) - p_12_estimated);  // :183
        residue_ptr[0] = delta_t.x; residue_ptr[1] = delta_t.y; residue_ptr[2] = delta_t.z;                 // :187
        residue_ptr[3] = T(2.0) * delta_q.x; residue_ptr[4] = T(2.0) * delta_q.y; residue_ptr[5] = T(2.0) * delta_q.z;  // :188
        residue_ptr[6] = T(1.0) * (T(1.0) - switching_var[0]);                    // :189
        const T s = switching_var[0];                                             // :197
        for (int i = 0; i < 7; ++i) residue_ptr[i] = residue_ptr[i] * s;          // :198  (weight NOT applied)
        return true;
    }
};

// ------------------------------------------------------------------------------------------------
// NodePoseRegularization — reference src/CeresResidues.h:96-141
// ------------------------------------------------------------------------------------------------
struct NodePoseRegularization {
    Mat4d nodepose;
    double weight;
    NodePoseRegularization(const Mat4d& _nodepose, double _weight) : nodepose(_nodepose), weight(_weight) {}  // :99
    // operator(), CeresResidues.h:104-127
    template <typename T>
    bool operator()(const T* const q1, const T* const t1, T* residue_ptr) const {
        const Vec3<T> p_1{t1[0], t1[1], t1[2]};                                   // :108
        const Quat<T> q_1{q1[0], q1[1], q1[2], q1[3]};                            // :109
        // npose = [R(q_1) p_1; 0 1]                                              // :110-112
        const Mat3<T> Rn = toRotationMatrix(q_1);
        // f = nodepose.cast<T>(); delta = f.inverse() * npose                    // :115-117
        // f carries no partials, so f.inverse() is the double inverse cast to T.
        const Mat4d fi = inverse4(nodepose);
        T delta[3][4];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c)
                delta[r][c] = T(fi(r, 0)) * Rn.m[0][c] + T(fi(r, 1)) * Rn.m[1][c] + T(fi(r, 2)) * Rn.m[2][c] + T(fi(r, 3)) * T(0.0);
            delta[r][3] = T(fi(r, 0)) * p_1.x + T(fi(r, 1)) * p_1.y + T(fi(r, 2)) * p_1.z + T(fi(r, 3)) * T(1.0);
        }
        Mat3<T> R;                                                                // :118
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R.m[r][c] = delta[r][c];
        const Quat<T> delta_q = fromRotationMatrix(R);                            // :119
        residue_ptr[0] = T(weight) * delta[0][3];                                 // :123
        residue_ptr[1] = T(weight) * delta[1][3];
        residue_ptr[2] = T(weight) * delta[2][3];
        residue_ptr[3] = T(weight) * T(2.0) * delta_q.x;                          // :124
        residue_ptr[4] = T(weight) * T(2.0) * delta_q.y;
        residue_ptr[5] = T(weight) * T(2.0) * delta_q.z;
        return true;
    }
};

// ------------------------------------------------------------------------------------------------
// ceres::EigenQuaternionParameterization (used at reference src/PoseGraphSLAM.cpp:1276,1352)
// ------------------------------------------------------------------------------------------------
// Plus: x_plus_delta = [sin|d| d/|d| ; cos|d|] (x) x       (|d| is the HALF angle)
inline void eigen_quaternion_plus(const double* x, const double* delta, double* x_plus_delta) {
    const double norm_delta = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    if (norm_delta > 0.0) {
        const double sin_delta_by_delta = std::sin(norm_delta) / norm_delta;
        const Quat<double> dq{sin_delta_by_delta * delta[0], sin_delta_by_delta * delta[1], sin_delta_by_delta * delta[2], std::cos(norm_delta)};
        const Quat<double> xq{x[0], x[1], x[2], x[3]};
        const Quat<double> r = dq * xq;
        x_plus_delta[0] = r.x; x_plus_delta[1] = r.y; x_plus_delta[2] = r.z; x_plus_delta[3] = r.w;
    } else {
        for (int i = 0; i < 4; ++i) x_plus_delta[i] = x[i];
    }
}
// ComputeJacobian: 4x3 row-major, rows in storage order x,y,z,w
inline void eigen_quaternion_plus_jacobian(const double* x, double* jac) {
    jac[0] = x[3];  jac[1] = x[2];   jac[2] = -x[1];
    jac[3] = -x[2]; jac[4] = x[3];   jac[5] = x[0];
    jac[6] = x[1];  jac[7] = -x[0];  jac[8] = x[3];
    jac[9] = -x[0]; jac[10] = -x[1]; jac[11] = -x[2];
}

}  // namespace orc
