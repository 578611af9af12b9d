// pgo_device_math.hpp — closed-form residuals and tangent-space Jacobian blocks of the three cost
// functors, as evaluated per lane by the K1 kernels.  fp64 throughout.
//
// What is replaced (reference file:line):
//   SixDOFError::operator() + AutoDiff<6,4,3,4,3>                         src/CeresResidues.h:32-69,74
//   SixDOFErrorWithSwitchingConstraints::operator() + AutoDiff<7,4,3,4,3,1> src/CeresResidues.h:158-201,206
//   NodePoseRegularization::operator() + AutoDiff<6,4,3>                  src/CeresResidues.h:104-127,131
//   ceres::EigenQuaternionParameterization::{Plus,ComputeJacobian}        src/PoseGraphSLAM.cpp:1276,1352
// The reference differentiates with Jets and projects with the 4x3 parameterization Jacobian; here the
// product (6x4)(4x3) is derived analytically — the left perturbation q <- (d,1) (x) q,
// i.e. R <- (I + 2[d]x) R.  Tangent ordering per pose: [dtheta(3), dt(3)].
//
// Compiles for host too (PGO_HD empty) so the algebra can be checked against tests/golden without a GPU.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define PGO_HD __host__ __device__ __forceinline__
#else
#define PGO_HD inline
#endif

namespace pgo {

struct Pose {            // w_T_c : unit quaternion (x,y,z,w as the reference stores it) + translation
    double qx, qy, qz, qw, tx, ty, tz;
};
struct Meas {            // observed c1_T_c2 : quaternion from Eigen's Matrix3 -> Quaternion rule + translation, and the edge weight
    double qx, qy, qz, qw, tx, ty, tz, w;
};

// R(q) — same polynomial as Eigen::toRotationMatrix (no normalisation), row-major
PGO_HD void quat_to_rot(double x, double y, double z, double w, double* R) {
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// Hamilton product (Eigen coefficient order x,y,z,w)
PGO_HD void quat_mul(const double* a, const double* b, double* r) {
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}

// M(a,b) = d/dd [ vec( a (x) (d,0) (x) b ) ]  (3x3 row-major), a = (av,aw), b = (bv,bw):
//   aw bw I - aw [bv]x - av bv^T + bw [av]x - [av]x [bv]x
PGO_HD void quat_sandwich_jac(const double* a, const double* b, double* M) {
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    const double ab = ax * bx + ay * by + az * bz;          // av . bv
    const double d = aw * bw + ab;                          // diagonal part: aw bw + (av.bv) from -[av]x[bv]x = (av.bv) I - bv av^T
    // -[av]x[bv]x = (av.bv) I - bv av^T ;  total = (aw bw + av.bv) I - av bv^T - bv av^T + [bw av - aw bv]x
    const double cx = bw * ax - aw * bx, cy = bw * ay - aw * by, cz = bw * az - aw * bz;
    M[0] = d - 2.0 * ax * bx;          M[1] = -(ax * by + bx * ay) - cz;  M[2] = -(ax * bz + bx * az) + cy;
    M[3] = -(ay * bx + by * ax) + cz;  M[4] = d - 2.0 * ay * by;          M[5] = -(ay * bz + by * az) - cx;
    M[6] = -(az * bx + bz * ax) - cy;  M[7] = -(az * by + bz * ay) + cx;  M[8] = d - 2.0 * az * bz;
}

// ---------------------------------------------------------------------------------------------
// Relative-pose residual (SixDOFError) at weight w:
//   a = R1 t_o ; v = p1 + a - p2 ; dt = R2^T v ; dq = q2* (x) q1 (x) q_o ; r = w [dt ; 2 dq.vec]
//   dr/d(theta1) = w [ -2 R2^T [a]x ; 2 M ]      dr/d(p1) = w [ R2^T ; 0 ]
//   dr/d(theta2) = w [  2 R2^T [v]x ; -2 M ]     dr/d(p2) = w [ -R2^T ; 0 ]        M = M(q2*, q1 (x) q_o)
// Outputs: r[6]; J1[36], J2[36] row-major 6x6 (cols = [dtheta, dt]).  If !WANT_J only r is written.
// ---------------------------------------------------------------------------------------------
template <bool WANT_J>
PGO_HD void relpose_residual(const Pose& c1, const Pose& c2, const Meas& m, double w, double* r, double* J1, double* J2) {
    double R1[9], R2[9];
    quat_to_rot(c1.qx, c1.qy, c1.qz, c1.qw, R1);
    quat_to_rot(c2.qx, c2.qy, c2.qz, c2.qw, R2);
    const double a0 = R1[0] * m.tx + R1[1] * m.ty + R1[2] * m.tz;
    const double a1 = R1[3] * m.tx + R1[4] * m.ty + R1[5] * m.tz;
    const double a2 = R1[6] * m.tx + R1[7] * m.ty + R1[8] * m.tz;
    const double v0 = c1.tx + a0 - c2.tx, v1 = c1.ty + a1 - c2.ty, v2 = c1.tz + a2 - c2.tz;
    // dt = R2^T v
    const double d0 = R2[0] * v0 + R2[3] * v1 + R2[6] * v2;
    const double d1 = R2[1] * v0 + R2[4] * v1 + R2[7] * v2;
    const double d2 = R2[2] * v0 + R2[5] * v1 + R2[8] * v2;
    const double q1[4] = {c1.qx, c1.qy, c1.qz, c1.qw};
    const double qo[4] = {m.qx, m.qy, m.qz, m.qw};
    const double q2c[4] = {-c2.qx, -c2.qy, -c2.qz, c2.qw};
    double b[4], dq[4];
    quat_mul(q1, qo, b);
    quat_mul(q2c, b, dq);
    r[0] = w * d0; r[1] = w * d1; r[2] = w * d2;
    r[3] = w * 2.0 * dq[0]; r[4] = w * 2.0 * dq[1]; r[5] = w * 2.0 * dq[2];
    if (WANT_J) {
        // a' = R2^T a  ->  R2^T [a]x = [a']x R2^T ;  R2^T [v]x = [dt]x R2^T
        const double p0 = R2[0] * a0 + R2[3] * a1 + R2[6] * a2;
        const double p1 = R2[1] * a0 + R2[4] * a1 + R2[7] * a2;
        const double p2 = R2[2] * a0 + R2[5] * a1 + R2[8] * a2;
        double M[9];
        quat_sandwich_jac(q2c, b, M);
        const double w2 = 2.0 * w;
        // rows 0..2 of [x]x R2^T :  row i = e_i^T [x]x R2^T ; ([x]x R2^T)(i,j) = sum_k [x]x(i,k) R2(j,k)
#define PGO_CROSS_RT(X0, X1, X2, i, j) \
        ((i) == 0 ? (-(X2) * R2[(j) * 3 + 1] + (X1) * R2[(j) * 3 + 2]) : (i) == 1 ? ((X2) * R2[(j) * 3 + 0] - (X0) * R2[(j) * 3 + 2]) : (-(X1) * R2[(j) * 3 + 0] + (X0) * R2[(j) * 3 + 1]))
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double rt = R2[j * 3 + i];                         // R2^T(i,j)
                J1[i * 6 + j] = -w2 * PGO_CROSS_RT(p0, p1, p2, i, j);    // -2 w [a']x R2^T
                J1[i * 6 + 3 + j] = w * rt;
                J2[i * 6 + j] = w2 * PGO_CROSS_RT(d0, d1, d2, i, j);     //  2 w [dt]x R2^T
                J2[i * 6 + 3 + j] = -w * rt;
                J1[(3 + i) * 6 + j] = w2 * M[i * 3 + j];
                J1[(3 + i) * 6 + 3 + j] = 0.0;
                J2[(3 + i) * 6 + j] = -w2 * M[i * 3 + j];
                J2[(3 + i) * 6 + 3 + j] = 0.0;
            }
        }
#undef PGO_CROSS_RT
    }
}

// ---------------------------------------------------------------------------------------------
// Switchable residual (SixDOFErrorWithSwitchingConstraints): with r6, A1, A2 the relative-pose residual
// and blocks at w = 1 (the edge weight is ignored exactly as CeresResidues.h:198 ignores it):
//   r = [ s r6 ; s (1 - s) ]   J1 = s A1, J2 = s A2 (7th row zero, not stored)   dr/ds = [ r6 ; 1 - 2 s ]
// ---------------------------------------------------------------------------------------------
template <bool WANT_J>
PGO_HD void switch_residual(const Pose& c1, const Pose& c2, const Meas& m, double s, double* r7, double* J1, double* J2, double* Js7) {
    double r6[6];
    relpose_residual<WANT_J>(c1, c2, m, 1.0, r6, J1, J2);
#pragma unroll
    for (int i = 0; i < 6; ++i) r7[i] = s * r6[i];
    r7[6] = s * (1.0 - s);
    if (WANT_J) {
#pragma unroll
        for (int i = 0; i < 36; ++i) { J1[i] *= s; J2[i] *= s; }
#pragma unroll
        for (int i = 0; i < 6; ++i) Js7[i] = r6[i];
        Js7[6] = 1.0 - 2.0 * s;
    }
}

// ---------------------------------------------------------------------------------------------
// Node regulariser (NodePoseRegularization) with target f = (R_f, t_f) given as a rigid Matrix4d and
// q_f = Eigen quaternion of R_f:   delta = f^-1 [R(q1) p1; 0 1]
//   r = w [ R_f^T (p1 - t_f) ; 2 sigma (q_f* (x) q1).vec ],  sigma = sign chosen by Eigen's Matrix3->Quaternion
//   branch rule applied to R_delta (w > 0 when trace > 0, else the largest-diagonal component > 0).
//   dr/dp1 = w [ R_f^T ; 0 ]     dr/dtheta1 = w [ 0 ; 2 sigma M(q_f*, q1) ]
// Rf is row-major 3x3.
// ---------------------------------------------------------------------------------------------
template <bool WANT_J>
PGO_HD void prior_residual(const Pose& c1, const double* Rf, const double* tf, const double* qf, double w, double* r, double* J1) {
    const double e0 = c1.tx - tf[0], e1 = c1.ty - tf[1], e2 = c1.tz - tf[2];
    r[0] = w * (Rf[0] * e0 + Rf[3] * e1 + Rf[6] * e2);
    r[1] = w * (Rf[1] * e0 + Rf[4] * e1 + Rf[7] * e2);
    r[2] = w * (Rf[2] * e0 + Rf[5] * e1 + Rf[8] * e2);
    const double qfc[4] = {-qf[0], -qf[1], -qf[2], qf[3]};
    const double q1[4] = {c1.qx, c1.qy, c1.qz, c1.qw};
    double dq[4];
    quat_mul(qfc, q1, dq);
    // Eigen branch rule on R_delta = R(dq): trace = 3 - 4 |v|^2 ; diag_i = 1 - 2 (|v|^2 - v_i^2)
    const double xx = dq[0] * dq[0], yy = dq[1] * dq[1], zz = dq[2] * dq[2];
    const double trace = 3.0 - 4.0 * (xx + yy + zz);
    double sigma;
    if (trace > 0.0) sigma = dq[3] >= 0.0 ? 1.0 : -1.0;
    else {
        const double m00 = 1.0 - 2.0 * (yy + zz), m11 = 1.0 - 2.0 * (xx + zz), m22 = 1.0 - 2.0 * (xx + yy);
        int i = 0; double mii = m00;
        if (m11 > m00) { i = 1; mii = m11; }
        if (m22 > mii) { i = 2; }
        sigma = dq[i] >= 0.0 ? 1.0 : -1.0;
    }
    const double w2 = 2.0 * w * sigma;
    r[3] = w2 * dq[0]; r[4] = w2 * dq[1]; r[5] = w2 * dq[2];
    if (WANT_J) {
        double M[9];
        quat_sandwich_jac(qfc, q1, M);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                J1[i * 6 + j] = 0.0;
                J1[i * 6 + 3 + j] = w * Rf[j * 3 + i];
                J1[(3 + i) * 6 + j] = w2 * M[i * 3 + j];
                J1[(3 + i) * 6 + 3 + j] = 0.0;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Matrix-free normal-equation operator.  The two Jacobian blocks of an edge are functions of 15 numbers:
//   q2 (4), b = q1 (x) q_o (4), a' = R2^T R1 t_o (3), dt = R2^T v (3, unweighted), ws (edge weight w, or the switch value s)
//   J1 = ws [ -2 [a']x R2^T , R2^T ; 2 M , 0 ]      J2 = ws [ 2 [dt]x R2^T , -R2^T ; -2 M , 0 ]      M = M(q2*, b)
// (+ r6 = [dt ; 2 (q2* (x) b).vec] for switchable edges, whose Schur term is  u <- u - k (k.u),  k = r6 sqrt(1/(Js^T Js + lambda_s))).
// compact_apply returns this edge's contribution to (J^T J p) at ONE endpoint: y = J_side^T (J1 p1 + J2 p2).
// ---------------------------------------------------------------------------------------------
constexpr int COMPACT_DOUBLES = 22;   // q2[4] b[4] ap[3] dt[3] | ws pad | r6[6]   (rec[14] = ws, rec[16..21] = r6)

PGO_HD void edge_compact(const Pose& c1, const Pose& c2, const Meas& m, double ws, bool want_r6, double* rec) {
    double R1[9], R2[9];
    quat_to_rot(c1.qx, c1.qy, c1.qz, c1.qw, R1);
    quat_to_rot(c2.qx, c2.qy, c2.qz, c2.qw, R2);
    const double a0 = R1[0] * m.tx + R1[1] * m.ty + R1[2] * m.tz;
    const double a1 = R1[3] * m.tx + R1[4] * m.ty + R1[5] * m.tz;
    const double a2 = R1[6] * m.tx + R1[7] * m.ty + R1[8] * m.tz;
    const double v0 = c1.tx + a0 - c2.tx, v1 = c1.ty + a1 - c2.ty, v2 = c1.tz + a2 - c2.tz;
    const double q1[4] = {c1.qx, c1.qy, c1.qz, c1.qw};
    const double qo[4] = {m.qx, m.qy, m.qz, m.qw};
    double b[4];
    quat_mul(q1, qo, b);
    rec[0] = c2.qx; rec[1] = c2.qy; rec[2] = c2.qz; rec[3] = c2.qw;
    rec[4] = b[0]; rec[5] = b[1]; rec[6] = b[2]; rec[7] = b[3];
    rec[8] = R2[0] * a0 + R2[3] * a1 + R2[6] * a2;
    rec[9] = R2[1] * a0 + R2[4] * a1 + R2[7] * a2;
    rec[10] = R2[2] * a0 + R2[5] * a1 + R2[8] * a2;
    rec[11] = R2[0] * v0 + R2[3] * v1 + R2[6] * v2;
    rec[12] = R2[1] * v0 + R2[4] * v1 + R2[7] * v2;
    rec[13] = R2[2] * v0 + R2[5] * v1 + R2[8] * v2;
    rec[14] = ws;
    if (want_r6) {
        const double q2c[4] = {-c2.qx, -c2.qy, -c2.qz, c2.qw};
        double dq[4];
        quat_mul(q2c, b, dq);
        rec[16] = rec[11]; rec[17] = rec[12]; rec[18] = rec[13];
        rec[19] = 2.0 * dq[0]; rec[20] = 2.0 * dq[1]; rec[21] = 2.0 * dq[2];
    } else {
        rec[16] = 0.0; rec[17] = 0.0; rec[18] = 0.0; rec[19] = 0.0; rec[20] = 0.0; rec[21] = 0.0;
    }
    rec[15] = 0.0;
}

// side 0: own = c1, other = c2.  side 1: own = c2, other = c1.  kscale = sqrt(a_inv) for switchable edges, 0 otherwise.
// Neither R2 nor M is formed: rotations act through the quaternion (v' = v + w t + t x qv, t = 2 v x qv for R2^T; mirrored for R2) and
// M = d I - av bv^T - bv av^T + [c]x through its three vectors — 18 fewer live doubles than holding two 3x3 matrices.  Measured on MI355X
// (C3): the same 28 us per matvec as the matrix form at 4 wavefronts/SIMD; forcing 5 or 6 (amdgpu_waves_per_eu) still spills 36 / 108 B
// per lane because the record and both endpoint vectors are in flight at once, and runs 35 / 46 us.
PGO_HD void rot_conj(const double* q, double v0, double v1, double v2, double& o0, double& o1, double& o2) {   // R(q)^T v
    const double t0 = 2.0 * (v1 * q[2] - v2 * q[1]), t1 = 2.0 * (v2 * q[0] - v0 * q[2]), t2 = 2.0 * (v0 * q[1] - v1 * q[0]);
    o0 = v0 + q[3] * t0 + (t1 * q[2] - t2 * q[1]);
    o1 = v1 + q[3] * t1 + (t2 * q[0] - t0 * q[2]);
    o2 = v2 + q[3] * t2 + (t0 * q[1] - t1 * q[0]);
}
PGO_HD void rot_fwd(const double* q, double v0, double v1, double v2, double& o0, double& o1, double& o2) {    // R(q) v
    const double t0 = 2.0 * (q[1] * v2 - q[2] * v1), t1 = 2.0 * (q[2] * v0 - q[0] * v2), t2 = 2.0 * (q[0] * v1 - q[1] * v0);
    o0 = v0 + q[3] * t0 + (q[1] * t2 - q[2] * t1);
    o1 = v1 + q[3] * t1 + (q[2] * t0 - q[0] * t2);
    o2 = v2 + q[3] * t2 + (q[0] * t1 - q[1] * t0);
}
// u = W (J1 p1 + J2 p2) (+ the switch Schur term): the part of an edge's product that both endpoints share
PGO_HD void compact_u(const double* rec, const double* p1, const double* p2, double kscale, double* u) {
    const double* q2 = rec;          // (x, y, z, w)
    const double* b = rec + 4;
    const double ap0 = rec[8], ap1 = rec[9], ap2 = rec[10], d0 = rec[11], d1 = rec[12], d2 = rec[13], ws = rec[14];
    // g1 = R2^T theta1, g2 = R2^T theta2, f = R2^T (tau1 - tau2)
    double g10, g11, g12, g20, g21, g22, f0, f1, f2;
    rot_conj(q2, p1[0], p1[1], p1[2], g10, g11, g12);
    rot_conj(q2, p2[0], p2[1], p2[2], g20, g21, g22);
    rot_conj(q2, p1[3] - p2[3], p1[4] - p2[4], p1[5] - p2[5], f0, f1, f2);
    // u_t = ws ( f + 2 (dt x g2 - a' x g1) ),  u_q = 2 ws M (theta1 - theta2)
    u[0] = ws * (f0 + 2.0 * ((d1 * g22 - d2 * g21) - (ap1 * g12 - ap2 * g11)));
    u[1] = ws * (f1 + 2.0 * ((d2 * g20 - d0 * g22) - (ap2 * g10 - ap0 * g12)));
    u[2] = ws * (f2 + 2.0 * ((d0 * g21 - d1 * g20) - (ap0 * g11 - ap1 * g10)));
    // M(a, b) with a = conj(q2): av = -q2.vec, aw = q2.w;  M = dd I - av bv^T - bv av^T + [c]x,  c = bw av - aw bv
    const double av0 = -q2[0], av1 = -q2[1], av2 = -q2[2], aw = q2[3];
    const double dd = aw * b[3] + (av0 * b[0] + av1 * b[1] + av2 * b[2]);
    const double c0 = b[3] * av0 - aw * b[0], c1 = b[3] * av1 - aw * b[1], c2 = b[3] * av2 - aw * b[2];
    {
        const double e0 = p1[0] - p2[0], e1 = p1[1] - p2[1], e2 = p1[2] - p2[2];
        const double sb = b[0] * e0 + b[1] * e1 + b[2] * e2, sa = av0 * e0 + av1 * e1 + av2 * e2;
        const double w2 = 2.0 * ws;
        u[3] = w2 * (dd * e0 - av0 * sb - b[0] * sa + (c1 * e2 - c2 * e1));
        u[4] = w2 * (dd * e1 - av1 * sb - b[1] * sa + (c2 * e0 - c0 * e2));
        u[5] = w2 * (dd * e2 - av2 * sb - b[2] * sa + (c0 * e1 - c1 * e0));
    }
    if (kscale != 0.0) {
        double k[6], d = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) { k[i] = rec[16 + i] * kscale; d += k[i] * u[i]; }
#pragma unroll
        for (int i = 0; i < 6; ++i) u[i] -= k[i] * d;
    }
}
// y = J_side^T u
PGO_HD void compact_side(const double* rec, const double* u, int side, double* y) {
    const double* q2 = rec;
    const double* b = rec + 4;
    const double ap0 = rec[8], ap1 = rec[9], ap2 = rec[10], d0 = rec[11], d1 = rec[12], d2 = rec[13], ws = rec[14];
    const double av0 = -q2[0], av1 = -q2[1], av2 = -q2[2], aw = q2[3];
    const double dd = aw * b[3] + (av0 * b[0] + av1 * b[1] + av2 * b[2]);
    const double c0 = b[3] * av0 - aw * b[0], c1 = b[3] * av1 - aw * b[1], c2 = b[3] * av2 - aw * b[2];
    // x = (a' or dt) x u_t ; w3 = R2 x + M^T u_q ;  M^T u = dd u - bv (av.u) - av (bv.u) - c x u
    const double s0 = side ? d0 : ap0, s1 = side ? d1 : ap1, s2 = side ? d2 : ap2;
    const double x0 = s1 * u[2] - s2 * u[1], x1 = s2 * u[0] - s0 * u[2], x2 = s0 * u[1] - s1 * u[0];
    const double ua = av0 * u[3] + av1 * u[4] + av2 * u[5], ub = b[0] * u[3] + b[1] * u[4] + b[2] * u[5];
    const double m0 = dd * u[3] - b[0] * ua - av0 * ub - (c1 * u[5] - c2 * u[4]);
    const double m1 = dd * u[4] - b[1] * ua - av1 * ub - (c2 * u[3] - c0 * u[5]);
    const double m2 = dd * u[5] - b[2] * ua - av2 * ub - (c0 * u[4] - c1 * u[3]);
    const double sg = side ? -ws : ws;
    double rx0, rx1, rx2, ru0, ru1, ru2;
    rot_fwd(q2, x0, x1, x2, rx0, rx1, rx2);
    rot_fwd(q2, u[0], u[1], u[2], ru0, ru1, ru2);
    y[0] = 2.0 * sg * (rx0 + m0);
    y[1] = 2.0 * sg * (rx1 + m1);
    y[2] = 2.0 * sg * (rx2 + m2);
    y[3] = sg * ru0;
    y[4] = sg * ru1;
    y[5] = sg * ru2;
}
PGO_HD void compact_apply(const double* rec, int side, const double* p_own, const double* p_other, double kscale, double* y) {
    double u[6];
    compact_u(rec, side ? p_other : p_own, side ? p_own : p_other, kscale, u);
    compact_side(rec, u, side, y);
}
// both endpoints of one edge from ONE evaluation of u (the lane of an edge whose two keyframes sit in the same matvec tile): the same
// expressions as two compact_apply calls, so the results are bit-identical to them; what the two sides share (M^T u_q, R2 u_t) is computed once
PGO_HD void compact_apply_both(const double* rec, const double* p1, const double* p2, double kscale, double* y1, double* y2) {
    double u[6];
    compact_u(rec, p1, p2, kscale, u);
    compact_side(rec, u, 0, y1);
    compact_side(rec, u, 1, y2);
}

// ceres::EigenQuaternionParameterization::Plus:  q+ = [sin|d| d/|d| ; cos|d|] (x) q
PGO_HD void quat_plus(const double* q, const double* d, double* out) {
    const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (n > 0.0) {
        const double sbd = sin(n) / n;
        const double dq[4] = {sbd * d[0], sbd * d[1], sbd * d[2], cos(n)};
        quat_mul(dq, q, out);
    } else {
        out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
    }
}

// Eigen's `Quaterniond(Matrix3d)` (CeresResidues.h:24,150; PoseManipUtils.cpp:87-98): R row-major -> q (x,y,z,w).  Restated from
// Eigen's published algorithm (branch on trace / largest diagonal); the three non-trace cases are written out so that device code
// needs no dynamically indexed registers.
PGO_HD void eigen_matrix_to_quat(const double* R, double* q) {
    double t = R[0] + R[4] + R[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t;
        q[1] = (R[2] - R[6]) * t;
        q[2] = (R[3] - R[1]) * t;
    } else if (R[0] >= R[4] && R[0] >= R[8]) {          // i=0, j=1, k=2   (Eigen: i=1 only if R11 > R00; i=2 only if R22 > R[i][i])
        t = sqrt(R[0] - R[4] - R[8] + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[7] - R[5]) * t;
        q[1] = (R[3] + R[1]) * t;
        q[2] = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && R[4] >= R[8]) {           // i=1, j=2, k=0
        t = sqrt(R[4] - R[8] - R[0] + 1.0);
        q[1] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[2] - R[6]) * t;
        q[2] = (R[7] + R[5]) * t;
        q[0] = (R[1] + R[3]) * t;
    } else {                                            // i=2, j=0, k=1
        t = sqrt(R[8] - R[0] - R[4] + 1.0);
        q[2] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[3] - R[1]) * t;
        q[0] = (R[2] + R[6]) * t;
        q[1] = (R[5] + R[7]) * t;
    }
}

// ---- graph construction from raw VIO poses (reference src/PoseGraphSLAM.cpp:1597-1606, :1770-1778) ----
// Matrix4d inputs are column-major 16 doubles with bottom row (0,0,0,1), as Eigen stores them.

// u_M_umf = w_M_u^-1 * w_M_umf  (:1597-1599).  The reference calls the general Matrix4d::inverse(); for an affine matrix that is
// [A^-1, -A^-1 t], with A^-1 by cofactors as Eigen's fixed-size inverse computes it — also right for a not-quite-orthonormal A.
// Outputs R (row-major 3x3) and t of the product.
PGO_HD void vio_relative_pose(const double* Mu, const double* Mm, double* R, double* t) {
    const double a00 = Mu[0], a10 = Mu[1], a20 = Mu[2], a01 = Mu[4], a11 = Mu[5], a21 = Mu[6], a02 = Mu[8], a12 = Mu[9], a22 = Mu[10];
    const double c00 = a11 * a22 - a12 * a21, c01 = a02 * a21 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const double c10 = a12 * a20 - a10 * a22, c11 = a00 * a22 - a02 * a20, c12 = a02 * a10 - a00 * a12;
    const double c20 = a10 * a21 - a11 * a20, c21 = a01 * a20 - a00 * a21, c22 = a00 * a11 - a01 * a10;
    const double idet = 1.0 / (a00 * c00 + a01 * c10 + a02 * c20);
    const double I[9] = {c00 * idet, c01 * idet, c02 * idet, c10 * idet, c11 * idet, c12 * idet, c20 * idet, c21 * idet, c22 * idet};   // A^-1 row-major
    const double d0 = Mm[12] - Mu[12], d1 = Mm[13] - Mu[13], d2 = Mm[14] - Mu[14];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) R[r * 3 + c] = I[r * 3] * Mm[c * 4] + I[r * 3 + 1] * Mm[c * 4 + 1] + I[r * 3 + 2] * Mm[c * 4 + 2];
        t[r] = I[r * 3] * d0 + I[r * 3 + 1] * d1 + I[r * 3 + 2] * d2;
    }
}

// odometry edge record (q_obs xyzw, t_obs, weight): weight = 0.9^f * exp(-yaw^2/6), yaw = R2ypr(R)(0) in DEGREES
// (:1603-1606; PoseManipUtils.cpp:143-158: y = atan2(n(1), n(0)) with n = R.col(0))
PGO_HD void vio_odometry_record(const double* Mu, const double* Mm, int f, bool yaw_weight, double* out8) {
    double R[9], t[3];
    vio_relative_pose(Mu, Mm, R, t);
    eigen_matrix_to_quat(R, out8);
    out8[4] = t[0]; out8[5] = t[1]; out8[6] = t[2];
    double w = pow(0.9, (double)f);
    if (yaw_weight) {
        const double yaw = atan2(R[3], R[0]) / 3.14159265358979323846 * 180.0;
        w *= exp(-yaw * yaw / 6.0);
    }
    out8[7] = w;
}

// initial guess of a not-yet-solved keyframe (:1770-1778): pose = L * w_M_u with L = w_T_last * w_M_last^-1 (or wset_T_w), stored as
// (xyzw, t) the way update_opt_variable_with -> eigenmat_to_raw_xyzw does
PGO_HD void vio_left_compose(const double* L, const double* Mu, double* q, double* t) {
    double R[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) R[r * 3 + c] = L[r] * Mu[c * 4] + L[4 + r] * Mu[c * 4 + 1] + L[8 + r] * Mu[c * 4 + 2];
        t[r] = L[r] * Mu[12] + L[4 + r] * Mu[13] + L[8 + r] * Mu[14] + L[12 + r];
    }
    eigen_matrix_to_quat(R, q);
}

}  // namespace pgo
