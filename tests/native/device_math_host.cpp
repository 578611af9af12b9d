// Test-only host instantiation of csrc/pgo_device_math.hpp (the per-lane algebra of the K1 kernels) so the
// analytic Jacobians can be checked against tests/golden/ on a machine without a GPU.  Not a product path:
// libpgo never runs this on the CPU.
#include "pgo_device_math.hpp"
using namespace pgo;
extern "C" {
static Pose mk(const double* q, const double* t) { return Pose{q[0], q[1], q[2], q[3], t[0], t[1], t[2]}; }
static Meas mkm(const double* T16, double w) {
    double R[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[r * 3 + c] = T16[c * 4 + r];
    double q[4]; eigen_matrix_to_quat(R, q);
    return Meas{q[0], q[1], q[2], q[3], T16[12], T16[13], T16[14], w};
}
void dm_relpose(const double* q1, const double* t1, const double* q2, const double* t2, const double* T16, double w, double* r, double* J1, double* J2) {
    Meas m = mkm(T16, w);
    relpose_residual<true>(mk(q1, t1), mk(q2, t2), m, w, r, J1, J2);
}
void dm_relpose_cost_only(const double* q1, const double* t1, const double* q2, const double* t2, const double* T16, double w, double* r) {
    Meas m = mkm(T16, w);
    relpose_residual<false>(mk(q1, t1), mk(q2, t2), m, w, r, nullptr, nullptr);
}
void dm_switch(const double* q1, const double* t1, const double* q2, const double* t2, double s, const double* T16, double* r, double* J1, double* J2, double* Js) {
    Meas m = mkm(T16, 1.0);
    switch_residual<true>(mk(q1, t1), mk(q2, t2), m, s, r, J1, J2, Js);
}
void dm_prior(const double* q1, const double* t1, const double* T16, double w, double* r, double* J1) {
    double R[9], q[4], t[3] = {T16[12], T16[13], T16[14]};
    for (int rr = 0; rr < 3; ++rr) for (int c = 0; c < 3; ++c) R[rr * 3 + c] = T16[c * 4 + rr];
    eigen_matrix_to_quat(R, q);
    prior_residual<true>(mk(q1, t1), R, t, q, w, r, J1);
}
// matrix-free operator: contribution of one edge to (J^T J p) at both endpoints, from the compact record
void dm_compact_apply(const double* q1, const double* t1, const double* q2, const double* t2, const double* T16, double ws, int is_switch, double kscale,
                      const double* p1, const double* p2, double* y1, double* y2) {
    Meas m = mkm(T16, ws);
    double rec[COMPACT_DOUBLES];
    edge_compact(mk(q1, t1), mk(q2, t2), m, ws, is_switch != 0, rec);
    compact_apply(rec, 0, p1, p2, kscale, y1);
    compact_apply(rec, 1, p2, p1, kscale, y2);
}
void dm_plus(const double* q, const double* d, double* out) { quat_plus(q, d, out); }
void dm_mat_to_quat(const double* T16, double* q) {
    double R[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[r * 3 + c] = T16[c * 4 + r];
    eigen_matrix_to_quat(R, q);
}
void dm_vio_odometry_record(const double* Mu, const double* Mm, int f, int yaw_weight, double* out8) { vio_odometry_record(Mu, Mm, f, yaw_weight != 0, out8); }
void dm_vio_left_compose(const double* L, const double* Mu, double* q, double* t) { vio_left_compose(L, Mu, q, t); }
}
