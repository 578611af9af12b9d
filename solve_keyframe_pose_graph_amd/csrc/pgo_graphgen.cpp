// pgo_graphgen.cpp — synthetic keyframe pose graphs; see include/pgo_graphgen.h.
// Host-only (g++).  Mirrors what the reference's caller layer feeds the solver trigger
// (reference src/PoseGraphSLAM.cpp:1381-1386,1459-1464,1550-1556,1570-1633,1770-1783,1817-1849).
#include "pgo_graphgen.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>
#include <unordered_map>
#include <vector>

namespace {

struct M3 { double m[9]; };   // row-major
struct V3 { double x, y, z; };
struct SE3 { M3 R; V3 t; };

inline M3 m3_identity() { M3 r{}; r.m[0] = r.m[4] = r.m[8] = 1.0; return r; }
inline M3 m3_mul(const M3& a, const M3& b) {
    M3 c{};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += a.m[i * 3 + k] * b.m[k * 3 + j]; c.m[i * 3 + j] = s; }
    return c;
}
inline M3 m3_t(const M3& a) { M3 c{}; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c.m[i * 3 + j] = a.m[j * 3 + i]; return c; }
inline V3 m3_v(const M3& a, const V3& v) { return V3{a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z, a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z}; }
inline V3 operator+(const V3& a, const V3& b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(const V3& a, const V3& b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline SE3 se3_mul(const SE3& a, const SE3& b) { return SE3{m3_mul(a.R, b.R), m3_v(a.R, b.t) + a.t}; }
inline SE3 se3_inv(const SE3& a) { M3 Rt = m3_t(a.R); V3 t = m3_v(Rt, a.t); return SE3{Rt, V3{-t.x, -t.y, -t.z}}; }
inline SE3 se3_identity() { return SE3{m3_identity(), V3{0, 0, 0}}; }

// Rodrigues: exp([w]x)
inline M3 so3_exp(const V3& w) {
    const double th = std::sqrt(w.x * w.x + w.y * w.y + w.z * w.z);
    if (th < 1e-300) return m3_identity();
    const double kx = w.x / th, ky = w.y / th, kz = w.z / th, c = std::cos(th), s = std::sin(th), v = 1 - c;
    M3 R;
    R.m[0] = c + kx * kx * v;      R.m[1] = kx * ky * v - kz * s; R.m[2] = kx * kz * v + ky * s;
    R.m[3] = ky * kx * v + kz * s; R.m[4] = c + ky * ky * v;      R.m[5] = ky * kz * v - kx * s;
    R.m[6] = kz * kx * v - ky * s; R.m[7] = kz * ky * v + kx * s; R.m[8] = c + kz * kz * v;
    return R;
}

// Shepperd-style rotation -> unit quaternion (x,y,z,w), w >= 0.  (Only used for the pose ARRAYS handed to the
// solver, where any sign is a valid representation.)
inline void m3_to_quat(const M3& R, double* q) {
    const double tr = R.m[0] + R.m[4] + R.m[8];
    double x, y, z, w;
    if (tr > 0) { double s = std::sqrt(tr + 1.0) * 2; w = 0.25 * s; x = (R.m[7] - R.m[5]) / s; y = (R.m[2] - R.m[6]) / s; z = (R.m[3] - R.m[1]) / s; }
    else if (R.m[0] > R.m[4] && R.m[0] > R.m[8]) { double s = std::sqrt(1.0 + R.m[0] - R.m[4] - R.m[8]) * 2; w = (R.m[7] - R.m[5]) / s; x = 0.25 * s; y = (R.m[1] + R.m[3]) / s; z = (R.m[2] + R.m[6]) / s; }
    else if (R.m[4] > R.m[8]) { double s = std::sqrt(1.0 + R.m[4] - R.m[0] - R.m[8]) * 2; w = (R.m[2] - R.m[6]) / s; x = (R.m[1] + R.m[3]) / s; y = 0.25 * s; z = (R.m[5] + R.m[7]) / s; }
    else { double s = std::sqrt(1.0 + R.m[8] - R.m[0] - R.m[4]) * 2; w = (R.m[3] - R.m[1]) / s; x = (R.m[2] + R.m[6]) / s; y = (R.m[5] + R.m[7]) / s; z = 0.25 * s; }
    const double n = std::sqrt(x * x + y * y + z * z + w * w);
    if (w < 0) { x = -x; y = -y; z = -z; w = -w; }
    q[0] = x / n; q[1] = y / n; q[2] = z / n; q[3] = w / n;
}
inline void se3_to_colmajor(const SE3& T, double* d) {
    for (int c = 0; c < 3; ++c) { for (int r = 0; r < 3; ++r) d[c * 4 + r] = T.R.m[r * 3 + c]; d[c * 4 + 3] = 0.0; }
    d[12] = T.t.x; d[13] = T.t.y; d[14] = T.t.z; d[15] = 1.0;
}
// re-orthonormalise (Gram-Schmidt) so that long chains stay rotations
inline void m3_orthonormalize(M3& R) {
    V3 a{R.m[0], R.m[3], R.m[6]}, b{R.m[1], R.m[4], R.m[7]};
    double n = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); a = V3{a.x / n, a.y / n, a.z / n};
    double d = a.x * b.x + a.y * b.y + a.z * b.z; b = V3{b.x - d * a.x, b.y - d * a.y, b.z - d * a.z};
    n = std::sqrt(b.x * b.x + b.y * b.y + b.z * b.z); b = V3{b.x / n, b.y / n, b.z / n};
    V3 c{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    R.m[0] = a.x; R.m[3] = a.y; R.m[6] = a.z; R.m[1] = b.x; R.m[4] = b.y; R.m[7] = b.z; R.m[2] = c.x; R.m[5] = c.y; R.m[8] = c.z;
}

struct Rng {
    std::mt19937_64 g;
    explicit Rng(uint64_t s) : g(s) {}
    double uniform() { return (double)(g() >> 11) * (1.0 / 9007199254740992.0); }   // [0,1)
    uint64_t below(uint64_t n) { return (uint64_t)(uniform() * (double)n) % n; }
    double normal() { double u1 = uniform(), u2 = uniform(); if (u1 < 1e-300) u1 = 1e-300; return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925286766559 * u2); }
};

// yaw in DEGREES exactly as reference src/utils/PoseManipUtils.cpp:143-158 (first component of R2ypr)
inline double yaw_deg(const M3& R) { return std::atan2(R.m[3], R.m[0]) / M_PI * 180.0; }

}  // namespace

struct pgo_gen_graph {
    pgo_gen_config cfg;
    std::vector<SE3> truth, vio, init;
    std::vector<int32_t> world;
    std::vector<int32_t> o_c1, o_c2; std::vector<SE3> o_T; std::vector<double> o_w;
    std::vector<int32_t> l_c1, l_c2, l_out; std::vector<SE3> l_T; std::vector<double> l_w;
    std::vector<int32_t> r_node; std::vector<SE3> r_T; std::vector<double> r_w;
};

extern "C" {

void pgo_gen_config_init(pgo_gen_config* c) {
    std::memset(c, 0, sizeof(*c));
    c->n_poses = 200; c->n_loops = 20; c->odom_f_max = 1; c->apply_yaw_weight = 0; c->n_worlds = 1;
    c->inter_world_frac = 0.25; c->outlier_frac = 0.10;
    c->odom_sigma_t = 0.005; c->odom_sigma_r = 0.0002; c->loop_sigma_t = 0.04; c->loop_sigma_r = 0.02;
    c->box_scale = 1.5; c->seed = 1;
    c->turn_deg_per_keyframe = 2.0; c->loop_radius = 1.5; c->straight_min = 5; c->straight_max = 25; c->min_loop_gap = 50;
}

static void generate_truth(const pgo_gen_config& c, double box_scale, Rng& rng, std::vector<SE3>& truth) {
    const int64_t N = c.n_poses;
    truth.resize(N);
    const double side = std::max(8.0, box_scale * std::cbrt((double)N));
    const double half = side / 2;
    SE3 cur = se3_identity();
    const int smin = std::max(1, c.straight_min), srange = std::max(1, c.straight_max - c.straight_min + 1);
    int straight_left = smin + (int)rng.below(srange);
    int turn_left = 0;
    V3 turn_axis{0, 0, 1};
    const int turn_steps = std::max(1, (int)std::lround(90.0 / c.turn_deg_per_keyframe));
    const double dth = (M_PI / 2) / turn_steps;   // default 2 degrees per keyframe, 45 keyframes per 90-degree turn
    for (int64_t k = 0; k < N; ++k) {
        truth[k] = cur;
        if (turn_left > 0) {
            M3 dR = so3_exp(V3{turn_axis.x * dth, turn_axis.y * dth, turn_axis.z * dth});
            cur.R = m3_mul(cur.R, dR);
            if ((k & 63) == 0) m3_orthonormalize(cur.R);
            --turn_left;
            if (turn_left == 0) straight_left = smin + (int)rng.below(srange);
        } else {
            --straight_left;
            if (straight_left <= 0) {
                // choose among +-body-y, +-body-z; outside the box take the one that heads back to the centre
                const V3 cand[4] = {{0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
                int pick = (int)rng.below(4);
                const bool outside = std::fabs(cur.t.x) > half || std::fabs(cur.t.y) > half || std::fabs(cur.t.z) > half;
                if (outside) {
                    double best = -1e300;
                    for (int q = 0; q < 4; ++q) {
                        M3 R90 = m3_mul(cur.R, so3_exp(V3{cand[q].x * M_PI / 2, cand[q].y * M_PI / 2, cand[q].z * M_PI / 2}));
                        const V3 h{R90.m[0], R90.m[3], R90.m[6]};
                        const double sc = -(h.x * cur.t.x + h.y * cur.t.y + h.z * cur.t.z);
                        if (sc > best) { best = sc; pick = q; }
                    }
                }
                turn_axis = cand[pick];
                turn_left = turn_steps;
            }
        }
        // step 1 m along body x
        cur.t = cur.t + V3{cur.R.m[0], cur.R.m[3], cur.R.m[6]};
    }
}

int pgo_gen_create(const pgo_gen_config* cfg, pgo_gen_graph** out) {
    if (!cfg || !out || cfg->n_poses < 2 || cfg->n_worlds < 1 || cfg->odom_f_max < 1) return -1;
    pgo_gen_graph* G = new pgo_gen_graph();
    G->cfg = *cfg;
    const pgo_gen_config& c = G->cfg;
    const int64_t N = c.n_poses;
    const int W = c.n_worlds;
    const int64_t per_world = (N + W - 1) / W;

    // ---- ground truth + spatial loop candidates; shrink the box until there are enough candidates
    std::vector<std::pair<int32_t, int32_t>> cand;  // (a newer, b older)
    double box = c.box_scale;
    for (int attempt = 0; attempt < 12; ++attempt) {
        Rng rng(c.seed * 0x9E3779B97F4A7C15ull + 12345 + attempt);
        generate_truth(c, box, rng, G->truth);
        cand.clear();
        const double rad = c.loop_radius, cell = c.loop_radius;
        std::unordered_map<uint64_t, std::vector<int32_t>> grid;
        grid.reserve((size_t)N);
        auto key = [&](int64_t ix, int64_t iy, int64_t iz) { return (uint64_t)((ix + 1048576) & 0x1FFFFF) | ((uint64_t)((iy + 1048576) & 0x1FFFFF) << 21) | ((uint64_t)((iz + 1048576) & 0x1FFFFF) << 42); };
        for (int64_t a = 0; a < N; ++a) {
            const V3& p = G->truth[a].t;
            const int64_t ix = (int64_t)std::floor(p.x / cell), iy = (int64_t)std::floor(p.y / cell), iz = (int64_t)std::floor(p.z / cell);
            for (int dx = -1; dx <= 1; ++dx) for (int dy = -1; dy <= 1; ++dy) for (int dz = -1; dz <= 1; ++dz) {
                auto it = grid.find(key(ix + dx, iy + dy, iz + dz));
                if (it == grid.end()) continue;
                for (int32_t b : it->second) {
                    if (a - b <= c.min_loop_gap) continue;
                    const V3 d = p - G->truth[b].t;
                    if (d.x * d.x + d.y * d.y + d.z * d.z < rad * rad) cand.push_back({(int32_t)a, b});
                }
            }
            grid[key(ix, iy, iz)].push_back((int32_t)a);
        }
        if ((int64_t)cand.size() >= c.n_loops) break;
        box *= 0.85;
    }
    std::sort(cand.begin(), cand.end());
    Rng rng(c.seed * 0xD1B54A32D192ED03ull + 777);

    // ---- worlds + VIO chains (each world's VIO starts at identity in its own frame)
    G->world.resize(N);
    G->vio.resize(N);
    for (int64_t k = 0; k < N; ++k) {
        const int w = (int)std::min<int64_t>(k / per_world, W - 1);
        G->world[k] = w;
        if (k == 0 || G->world[k - 1] != w) { G->vio[k] = se3_identity(); continue; }
        SE3 rel = se3_mul(se3_inv(G->truth[k - 1]), G->truth[k]);   // true motion k-1 -> k
        SE3 noise{so3_exp(V3{rng.normal() * c.odom_sigma_r, rng.normal() * c.odom_sigma_r, rng.normal() * c.odom_sigma_r}),
                  V3{rng.normal() * c.odom_sigma_t, rng.normal() * c.odom_sigma_t, rng.normal() * c.odom_sigma_t}};
        G->vio[k] = se3_mul(G->vio[k - 1], se3_mul(rel, noise));
        if ((k & 63) == 0) m3_orthonormalize(G->vio[k].R);
    }

    // ---- odometry edges exactly as the trigger builds them (src/PoseGraphSLAM.cpp:1570-1633); within a world only here
    for (int64_t u = 0; u < N; ++u) {
        for (int f = 1; f <= c.odom_f_max; ++f) {
            if (u - f < 0) continue;
            if (G->world[u - f] != G->world[u]) continue;
            SE3 u_M_umf = se3_mul(se3_inv(G->vio[u]), G->vio[u - f]);     // :1599
            double w = std::pow(0.9, f);                                  // :1604
            if (c.apply_yaw_weight) { const double y = yaw_deg(u_M_umf.R); w *= std::exp(-y * y / 6.0); }  // :1605-1606
            G->o_c1.push_back((int32_t)u); G->o_c2.push_back((int32_t)(u - f)); G->o_T.push_back(u_M_umf); G->o_w.push_back(w);
        }
    }

    // ---- loop edges: sample candidates (partial Fisher-Yates); force a fraction to be inter-world when W > 1
    {
        std::vector<std::pair<int32_t, int32_t>> inter, intra;
        for (auto& pr : cand) (G->world[pr.first] != G->world[pr.second] ? inter : intra).push_back(pr);
        int64_t want = std::min<int64_t>(c.n_loops, (int64_t)cand.size());
        int64_t want_inter = W > 1 ? std::min<int64_t>((int64_t)std::llround(c.inter_world_frac * want), (int64_t)inter.size()) : 0;
        int64_t want_intra = std::min<int64_t>(want - want_inter, (int64_t)intra.size());
        auto sample = [&](std::vector<std::pair<int32_t, int32_t>>& v, int64_t k) {
            for (int64_t i = 0; i < k; ++i) { const uint64_t j = i + rng.below((uint64_t)(v.size() - i)); std::swap(v[i], v[j]); }
            v.resize(k);
        };
        sample(inter, want_inter);
        sample(intra, want_intra);
        std::vector<std::pair<int32_t, int32_t>> chosen(intra);
        chosen.insert(chosen.end(), inter.begin(), inter.end());
        // loop-closure messages arrive in time order of the newer keyframe
        std::sort(chosen.begin(), chosen.end());
        for (auto& pr : chosen) {
            const int32_t a = pr.first, b = pr.second;
            SE3 bTa = se3_mul(se3_inv(G->truth[b]), G->truth[a]);
            int outl = rng.uniform() < c.outlier_frac ? 1 : 0;
            if (outl) {
                V3 ax{rng.normal(), rng.normal(), rng.normal()};
                const double n = std::sqrt(ax.x * ax.x + ax.y * ax.y + ax.z * ax.z) + 1e-300, ang = rng.uniform() * M_PI;
                V3 dir{rng.normal(), rng.normal(), rng.normal()};
                const double dn = std::sqrt(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z) + 1e-300, len = rng.uniform() * 5.0;
                bTa = SE3{so3_exp(V3{ax.x / n * ang, ax.y / n * ang, ax.z / n * ang}), V3{dir.x / dn * len, dir.y / dn * len, dir.z / dn * len}};
            } else {
                SE3 noise{so3_exp(V3{rng.normal() * c.loop_sigma_r, rng.normal() * c.loop_sigma_r, rng.normal() * c.loop_sigma_r}),
                          V3{rng.normal() * c.loop_sigma_t, rng.normal() * c.loop_sigma_t, rng.normal() * c.loop_sigma_t}};
                bTa = se3_mul(bTa, noise);
            }
            // c1 = b (second, older), c2 = a (first, newer): src/PoseGraphSLAM.cpp:1552-1555
            G->l_c1.push_back(b); G->l_c2.push_back(a); G->l_T.push_back(bTa); G->l_w.push_back(1.0); G->l_out.push_back(outl);
        }
    }

    // ---- world merging + initial guess as the trigger does it (src/PoseGraphSLAM.cpp:1459-1464, 1770-1783)
    std::vector<int> root(W); std::vector<SE3> root_T_w(W, se3_identity()); std::vector<char> known(W, 0);
    for (int w = 0; w < W; ++w) root[w] = w;
    known[0] = 1;
    // iterate loop edges in arrival order; a world joins the set of world 0 through the first edge that reaches it
    // from an already-joined world.  (Unjoined worlds keep their own frame and their own regulariser.)
    bool progress = true;
    while (progress) {
        progress = false;
        for (size_t e = 0; e < G->l_c1.size(); ++e) {
            const int b = G->l_c1[e], a = G->l_c2[e];
            const int wb = G->world[b], wa = G->world[a];
            if (wa == wb || known[wa] == known[wb]) continue;
            if (G->l_out[e]) continue;   // keep the synthetic merge well-posed: the first contact edge is an inlier
            // wb_T_wa = wb_T_b * b_T_a * (wa_T_a)^-1 from ODOMETRY poses (:1459-1464)
            SE3 wb_T_wa = se3_mul(se3_mul(G->vio[b], G->l_T[e]), se3_inv(G->vio[a]));
            if (known[wb]) { root_T_w[wa] = se3_mul(root_T_w[wb], wb_T_wa); known[wa] = 1; root[wa] = 0; }
            else { root_T_w[wb] = se3_mul(root_T_w[wa], se3_inv(wb_T_wa)); known[wb] = 1; root[wb] = 0; }
            progress = true;
        }
    }
    G->init.resize(N);
    for (int64_t k = 0; k < N; ++k) G->init[k] = se3_mul(root_T_w[G->world[k]], G->vio[k]);   // wset_T_w * w_M_u (:1780)

    // ---- regularisers (src/PoseGraphSLAM.cpp:1817-1849): first node of each world that is its own set root
    for (int w = 0; w < W; ++w) {
        if (root[w] != w) continue;
        const int64_t start = (int64_t)w * per_world, end = std::min<int64_t>(N, start + per_world) - 1;
        if (start >= N) continue;
        G->r_node.push_back((int32_t)start);
        G->r_T.push_back(G->init[start]);
        G->r_w.push_back(std::max(1.1, std::log(1.0 + (double)(end - start)) / 2.0));      // :1839
    }
    *out = G;
    return 0;
}

void pgo_gen_destroy(pgo_gen_graph* g) { delete g; }
int64_t pgo_gen_num_poses(const pgo_gen_graph* g) { return (int64_t)g->truth.size(); }
int64_t pgo_gen_num_odom(const pgo_gen_graph* g) { return (int64_t)g->o_c1.size(); }
int64_t pgo_gen_num_loops(const pgo_gen_graph* g) { return (int64_t)g->l_c1.size(); }
int64_t pgo_gen_num_regularizers(const pgo_gen_graph* g) { return (int64_t)g->r_node.size(); }

int pgo_gen_get_poses(const pgo_gen_graph* g, double* tq, double* tt, double* iq, double* it, int32_t* world) {
    const size_t N = g->truth.size();
    for (size_t k = 0; k < N; ++k) {
        if (tq) m3_to_quat(g->truth[k].R, tq + 4 * k);
        if (tt) { tt[3 * k] = g->truth[k].t.x; tt[3 * k + 1] = g->truth[k].t.y; tt[3 * k + 2] = g->truth[k].t.z; }
        if (iq) m3_to_quat(g->init[k].R, iq + 4 * k);
        if (it) { it[3 * k] = g->init[k].t.x; it[3 * k + 1] = g->init[k].t.y; it[3 * k + 2] = g->init[k].t.z; }
        if (world) world[k] = g->world[k];
    }
    return 0;
}
int pgo_gen_get_odom(const pgo_gen_graph* g, int32_t* c1, int32_t* c2, double* T, double* w) {
    for (size_t k = 0; k < g->o_c1.size(); ++k) {
        if (c1) c1[k] = g->o_c1[k];
        if (c2) c2[k] = g->o_c2[k];
        if (T) se3_to_colmajor(g->o_T[k], T + 16 * k);
        if (w) w[k] = g->o_w[k];
    }
    return 0;
}
int pgo_gen_get_loops(const pgo_gen_graph* g, int32_t* c1, int32_t* c2, double* T, double* w, int32_t* outl) {
    for (size_t k = 0; k < g->l_c1.size(); ++k) {
        if (c1) c1[k] = g->l_c1[k];
        if (c2) c2[k] = g->l_c2[k];
        if (T) se3_to_colmajor(g->l_T[k], T + 16 * k);
        if (w) w[k] = g->l_w[k];
        if (outl) outl[k] = g->l_out[k];
    }
    return 0;
}
int pgo_gen_get_regularizers(const pgo_gen_graph* g, int32_t* node, double* T, double* w) {
    for (size_t k = 0; k < g->r_node.size(); ++k) {
        if (node) node[k] = g->r_node[k];
        if (T) se3_to_colmajor(g->r_T[k], T + 16 * k);
        if (w) w[k] = g->r_w[k];
    }
    return 0;
}

}  // extern "C"
