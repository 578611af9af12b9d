/*
 * pgo_graphgen.h — deterministic synthetic 3D "Manhattan-world" keyframe pose graphs (BASELINE.json configs).
 *
 * This is the workload generator for bench.py and the parity tests, not part of the solver.  It plays the
 * role of the reference's CALLER: it produces exactly what `NodeDataManager` hands to the solver trigger —
 * a VIO pose array per world, loop-closure messages (a, b, b_T_a, weight) — and then derives the solver's
 * inputs the way reference src/PoseGraphSLAM.cpp does:
 *   odometry edges   (u, u-f), f = 1..odom_f_max, measurement u_M_umf = w_M_u^-1 * w_M_umf       (:1597-1599)
 *                    weight 0.9^f [* exp(-yaw_deg^2/6) when apply_yaw_weight]                      (:1603-1606)
 *   loop edges       c1 = b (older), c2 = a (newer), measurement b_T_a, one switch per edge        (:1550-1556)
 *   initial guess    VIO poses mapped into the merged set's frame, w0_T_wk from the first
 *                    inter-world loop edge using odometry poses                                    (:1459-1464,:1770-1783)
 *   regularisers     first node of every world that is its own set root, weight max(1.1, ln(1+end-start)/2),
 *                    target = initial pose of that node                                            (:1817-1849)
 * All randomness: std::mt19937_64 + an explicit Box-Muller (no implementation-defined distributions).
 */
#ifndef PGO_GRAPHGEN_H_
#define PGO_GRAPHGEN_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgo_gen_config {
    int64_t n_poses;          /* total keyframes over all worlds */
    int64_t n_loops;          /* requested loop closures (sampled uniformly from the spatial candidates) */
    int32_t odom_f_max;       /* odometry edges (u,u-f) for f = 1..odom_f_max; reference uses 5 (PoseGraphSLAM.cpp:1577) */
    int32_t apply_yaw_weight; /* multiply odom weight by exp(-yaw_deg^2/6) as the reference does */
    int32_t n_worlds;         /* >= 1; worlds are equal-length consecutive segments separated by a kidnap */
    int32_t reserved_;
    double inter_world_frac;  /* fraction of loop edges forced to connect different worlds (n_worlds > 1) */
    double outlier_frac;      /* fraction of loop edges whose measurement is a random SE(3) */
    double odom_sigma_t;      /* VIO drift per keyframe: translation N(0, sigma^2) per axis [m] */
    double odom_sigma_r;      /*                         rotation exp(N(0, sigma^2 I)) [rad] */
    double loop_sigma_t;      /* loop-closure measurement noise */
    double loop_sigma_r;
    double box_scale;         /* trajectory confined to a cube of side box_scale * n_poses^(1/3) metres */
    double turn_deg_per_keyframe; /* a 90-degree turn is executed at this rate (2 deg: SURVEY.md §8d) */
    double loop_radius;       /* loop candidates: |p_a - p_b| < loop_radius (1.5 m) */
    int32_t straight_min, straight_max; /* straight run length ~ U{min..max} keyframes (5..25) */
    int32_t min_loop_gap;     /* loop candidates need a - b > min_loop_gap (50) */
    int32_t reserved2_;
    uint64_t seed;
} pgo_gen_config;

typedef struct pgo_gen_graph pgo_gen_graph;

void pgo_gen_config_init(pgo_gen_config* c);   /* defaults: f_max 1, no yaw weight, 1 world, 10% outliers, seed 1 */
int pgo_gen_create(const pgo_gen_config* c, pgo_gen_graph** out);
void pgo_gen_destroy(pgo_gen_graph* g);

int64_t pgo_gen_num_poses(const pgo_gen_graph* g);
int64_t pgo_gen_num_odom(const pgo_gen_graph* g);
int64_t pgo_gen_num_loops(const pgo_gen_graph* g);
int64_t pgo_gen_num_regularizers(const pgo_gen_graph* g);

/* Any pointer may be NULL.  quat x,y,z,w; T column-major 4x4. */
int pgo_gen_get_poses(const pgo_gen_graph* g, double* truth_quat, double* truth_t, double* init_quat, double* init_t, int32_t* world_of_node);
int pgo_gen_get_odom(const pgo_gen_graph* g, int32_t* c1, int32_t* c2, double* c1_T_c2, double* weight);
int pgo_gen_get_loops(const pgo_gen_graph* g, int32_t* c1, int32_t* c2, double* c1_T_c2, double* weight, int32_t* is_outlier);
int pgo_gen_get_regularizers(const pgo_gen_graph* g, int32_t* node, double* target, double* weight);

#ifdef __cplusplus
}
#endif
#endif
