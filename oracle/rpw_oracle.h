/*
 * rpw_oracle.h -- C interface of the CPU ORACLE for the Recursive Patchwork hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker / reported CPU baseline. The product library
 * (librpw_hip.so) neither links nor calls anything declared here.
 *
 * PARITY STATUS: "parity unpinned". The reference holds no golden vectors for
 * this path (its tests assert sizes on random_device inputs only) and cannot be
 * compiled in this image (its header needs <Eigen/Dense>, which is absent), so
 * this restatement is pinned only by independent sanity checks (tests/), not by
 * outputs of the reference itself. See oracle/README.md.
 */
#ifndef RPW_ORACLE_H
#define RPW_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors recursive_patchwork::PatchworkConfig field for field
 * (reference include/recursive_patchwork.hpp:25-36). */
typedef struct rpwo_config {
    float   sensor_height;        /* 1.2  */
    float   max_range;            /* 150 (never read by the path) */
    int32_t num_sectors;          /* 10   */
    int32_t max_iter;             /* 100  */
    int32_t adaptive_seed_height; /* 1    */
    float   th_seeds;             /* 0.15 */
    float   th_dist;              /* 0.2  */
    float   th_outlier;           /* 0.08 (never read by the path) */
    float   filtering_radius;     /* 150  */
    int32_t max_split_depth;      /* 1000 */
} rpwo_config;

/* label values, one per INPUT point */
enum {
    RPWO_NONGROUND = 0,   /* in the zone (d <= R), not ground                      */
    RPWO_GROUND    = 1,   /* in the zone, ground                                   */
    RPWO_DROPPED   = 2,   /* non-finite coordinate, removed by cleanPoints         */
    RPWO_OUTSIDE   = 3    /* finite, d > R: appended to non-ground after the zone  */
};

/* node kinds in the per-patch trace */
enum {
    RPWO_NODE_FIT      = 0, /* leaf: iterative fit ran, mask = last loop mask      */
    RPWO_NODE_REJECT   = 1, /* leaf: n < 3 or depth > max_split_depth -> all false */
    RPWO_NODE_SMALL    = 2, /* leaf: bbox area < 25 && depth > 0 -> all true       */
    RPWO_NODE_FLAT     = 3, /* leaf: z-range < 0.05 && n > 10 -> all true          */
    RPWO_NODE_SPLIT    = 4  /* internal node: fit ran, then split                  */
};

typedef struct rpwo_node {
    int32_t  patch;       /* ring * num_sectors + sector                     */
    int32_t  depth;
    uint32_t start;       /* first slot, relative to the patch's first slot  */
    uint32_t len;
    int32_t  kind;
    int32_t  iterations;  /* plane refits executed inside the loop           */
    uint32_t n_inliers;   /* trues in the node's final loop mask             */
    int32_t  split_axis;  /* SPLIT only: 0 = x, 1 = y                        */
    float    split_value; /* SPLIT only: the median                          */
    float    centroid[3]; /* final plane (RP.cpp:228)                        */
    float    normal[3];
    float    residual;
} rpwo_node;

typedef struct rpwo_stats {
    uint32_t n_in, n_clean, n_zone, n_ground;
    uint32_t n_patches;        /* non-empty patches                  */
    uint32_t n_nodes;          /* nodes visited (all kinds)          */
    uint32_t n_splits;
    uint32_t max_depth;
    uint64_t total_iterations;
} rpwo_stats;

/* Whole path: filterGroundPoints (RP.cpp:310-426) restated.
 *   xyz       n x 3 floats, AoS (Point3D layout, RP.hpp:18-22)
 *   labels    n bytes out (enum above)
 *   residual  n floats out or NULL: |(p_i - c).n| against the final plane of the
 *             FIT leaf whose slot mask gave point i its label; -1 otherwise
 *   nodes     trace buffer or NULL; at most nodes_cap records are written, the
 *             stats say how many nodes existed
 * returns 0 on success. */
int rpwo_segment(const rpwo_config* cfg, const float* xyz, uint32_t n,
                 uint8_t* labels, float* residual,
                 rpwo_node* nodes, uint32_t nodes_cap, rpwo_stats* stats);

/* The reference's return value, in the reference's order (RP.cpp:402-425):
 * ground = zone points flagged, zone order; nonground = zone points not flagged,
 * zone order, then cleaned points with d > R in cleaned order.
 * ground_xyz / nonground_xyz must each hold n points. */
int rpwo_filter_ground_points(const rpwo_config* cfg, const float* xyz, uint32_t n,
                              float* ground_xyz, uint32_t* n_ground,
                              float* nonground_xyz, uint32_t* n_nonground);

/* LidarFusion pre-step (reference src/lidar_fusion.cpp:42-126,148-159,184-187): per-sensor yaw rotation,
 * ego-radius removal, concatenation. out_xyz / out_src must hold sensor_off[n_sensors] entries. */
int rpwo_fuse(const float* xyz, const uint32_t* sensor_off, uint32_t n_sensors, const float* rotation_deg,
              const float* ego_radius, float* out_xyz, uint32_t* out_src, uint32_t* n_out);

/* sampleGroundAndObstacles (reference src/recursive_patchwork.cpp:428-465, randomSubsample src/point_cloud_processor.cpp:121-146)
 * with the std::mt19937 seed as an argument (the reference seeds from std::random_device). out_xyz must hold n points. */
int rpwo_sample_ground_and_obstacles(const rpwo_config* cfg, const float* xyz, uint32_t n, float target_height,
                                     float base_tol, uint32_t seed, float* out_xyz, uint32_t* n_out);

/* "as shipped" mode: print, flushed, the lines a CUDA-less build of the reference prints on the path
 * (cuda_interface.cu:57,586-651; recursive_patchwork.cpp:30,421-423). Process-wide; off by default. */
void rpwo_set_chatter(int on);

/* CPU baseline harness (bench.py cpu_baseline): native threads, one frame per thread at a time. frames are
 * concatenated in xyz, frame f = points [frame_off[f], frame_off[f+1]). n_threads = 0: hardware_concurrency. */
typedef struct rpwo_bench_result {
    double   single_thread_median_ms;   /* median over single_thread_frames frames, after 3 warm-ups */
    uint32_t single_thread_frames;
    uint32_t threads;                   /* threads of the parallel phase                              */
    uint32_t hardware_concurrency;      /* std::thread::hardware_concurrency()                        */
    uint64_t frames_done;               /* parallel phase                                             */
    double   elapsed_s;
} rpwo_bench_result;
int rpwo_bench_threads(const rpwo_config* cfg, const float* xyz, const uint32_t* frame_off, uint32_t n_frames,
                       uint32_t n_threads, double seconds, uint32_t single_reps, rpwo_bench_result* out);

/* analysis aid: record {patch, depth, iter, inlier count, mask hash} (5 x uint64) for every loop
 * iteration of the following rpwo_segment calls; buf = NULL switches it off */
void     rpwo_set_iter_trace(uint64_t* buf, uint32_t cap);
uint32_t rpwo_iter_trace_count(void);

/* pieces exposed for unit tests */
/* Eigen 3.4.0 SelfAdjointEigenSolver<Matrix3f>(cov) restated; cov row-major 9 floats
 * (only the lower triangle is read); evals[3], evecs column-major 9 floats
 * (evecs[3*c + r] = row r of eigenvector c); returns 0 = Success, 1 = NoConvergence */
int  rpwo_eig3(const float* cov, float* evals, float* evecs);
/* fitPlanePCA (RP.cpp:77-107): out[7] = centroid, normal, residual */
void rpwo_fit_plane(const float* xyz, uint32_t n, float* out7);
/* ring edges / sector bounds tables (RP.cpp:344-352,364): edges[9], bounds[S+1] */
void rpwo_tables(const rpwo_config* cfg, float* edges9, float* bounds);
/* computeAngles (cuda_interface.cu:617-632) for one point */
float rpwo_angle(float x, float y);

#ifdef __cplusplus
}
#endif
#endif
