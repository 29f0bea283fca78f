// rpw_oracle.cpp -- CPU ORACLE: a restatement of the reference's Recursive Patchwork
// hot path, used ONLY as the parity checker and as the reported CPU baseline.
//
// TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg load this; the product (librpw_hip.so) never does.
//
// PARITY STATUS: parity unpinned (see rpw_oracle.h / oracle/README.md). The
// reference cannot be built here (needs <Eigen/Dense>) and has no golden vectors.
//
// What is restated (paths relative to /root/reference/src/recursive_patchwork):
//   src/recursive_patchwork.cpp:19-35    cleanPoints
//   src/recursive_patchwork.cpp:77-107   fitPlanePCA
//   src/recursive_patchwork.cpp:109-308  fitPlaneAndSplit
//   src/recursive_patchwork.cpp:310-426  filterGroundPoints
//   src/point_cloud_processor.cpp:58-86  computeCentroid / computeCovariance
//   cuda/cuda_interface.cu:582-654       CPU branches of the four cuda::ops used
//   Eigen 3.4.0 (third party, NOT in /root/reference; find_package(Eigen3) in
//   CMakeLists.txt:15, unpinned, 3.4.0 = Ubuntu 22.04 / ROS2 Humble):
//     SelfAdjointEigenSolver<Matrix3f>::compute, tridiagonalization_inplace 3x3 real,
//     computeFromTridiagonal_impl, tridiagonal_qr_step, JacobiRotation::makeGivens,
//     numext::hypot, the size-3 unrolled redux used by Vector3f::dot.
//
// Semantics: strict IEEE binary32, one rounding per operation, no FMA contraction,
// no fast-math (the flavour of the reference's own test target, CMakeLists.txt:238-243).
// Build with: g++ -O2 -ffp-contract=off -fno-fast-math (see Makefile).

#include "rpw_oracle.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstring>
#include <iostream>
#include <malloc.h>
#include <limits>
#include <numeric>
#include <random>
#include <thread>
#include <vector>

namespace {

// optional per-iteration trace (analysis aid): {patch, depth, iter, count, hash} as 5 x uint64
uint64_t* g_iter_trace = nullptr; uint32_t g_iter_cap = 0, g_iter_n = 0;

// "As shipped" mode (rpwo_set_chatter): print what a CUDA-less build of the reference prints on its way through the
// path, line for line and flushed with std::endl like there -- cuda_interface.cu:57 (isAvailable, once per cuda::ops
// call), :586,592 (computeDistances2D), :603,611-612 (filterPointsByRadius), :621,629 (computeAngles), :637,642,643,651
// (computePlaneDistances: once per refit iteration per patch), recursive_patchwork.cpp:30-31,421-423. Off by default:
// the checker and the headline CPU baseline run silent.
int g_chatter = 0;
inline void say_unavailable() { if (g_chatter) std::cout << "USE_CUDA not defined, CUDA not available" << std::endl; }

struct P3 { float x, y, z; };

struct Plane {
    float c[3];
    float n[3];
    float residual;
};

// ---------------------------------------------------------------------------------
// Eigen 3.4.0 pieces
// ---------------------------------------------------------------------------------

// numext::hypot -> positive_real_hypot(|x|, |y|)  (Eigen/src/Core/MathFunctionsImpl.h)
inline float eigen_hypot(float a, float b)
{
    float x = std::fabs(a), y = std::fabs(b);
    if (std::isinf(x) || std::isinf(y)) return std::numeric_limits<float>::infinity();
    if (std::isnan(x) || std::isnan(y)) return std::numeric_limits<float>::quiet_NaN();
    float p = x > y ? x : y;            // numext::maxi(x, y)
    if (p == 0.0f) return 0.0f;
    float q = (y < x ? y : x) / p;      // numext::mini(y, x) / p
    return p * std::sqrt(1.0f + q * q);
}

// JacobiRotation<float>::makeGivens(p, q), real case (Eigen/src/Jacobi/Jacobi.h)
inline void make_givens(float p, float q, float& c, float& s)
{
    if (q == 0.0f) {
        c = p < 0.0f ? -1.0f : 1.0f;
        s = 0.0f;
    } else if (p == 0.0f) {
        c = 0.0f;
        s = q < 0.0f ? 1.0f : -1.0f;
    } else if (std::fabs(p) > std::fabs(q)) {
        float t = q / p;
        float u = std::sqrt(1.0f + t * t);
        if (p < 0.0f) u = -u;
        c = 1.0f / u;
        s = -t * c;
    } else {
        float t = p / q;
        float u = std::sqrt(1.0f + t * t);
        if (q < 0.0f) u = -u;
        s = -1.0f / u;
        c = -t * s;
    }
}

// internal::tridiagonal_qr_step<ColMajor>(diag, subdiag, start, end, Q, n=3)
// Q column-major: Q[3*col + row].
inline void qr_step(float* diag, float* sub, int start, int end, float* Q)
{
    float td = (diag[end - 1] - diag[end]) * 0.5f;
    float e  = sub[end - 1];
    float mu = diag[end];
    if (td == 0.0f) {
        mu -= std::fabs(e);
    } else if (e != 0.0f) {
        const float e2 = e * e;
        const float h  = eigen_hypot(td, e);
        if (e2 == 0.0f) mu -= e / ((td + (td > 0.0f ? h : -h)) / e);
        else            mu -= e2 / (td + (td > 0.0f ? h : -h));
    }
    float x = diag[start] - mu;
    float z = sub[start];
    for (int k = start; k < end && z != 0.0f; ++k) {
        float c, s;
        make_givens(x, z, c, s);
        // T = G' T G
        float sdk  = s * diag[k] + c * sub[k];
        float dkp1 = s * sub[k] + c * diag[k + 1];
        diag[k]     = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
        diag[k + 1] = s * sdk + c * dkp1;
        sub[k]      = c * sdk - s * dkp1;
        if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
        x = sub[k];
        if (k < end - 1) {
            z = -s * sub[k + 1];
            sub[k + 1] = c * sub[k + 1];
        }
        // Q = Q * G : applyOnTheRight(k, k+1, rot) == apply_rotation_in_the_plane with
        // rot.transpose() = (c, -s); skipped when c == 1 && s == 0.
        if (!(c == 1.0f && s == 0.0f)) {
            for (int i = 0; i < 3; ++i) {
                float xi = Q[3 * k + i], yi = Q[3 * (k + 1) + i];
                Q[3 * k + i]       = c * xi - s * yi;
                Q[3 * (k + 1) + i] = s * xi + c * yi;
            }
        }
    }
}

// SelfAdjointEigenSolver<Matrix3f>::compute(A, ComputeEigenvectors); A row-major,
// lower triangle only is read. Returns 0 Success / 1 NoConvergence.
int eig3(const float* A, float* evals, float* Q)
{
    // mat = lower triangular view (upper zero); scale to [-1, 1]
    float m00 = A[0], m10 = A[3], m11 = A[4], m20 = A[6], m21 = A[7], m22 = A[8];
    float scale = 0.0f;                       // cwiseAbs().maxCoeff() over all 9 (upper = 0)
    {
        const float v[6] = {m00, m10, m20, m11, m21, m22};
        for (float t : v) { float a = std::fabs(t); if (a > scale) scale = a; }
    }
    if (scale == 0.0f) scale = 1.0f;
    m00 /= scale; m10 /= scale; m11 /= scale; m20 /= scale; m21 /= scale; m22 /= scale;

    float diag[3], sub[2];
    // tridiagonalization_inplace_selector<Matrix3f, 3, false>::run
    {
        const float tol = FLT_MIN;
        diag[0] = m00;
        float v1norm2 = m20 * m20;
        if (v1norm2 <= tol) {
            diag[1] = m11; diag[2] = m22; sub[0] = m10; sub[1] = m21;
            Q[0] = 1; Q[1] = 0; Q[2] = 0; Q[3] = 0; Q[4] = 1; Q[5] = 0; Q[6] = 0; Q[7] = 0; Q[8] = 1;
        } else {
            float beta    = std::sqrt(m10 * m10 + v1norm2);
            float invBeta = 1.0f / beta;
            float m01 = m10 * invBeta;
            float m02 = m20 * invBeta;
            float q   = 2.0f * m01 * m21 + m02 * (m22 - m11);
            diag[1] = m11 + m02 * q;
            diag[2] = m22 - m02 * q;
            sub[0]  = beta;
            sub[1]  = m21 - m01 * q;
            // mat << 1,0,0, 0,m01,m02, 0,m02,-m01 (row-major listing) -> column-major store
            Q[0] = 1; Q[1] = 0;   Q[2] = 0;
            Q[3] = 0; Q[4] = m01; Q[5] = m02;
            Q[6] = 0; Q[7] = m02; Q[8] = -m01;
        }
    }
    // computeFromTridiagonal_impl, n = 3, m_maxIterations = 30
    const int n = 3, maxIterations = 30;
    int end = n - 1, start = 0, iter = 0;
    const float considerAsZero = FLT_MIN;
    const float precision_inv  = 1.0f / FLT_EPSILON;
    while (end > 0) {
        for (int i = start; i < end; ++i) {
            if (std::fabs(sub[i]) < considerAsZero) {
                sub[i] = 0.0f;
            } else {
                const float scaled = precision_inv * sub[i];
                if (scaled * scaled <= (std::fabs(diag[i]) + std::fabs(diag[i + 1]))) sub[i] = 0.0f;
            }
        }
        while (end > 0 && sub[end - 1] == 0.0f) end--;
        if (end <= 0) break;
        iter++;
        if (iter > maxIterations * n) break;
        start = end - 1;
        while (start > 0 && sub[start - 1] != 0.0f) start--;
        qr_step(diag, sub, start, end, Q);
    }
    int info = (iter <= maxIterations * n) ? 0 : 1;
    if (info == 0) {
        for (int i = 0; i < n - 1; ++i) {
            int k = 0;                               // segment(i, n-i).minCoeff(&k): first minimum
            for (int j = 1; j < n - i; ++j) if (diag[i + j] < diag[i + k]) k = j;
            if (k > 0) {
                std::swap(diag[i], diag[k + i]);
                for (int r = 0; r < 3; ++r) std::swap(Q[3 * i + r], Q[3 * (k + i) + r]);
            }
        }
    }
    for (int i = 0; i < 3; ++i) evals[i] = diag[i] * scale;
    return info;
}

// ---------------------------------------------------------------------------------
// point_cloud_processor.cpp:58-86 and recursive_patchwork.cpp:77-107
// ---------------------------------------------------------------------------------

inline void centroid_of(const std::vector<P3>& pts, float c[3])
{
    c[0] = c[1] = c[2] = 0.0f;
    if (pts.empty()) return;
    for (const P3& p : pts) { c[0] += p.x; c[1] += p.y; c[2] += p.z; }
    const float n = (float)pts.size();
    c[0] /= n; c[1] /= n; c[2] /= n;
}

inline void covariance_of(const std::vector<P3>& pts, const float c[3], float cov[9])
{
    for (int i = 0; i < 9; ++i) cov[i] = 0.0f;
    if (pts.size() < 2) return;
    for (const P3& p : pts) {
        const float d[3] = {p.x - c[0], p.y - c[1], p.z - c[2]};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                cov[3 * i + j] += d[i] * d[j];
    }
    const float m = (float)(pts.size() - 1);
    for (int i = 0; i < 9; ++i) cov[i] /= m;
}

// |(p - c).dot(n)| with Eigen's size-3 unrolled redux: e0 + (e1 + e2)
inline float plane_distance(const P3& p, const float c[3], const float n[3])
{
    const float e0 = (p.x - c[0]) * n[0];
    const float e1 = (p.y - c[1]) * n[1];
    const float e2 = (p.z - c[2]) * n[2];
    return std::fabs(e0 + (e1 + e2));
}

Plane fit_plane_pca(const std::vector<P3>& pts)
{
    Plane r;
    if (pts.size() < 3) {
        r.c[0] = r.c[1] = r.c[2] = 0.0f;
        r.n[0] = 0.0f; r.n[1] = 0.0f; r.n[2] = 1.0f;
        r.residual = FLT_MAX;
        return r;
    }
    float cov[9], evals[3], Q[9];
    centroid_of(pts, r.c);
    covariance_of(pts, r.c, cov);
    eig3(cov, evals, Q);
    r.n[0] = Q[0]; r.n[1] = Q[1]; r.n[2] = Q[2];      // eigenvectors().col(0)
    if (r.n[2] < 0.0f) { r.n[0] = -r.n[0]; r.n[1] = -r.n[1]; r.n[2] = -r.n[2]; }
    float residual = 0.0f;
    for (const P3& p : pts) residual += plane_distance(p, r.c, r.n);
    residual /= (float)pts.size();
    r.residual = residual;
    return r;
}

// ---------------------------------------------------------------------------------
// recursive_patchwork.cpp:109-308
// ---------------------------------------------------------------------------------

struct Ctx {
    const rpwo_config* cfg;
    rpwo_node* nodes;
    uint32_t nodes_cap;
    rpwo_stats* st;
    int patch;
    // per-slot outputs for the patch being processed (positional, like the reference's
    // returned vector<bool>): plane index of the FIT leaf that produced the slot, or -1
    std::vector<int>* slot_leaf;
    std::vector<Plane>* leaf_planes;
};

void emit(Ctx& cx, int depth, uint32_t start, uint32_t len, int kind, int iters,
          uint32_t n_inl, int axis, float split_value, const Plane* pl)
{
    rpwo_stats* st = cx.st;
    if (cx.nodes && st->n_nodes < cx.nodes_cap) {
        rpwo_node& nd = cx.nodes[st->n_nodes];
        nd.patch = cx.patch; nd.depth = depth; nd.start = start; nd.len = len; nd.kind = kind;
        nd.iterations = iters; nd.n_inliers = n_inl; nd.split_axis = axis; nd.split_value = split_value;
        for (int i = 0; i < 3; ++i) { nd.centroid[i] = pl ? pl->c[i] : 0.0f; nd.normal[i] = pl ? pl->n[i] : 0.0f; }
        nd.residual = pl ? pl->residual : 0.0f;
    }
    st->n_nodes++;
    if ((uint32_t)depth > st->max_depth) st->max_depth = (uint32_t)depth;
    st->total_iterations += (uint64_t)(iters > 0 ? iters : 0);
}

// Returns the positional mask; `start` is the slot offset of this call inside the patch
// (bookkeeping for the trace only -- the arithmetic never sees it).
std::vector<bool> fit_plane_and_split(Ctx& cx, const std::vector<P3>& pts, float mean_dist,
                                      int depth, uint32_t start)
{
    const rpwo_config& cfg = *cx.cfg;
    const size_t n = pts.size();
    if (n < 3 || depth > cfg.max_split_depth) {
        emit(cx, depth, start, (uint32_t)n, RPWO_NODE_REJECT, 0, 0, -1, 0.0f, nullptr);
        return std::vector<bool>(n, false);
    }
    float x_min = FLT_MAX, x_max = -FLT_MAX, y_min = FLT_MAX, y_max = -FLT_MAX;
    for (const P3& p : pts) {
        x_min = std::min(x_min, p.x); x_max = std::max(x_max, p.x);
        y_min = std::min(y_min, p.y); y_max = std::max(y_max, p.y);
    }
    const float area = (x_max - x_min) * (y_max - y_min);
    if (area < 25.0f && depth > 0) {
        emit(cx, depth, start, (uint32_t)n, RPWO_NODE_SMALL, 0, (uint32_t)n, -1, 0.0f, nullptr);
        return std::vector<bool>(n, true);
    }
    float z_min = FLT_MAX, z_max = -FLT_MAX;
    for (const P3& p : pts) { z_min = std::min(z_min, p.z); z_max = std::max(z_max, p.z); }
    if ((z_max - z_min) < 0.05f && n > 10) {
        emit(cx, depth, start, (uint32_t)n, RPWO_NODE_FLAT, 0, (uint32_t)n, -1, 0.0f, nullptr);
        return std::vector<bool>(n, true);
    }

    const float rel_dist = mean_dist / cfg.filtering_radius;
    float z_th;
    if (cfg.adaptive_seed_height) {
        z_th = cfg.sensor_height + 0.2f * rel_dist;
    } else {
        std::vector<float> sorted_z(n);
        for (size_t i = 0; i < n; ++i) sorted_z[i] = pts[i].z;
        std::sort(sorted_z.begin(), sorted_z.end());
        const size_t idx = static_cast<size_t>(0.1f * sorted_z.size());
        z_th = sorted_z[idx] + cfg.th_seeds;
    }

    std::vector<bool> ground(n, false);
    size_t seed_count = 0;
    for (size_t i = 0; i < n; ++i) if (pts[i].z < z_th) { ground[i] = true; ++seed_count; }
    if (seed_count < 3) {
        // three lowest points by z; tie order is that of libstdc++'s partial_sort, as in the reference
        std::vector<size_t> order(n);
        std::iota(order.begin(), order.end(), (size_t)0);
        std::partial_sort(order.begin(), order.begin() + 3, order.end(),
                          [&pts](size_t a, size_t b) { return pts[a].z < pts[b].z; });
        ground.assign(n, false);
        for (int i = 0; i < 3; ++i) ground[order[i]] = true;
    }

    const float threshold = cfg.th_dist * (1.0f + 0.2f * rel_dist);
    int iterations = 0;
    std::vector<P3> inl;
    inl.reserve(n);
    for (int iter = 0; iter < cfg.max_iter; ++iter) {
        inl.clear();
        for (size_t i = 0; i < n; ++i) if (ground[i]) inl.push_back(pts[i]);
        if (inl.size() < 3) break;
        const Plane pl = fit_plane_pca(inl);
        ++iterations;
        std::vector<bool> next(n, false);
        if (g_chatter) {                        // cuda::ops::computePlaneDistances, CPU branch (cuda_interface.cu:634-654)
            std::cout << "computePlaneDistances called with " << n << " points" << std::endl;
            say_unavailable();
            std::cout << "Using CPU for plane distances" << std::endl;
            std::cout << "[CPU] Computing plane distances for " << n << " points..." << std::endl;
        }
        for (size_t i = 0; i < n; ++i)
            if (plane_distance(pts[i], pl.c, pl.n) < threshold) next[i] = true;
        if (g_chatter) std::cout << "[CPU] Plane distance computation completed successfully" << std::endl;
        if (g_iter_trace && g_iter_n < g_iter_cap) {
            uint64_t h = 1469598103934665603ull, cnt = 0;
            for (size_t i = 0; i < n; ++i) if (next[i]) { h = (h ^ (uint64_t)i) * 1099511628211ull; ++cnt; }
            uint64_t* r = g_iter_trace + 5 * (size_t)g_iter_n++;
            r[0] = (uint64_t)cx.patch; r[1] = (uint64_t)depth; r[2] = (uint64_t)iter; r[3] = cnt; r[4] = h;
        }
        if (next == ground) break;
        ground.swap(next);
    }

    inl.clear();
    for (size_t i = 0; i < n; ++i) if (ground[i]) inl.push_back(pts[i]);
    const Plane fin = fit_plane_pca(inl);

    const float split_threshold = cfg.th_dist * (1.0f + 1.5f * depth);
    const size_t min_patch_size = 50 + 10 * (size_t)depth;
    if (fin.residual > split_threshold && depth < cfg.max_split_depth && n >= min_patch_size) {
        float c[3];
        centroid_of(pts, c);
        float var_x = 0.0f, var_y = 0.0f;
        for (const P3& p : pts) {
            const float dx = p.x - c[0];
            const float dy = p.y - c[1];
            var_x += dx * dx;
            var_y += dy * dy;
        }
        var_x /= (float)n;
        var_y /= (float)n;
        const int axis = (var_x > var_y) ? 0 : 1;
        std::vector<float> vals(n);
        for (size_t i = 0; i < n; ++i) vals[i] = axis == 0 ? pts[i].x : pts[i].y;
        {
            std::vector<float> sorted_vals = vals;
            std::sort(sorted_vals.begin(), sorted_vals.end());
            const float median = sorted_vals[n / 2];
            std::vector<P3> left, right;
            left.reserve(n / 2 + 1); right.reserve(n / 2 + 1);
            for (size_t i = 0; i < n; ++i) (vals[i] <= median ? left : right).push_back(pts[i]);
            emit(cx, depth, start, (uint32_t)n, RPWO_NODE_SPLIT, iterations, (uint32_t)inl.size(), axis, median, &fin);
            cx.st->n_splits++;
            // children: SAME mean_dist; result is left_result ++ right_result BY POSITION (RP.cpp:290-304)
            std::vector<bool> res_l = fit_plane_and_split(cx, left, mean_dist, depth + 1, start);
            std::vector<bool> res_r = fit_plane_and_split(cx, right, mean_dist, depth + 1, start + (uint32_t)left.size());
            std::vector<bool> result(n, false);
            size_t li = 0, ri = 0;
            for (size_t i = 0; i < n; ++i) {
                if (vals[i] <= median) { result[li] = res_l[li]; ++li; }
                else                   { result[ri + left.size()] = res_r[ri]; ++ri; }
            }
            return result;
        }
    }

    // FIT leaf: remember which final plane produced each slot of this call
    if (cx.slot_leaf) {
        const int leaf_id = (int)cx.leaf_planes->size();
        cx.leaf_planes->push_back(fin);
        for (size_t i = 0; i < n; ++i) (*cx.slot_leaf)[start + i] = leaf_id;
    }
    emit(cx, depth, start, (uint32_t)n, RPWO_NODE_FIT, iterations, (uint32_t)inl.size(), -1, 0.0f, &fin);
    return ground;
}

// cuda_interface.cu:617-632 (CPU branch): atan2f, wrap with a double add
inline float angle_of(float x, float y)
{
    float a = std::atan2(y, x);
    if (a < 0) a += 2.0f * M_PI;
    return a;
}

void make_tables(const rpwo_config& cfg, float edges[9], std::vector<float>& bounds)
{
    const float r_min = 1.0f, r_max = cfg.filtering_radius;
    const int num_rings = 8;
    for (int i = 0; i <= num_rings; ++i)
        edges[i] = r_min * std::pow(r_max / r_min, static_cast<float>(i) / num_rings);
    const float sector_angle = 2.0f * M_PI / cfg.num_sectors;
    const int S = cfg.num_sectors > 0 ? cfg.num_sectors : 0;
    bounds.resize((size_t)S + 1);
    for (int k = 0; k <= S; ++k) bounds[k] = k * sector_angle;
}

} // namespace

extern "C" {

// LidarFusion::fuseLidarPointClouds restated (reference src/lidar_fusion.cpp:42-86,88-126,148-159,184-187):
// per sensor: yaw rotation when |angle| > 1e-6 (cosf/sinf of the float radians, x*c - y*s, x*s + y*c), then
// removeEgoVehicle (points with sqrtf(x*x + y*y) <= radius are dropped), then concatenation in sensor order.
// xyz holds the raw clouds back to back, sensor k = [sensor_off[k], sensor_off[k+1]). out_src[j] = raw index.
int rpwo_fuse(const float* xyz, const uint32_t* sensor_off, uint32_t n_sensors, const float* rotation_deg,
              const float* ego_radius, float* out_xyz, uint32_t* out_src, uint32_t* n_out)
{
    uint32_t m = 0;
    for (uint32_t k = 0; k < n_sensors; ++k) {
        const float angle_degrees = rotation_deg[k];
        const bool rotate = std::abs(angle_degrees) > 1e-6f;
        float angle_rad = angle_degrees * M_PI / 180.0f;
        float cos_a = std::cos(angle_rad);
        float sin_a = std::sin(angle_rad);
        for (uint32_t i = sensor_off[k]; i < sensor_off[k + 1]; ++i) {
            float x = xyz[3 * i], y = xyz[3 * i + 1];
            const float z = xyz[3 * i + 2];
            if (rotate) {
                const float rx = x * cos_a - y * sin_a;
                const float ry = x * sin_a + y * cos_a;
                x = rx; y = ry;
            }
            const float distance_2d = std::sqrt(x * x + y * y);
            if (distance_2d <= ego_radius[k]) continue;
            out_xyz[3 * m] = x; out_xyz[3 * m + 1] = y; out_xyz[3 * m + 2] = z;
            if (out_src) out_src[m] = i;
            ++m;
        }
    }
    *n_out = m;
    return 0;
}

void rpwo_set_iter_trace(uint64_t* buf, uint32_t cap) { g_iter_trace = buf; g_iter_cap = cap; g_iter_n = 0; }
uint32_t rpwo_iter_trace_count(void) { return g_iter_n; }

int rpwo_eig3(const float* cov, float* evals, float* evecs) { return eig3(cov, evals, evecs); }

void rpwo_fit_plane(const float* xyz, uint32_t n, float* out7)
{
    std::vector<P3> pts(n);
    for (uint32_t i = 0; i < n; ++i) pts[i] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    Plane p = fit_plane_pca(pts);
    for (int i = 0; i < 3; ++i) { out7[i] = p.c[i]; out7[3 + i] = p.n[i]; }
    out7[6] = p.residual;
}

void rpwo_tables(const rpwo_config* cfg, float* edges9, float* bounds)
{
    std::vector<float> b;
    make_tables(*cfg, edges9, b);
    for (size_t i = 0; i < b.size(); ++i) bounds[i] = b[i];
}

float rpwo_angle(float x, float y) { return angle_of(x, y); }

int rpwo_segment(const rpwo_config* cfg, const float* xyz, uint32_t n, uint8_t* labels,
                 float* residual, rpwo_node* nodes, uint32_t nodes_cap, rpwo_stats* stats)
{
    if (!cfg || (n && (!xyz || !labels))) return 1;
    rpwo_stats local;
    rpwo_stats* st = stats ? stats : &local;
    std::memset(st, 0, sizeof(*st));
    st->n_in = n;
    if (residual) for (uint32_t i = 0; i < n; ++i) residual[i] = -1.0f;

    // cleanPoints (RP.cpp:19-35)
    std::vector<P3> cleaned; cleaned.reserve(n);
    std::vector<uint32_t> cleaned_src; cleaned_src.reserve(n);
    for (uint32_t i = 0; i < n; ++i) {
        const P3 p = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        if (std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z)) {
            cleaned.push_back(p); cleaned_src.push_back(i);
            labels[i] = RPWO_NONGROUND;
        } else {
            labels[i] = RPWO_DROPPED;
        }
    }
    st->n_clean = (uint32_t)cleaned.size();
    const auto t_start = std::chrono::high_resolution_clock::now();
    if (g_chatter && cleaned.size() < n)
        std::cout << "Removed " << (n - cleaned.size()) << " points with NaN/inf values" << std::endl;
    if (cleaned.empty()) return 0;

    // computeDistances2D (cuda_interface.cu:582-595), filterPointsByRadius (:597-615), RP.cpp:328-334
    if (g_chatter) { say_unavailable(); std::cout << "[CPU] Computing 2D distances for " << cleaned.size() << " points..." << std::endl; }
    std::vector<float> dist(cleaned.size());
    for (size_t i = 0; i < cleaned.size(); ++i)
        dist[i] = std::sqrt(cleaned[i].x * cleaned[i].x + cleaned[i].y * cleaned[i].y);
    if (g_chatter) {
        std::cout << "[CPU] 2D distance computation completed successfully" << std::endl;
        say_unavailable();
        std::cout << "[CPU] Filtering " << cleaned.size() << " points by radius " << cfg->filtering_radius << "..." << std::endl;
    }
    std::vector<P3> zone; std::vector<float> d_zone; std::vector<uint32_t> zone_src;
    zone.reserve(cleaned.size()); d_zone.reserve(cleaned.size()); zone_src.reserve(cleaned.size());
    for (size_t i = 0; i < cleaned.size(); ++i) {
        if (dist[i] <= cfg->filtering_radius) {
            zone.push_back(cleaned[i]); d_zone.push_back(dist[i]); zone_src.push_back(cleaned_src[i]);
        } else {
            labels[cleaned_src[i]] = RPWO_OUTSIDE;
        }
    }
    st->n_zone = (uint32_t)zone.size();
    if (g_chatter)
        std::cout << "[CPU] Point filtering completed successfully. Filtered: " << zone.size() << "/" << cleaned.size() << " points" << std::endl;
    if (zone.size() < 3) return 0;   // RP.cpp:339-341: everything cleaned is non-ground

    float edges[9];
    std::vector<float> bounds;
    make_tables(*cfg, edges, bounds);
    const int num_rings = 8;

    if (g_chatter) { say_unavailable(); std::cout << "[CPU] Computing angles for " << zone.size() << " points..." << std::endl; }
    std::vector<float> angles(zone.size());
    for (size_t i = 0; i < zone.size(); ++i) angles[i] = angle_of(zone[i].x, zone[i].y);
    if (g_chatter) std::cout << "[CPU] Angle computation completed successfully" << std::endl;

    std::vector<bool> is_ground(zone.size(), false);
    std::vector<P3> patch_pts;
    std::vector<size_t> patch_idx;
    std::vector<int> slot_leaf;
    std::vector<Plane> leaf_planes;

    for (int ring = 0; ring < num_rings; ++ring) {
        const float r0 = edges[ring], r1 = edges[ring + 1];
        for (int sector = 0; sector < cfg->num_sectors; ++sector) {
            const float a0 = bounds[sector], a1 = bounds[sector + 1];
            patch_pts.clear(); patch_idx.clear();
            for (size_t i = 0; i < zone.size(); ++i) {
                if (d_zone[i] >= r0 && d_zone[i] < r1 && angles[i] >= a0 && angles[i] < a1) {
                    patch_pts.push_back(zone[i]); patch_idx.push_back(i);
                }
            }
            if (patch_pts.empty()) continue;
            st->n_patches++;
            float mean_dist = 0.0f;
            for (size_t idx : patch_idx) mean_dist += d_zone[idx];
            mean_dist /= (float)patch_idx.size();

            Ctx cx;
            cx.cfg = cfg; cx.nodes = nodes; cx.nodes_cap = nodes_cap; cx.st = st;
            cx.patch = ring * cfg->num_sectors + sector;
            slot_leaf.assign(patch_pts.size(), -1);
            leaf_planes.clear();
            cx.slot_leaf = residual ? &slot_leaf : nullptr;
            cx.leaf_planes = &leaf_planes;
            std::vector<bool> mask = fit_plane_and_split(cx, patch_pts, mean_dist, 0, 0);
            for (size_t i = 0; i < patch_idx.size(); ++i) {
                if (mask[i]) is_ground[patch_idx[i]] = true;
                if (residual && slot_leaf[i] >= 0) {
                    const Plane& pl = leaf_planes[(size_t)slot_leaf[i]];
                    // the point that RECEIVES slot i's label (RP.cpp:393-397), not the point a
                    // split may have moved into slot i
                    residual[zone_src[patch_idx[i]]] = plane_distance(zone[patch_idx[i]], pl.c, pl.n);
                }
            }
        }
    }
    uint32_t ng = 0;
    for (size_t i = 0; i < zone.size(); ++i)
        if (is_ground[i]) { labels[zone_src[i]] = RPWO_GROUND; ++ng; }
    st->n_ground = ng;
    if (g_chatter) {                             // RP.cpp:421-423
        const double secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t_start).count();
        std::cout << "Ground filtering completed in " << secs << " seconds" << std::endl;
        std::cout << "Ground points: " << ng << ", Non-ground points: " << (cleaned.size() - ng) << std::endl;
    }
    return 0;
}

int rpwo_filter_ground_points(const rpwo_config* cfg, const float* xyz, uint32_t n,
                              float* ground_xyz, uint32_t* n_ground,
                              float* nonground_xyz, uint32_t* n_nonground)
{
    std::vector<uint8_t> labels(n);
    rpwo_stats st;
    int rc = rpwo_segment(cfg, xyz, n, labels.data(), nullptr, nullptr, 0, &st);
    if (rc) return rc;
    uint32_t g = 0, ng = 0;
    auto push = [&](float* dst, uint32_t& k, uint32_t i) {
        dst[3 * k] = xyz[3 * i]; dst[3 * k + 1] = xyz[3 * i + 1]; dst[3 * k + 2] = xyz[3 * i + 2]; ++k;
    };
    if (st.n_clean == 0) { *n_ground = 0; *n_nonground = 0; return 0; }
    if (st.n_zone < 3) {   // {{}, cleaned_points}
        for (uint32_t i = 0; i < n; ++i) if (labels[i] != RPWO_DROPPED) push(nonground_xyz, ng, i);
        *n_ground = 0; *n_nonground = ng; return 0;
    }
    for (uint32_t i = 0; i < n; ++i) {
        if (labels[i] == RPWO_GROUND) push(ground_xyz, g, i);
        else if (labels[i] == RPWO_NONGROUND) push(nonground_xyz, ng, i);
    }
    for (uint32_t i = 0; i < n; ++i) if (labels[i] == RPWO_OUTSIDE) push(nonground_xyz, ng, i);
    *n_ground = g; *n_nonground = ng;
    return 0;
}

void rpwo_set_chatter(int on) { g_chatter = on ? 1 : 0; }

// sampleGroundAndObstacles (RP.cpp:428-465) with PointCloudProcessor::randomSubsample (PCP.cpp:121-146) restated.
// The reference seeds std::mt19937 from std::random_device; here the seed is an argument, everything else is the same:
// uniform_int_distribution<size_t>(0, size - 1) draws, rejection of repeats, points appended in draw order.
int rpwo_sample_ground_and_obstacles(const rpwo_config* cfg, const float* xyz, uint32_t n, float target_height,
                                     float base_tol, uint32_t seed, float* out_xyz, uint32_t* n_out)
{
    std::vector<float> g((size_t)n * 3 + 3), ng((size_t)n * 3 + 3);
    uint32_t cg = 0, cn = 0;
    int rc = rpwo_filter_ground_points(cfg, xyz, n, g.data(), &cg, ng.data(), &cn);
    if (rc) return rc;
    uint32_t k = 0;
    auto push = [&](const float* p) { out_xyz[3 * k] = p[0]; out_xyz[3 * k + 1] = p[1]; out_xyz[3 * k + 2] = p[2]; ++k; };
    if (cn == 0) {                                                     // RP.cpp:434-436: return ground_points
        for (uint32_t i = 0; i < cg; ++i) push(&g[3 * (size_t)i]);
        *n_out = k; return 0;
    }
    // ground sample first (RP.cpp:451-462), then the obstacle points
    if (cg > 0) {
        const size_t target = std::min<size_t>(2000, cg);
        if (cg <= target) {
            for (uint32_t i = 0; i < cg; ++i) push(&g[3 * (size_t)i]);
        } else {
            std::mt19937 gen(seed);
            std::uniform_int_distribution<size_t> dis(0, (size_t)cg - 1);
            std::vector<bool> selected(cg, false);
            size_t count = 0;
            while (count < target) {
                const size_t idx = dis(gen);
                if (!selected[idx]) { selected[idx] = true; push(&g[3 * idx]); ++count; }
            }
        }
    }
    for (uint32_t i = 0; i < cn; ++i) {
        const float* p = &ng[3 * (size_t)i];
        if (!(std::sqrt(p[0] * p[0] + p[1] * p[1]) > 2.5f)) continue;    // removeEgoVehicle(non_ground, 2.5f), RP.cpp:64-75
        if (std::abs(p[2] - target_height) <= base_tol) push(p);         // RP.cpp:445-449
    }
    *n_out = k;
    return 0;
}

// CPU baseline harness: no Python in the loop. Phase 1: one thread, every frame once after warm-ups, per-frame times
// (median reported). Phase 2: n_threads std::threads, thread t segments frames t, t + n_threads, ... round and round
// until `seconds` have passed; frames finished / wall time = frames per second on n_threads threads.
int rpwo_bench_threads(const rpwo_config* cfg, const float* xyz, const uint32_t* frame_off, uint32_t n_frames,
                       uint32_t n_threads, double seconds, uint32_t single_reps, rpwo_bench_result* out)
{
    if (!cfg || !xyz || !frame_off || !out || n_frames == 0) return 1;
    if (n_threads == 0) n_threads = std::thread::hardware_concurrency();
    if (n_threads == 0) n_threads = 1;
    // The path allocates and frees multi-megabyte vectors all the time (as the reference does). Left at glibc's defaults
    // each of them is an mmap / munmap pair, and hundreds of threads then queue on the process's address-space lock
    // instead of computing. The baseline gets its best shot: big blocks stay in the heap, freed memory is kept.
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_ARENA_MAX, (int)(2 * n_threads));
    auto run = [&](uint32_t f, std::vector<uint8_t>& lab) {
        const uint32_t nf = frame_off[f + 1] - frame_off[f];
        if (lab.size() < nf) lab.resize(nf);
        rpwo_stats st;
        (void)rpwo_segment(cfg, xyz + 3 * (size_t)frame_off[f], nf, lab.data(), nullptr, nullptr, 0, &st);
    };
    using clk = std::chrono::steady_clock;
    std::vector<uint8_t> lab;
    for (uint32_t w = 0; w < 3 && w < n_frames; ++w) run(w, lab);          // warm-ups
    std::vector<double> t;
    const uint32_t reps = single_reps ? single_reps : 20;
    for (uint32_t r = 0; r < reps; ++r) {
        const auto a = clk::now();
        run(r % n_frames, lab);
        t.push_back(std::chrono::duration<double, std::milli>(clk::now() - a).count());
    }
    std::sort(t.begin(), t.end());
    out->single_thread_median_ms = t[t.size() / 2];
    out->single_thread_frames = (uint32_t)t.size();
    out->hardware_concurrency = std::thread::hardware_concurrency();
    out->threads = n_threads;
    std::atomic<uint64_t> done{0};
    std::atomic<int> stop{0};
    std::vector<std::thread> th;
    const auto t0 = clk::now();
    for (uint32_t k = 0; k < n_threads; ++k)
        th.emplace_back([&, k]() {
            std::vector<uint8_t> l;
            for (uint32_t f = k % n_frames; !stop.load(std::memory_order_relaxed); f = (f + n_threads) % n_frames) {
                run(f, l);
                done.fetch_add(1, std::memory_order_relaxed);
            }
        });
    while (std::chrono::duration<double>(clk::now() - t0).count() < seconds) std::this_thread::sleep_for(std::chrono::milliseconds(5));
    stop.store(1);
    for (auto& x : th) x.join();                                             // (frames in progress at the bell are finished and counted)
    out->elapsed_s = std::chrono::duration<double>(clk::now() - t0).count();
    out->frames_done = done.load();
    return 0;
}

} // extern "C"
