// rpw_math.h -- scalar building blocks shared by the HIP kernels (device) and by the
// host-side test hooks: the libm atan2f model, the polar classification and the 3x3
// symmetric eigensolver. Every function is strict binary32: this file must be compiled
// with -ffp-contract=off and with correctly rounded fp32 divide / sqrt (hipcc default).
//
// Reference behaviour matched here (paths under /root/reference/src/recursive_patchwork):
//   cuda/cuda_interface.cu:582-632         sqrt(x*x+y*y), d <= R, atan2 + wrap by a double add
//   src/recursive_patchwork.cpp:344-378    ring / sector membership, half-open intervals
//   src/recursive_patchwork.cpp:89-95      SelfAdjointEigenSolver<Matrix3f>, col(0), sign flip
// Third-party arithmetic restated (not in /root/reference):
//   glibc 2.35 libm (Ubuntu 22.04) e_atan2f.c / s_atanf.c  -- what std::atan2(float,float) runs
//   Eigen 3.4.0 SelfAdjointEigenSolver.h / Tridiagonalization.h / Jacobi.h
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define RPW_HD __host__ __device__ __forceinline__
#else
#define RPW_HD inline
#endif

namespace rpw {

RPW_HD float u2f(uint32_t u) { union { uint32_t u; float f; } c; c.u = u; return c.f; }
RPW_HD uint32_t f2u(float f) { union { uint32_t u; float f; } c; c.f = f; return c.u; }
RPW_HD float fabs_(float x) { return u2f(f2u(x) & 0x7fffffffu); }
RPW_HD bool finite_(float x) { return (f2u(x) & 0x7f800000u) != 0x7f800000u; }

// Correctly rounded binary32 sqrt. On gfx950 hipcc lowers sqrtf() to v_sqrt_f32 plus a
// +-1 ulp fma-residual correction (the default -fhip-fp32-correctly-rounded-divide-sqrt);
// __fsqrt_rn() is NOT corrected there (bare v_sqrt_f32, 1 ulp), so it must not be used.
RPW_HD float sqrt_rn(float x) { return __builtin_sqrtf(x); }

// ---------------------------------------------------------------------------------------
// atanf / atan2f exactly as glibc 2.35's flt-32 implementation evaluates them (fdlibm
// lineage): same argument reduction, same polynomial split, same operation order.
// ---------------------------------------------------------------------------------------
RPW_HD float atanf_model(float x)
{
    const float atanhi0 = u2f(0x3eed6338u), atanhi1 = u2f(0x3f490fdau), atanhi2 = u2f(0x3f7b985eu), atanhi3 = u2f(0x3fc90fdau);
    const float atanlo0 = u2f(0x31ac3769u), atanlo1 = u2f(0x33222168u), atanlo2 = u2f(0x33140fb4u), atanlo3 = u2f(0x33a22168u);
    const float aT0 = u2f(0x3eaaaaabu), aT1 = u2f(0xbe4ccccdu), aT2 = u2f(0x3e124925u), aT3 = u2f(0xbde38e38u),
                aT4 = u2f(0x3dba2e6eu), aT5 = u2f(0xbd9d8795u), aT6 = u2f(0x3d886b35u), aT7 = u2f(0xbd6ef16bu),
                aT8 = u2f(0x3d4bda59u), aT9 = u2f(0xbd15a221u), aT10 = u2f(0x3c8569d7u);
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    float hi = 0.0f, lo = 0.0f;
    if (ix >= 0x4c000000) {                 // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;  // NaN
        return hx > 0 ? atanhi3 + atanlo3 : -atanhi3 - atanlo3;
    }
    if (ix < 0x3ee00000) {                  // |x| < 0.4375
        if (ix < 0x31000000) return x;      // |x| < 2^-29
        id = -1;
    } else {
        x = fabs_(x);
        if (ix < 0x3f980000) {              // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; hi = atanhi0; lo = atanlo0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else                 { id = 1; hi = atanhi1; lo = atanlo1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; hi = atanhi2; lo = atanlo2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else                 { id = 3; hi = atanhi3; lo = atanlo3; x = -1.0f / x; }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}

RPW_HD float atan2f_model(float y, float x)
{
    const float pi_o_2 = u2f(0x3fc90fdbu), pi = u2f(0x40490fdbu), pi_lo = u2f(0xb3bbbd2eu);
    const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;       // NaN
    if (hx == 0x3f800000) return atanf_model(y);                // x == 1.0
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);           // 2*sign(x) + sign(y)
    if (iy == 0) {                                              // y == +-0
        if (m < 2) return y;
        return m == 2 ? pi : -pi;                               // pi + tiny == pi in binary32
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 : pi_o_2;              // x == +-0
    if (ix == 0x7f800000) {                                     // x == inf (not reachable after cleanPoints)
        const float pi_o_4 = u2f(0x3f490fdbu);
        if (iy == 0x7f800000) {
            switch (m) { case 0: return pi_o_4; case 1: return -pi_o_4; case 2: return 3.0f * pi_o_4; default: return -3.0f * pi_o_4; }
        }
        switch (m) { case 0: return 0.0f; case 1: return -0.0f; case 2: return pi; default: return -pi; }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 : pi_o_2;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = atanf_model(fabs_(y / x));
    switch (m) {
        case 0:  return z;
        case 1:  return u2f(f2u(z) ^ 0x80000000u);
        case 2:  return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// cuda_interface.cu:625-626: angle = atan2(y, x); if (angle < 0) angle += 2.0f * M_PI  (double add, one rounding)
RPW_HD float polar_angle(float x, float y)
{
    float a = atan2f_model(y, x);
    if (a < 0.0f) a = (float)((double)a + 6.283185307179586);
    return a;
}

// ---------------------------------------------------------------------------------------
// Eigen 3.4.0 SelfAdjointEigenSolver<Matrix3f>: scale, 3x3 tridiagonalisation, implicit
// symmetric QR with Wilkinson shift, ascending sort. All state lives in named scalars so
// that nothing is indexed at run time (no scratch memory on the device).
// ---------------------------------------------------------------------------------------
struct Eig3 {
    float d0, d1, d2;      // diagonal
    float e0, e1;          // sub-diagonal
    float q00, q10, q20, q01, q11, q21, q02, q12, q22;   // q<row><col>
};

RPW_HD float hypot_eigen(float a, float b)
{
    const float x = fabs_(a), y = fabs_(b);
    // inf / nan inputs cannot occur: the matrix was scaled to [-1, 1]
    const float p = x > y ? x : y;
    if (p == 0.0f) return 0.0f;
    const float q = (y < x ? y : x) / p;
    return p * sqrt_rn(1.0f + q * q);
}

RPW_HD void givens(float p, float q, float& c, float& s)
{
    if (q == 0.0f) { c = p < 0.0f ? -1.0f : 1.0f; s = 0.0f; }
    else if (p == 0.0f) { c = 0.0f; s = q < 0.0f ? 1.0f : -1.0f; }
    else if (fabs_(p) > fabs_(q)) {
        const float t = q / p;
        float u = sqrt_rn(1.0f + t * t);
        if (p < 0.0f) u = -u;
        c = 1.0f / u;
        s = -t * c;
    } else {
        const float t = p / q;
        float u = sqrt_rn(1.0f + t * t);
        if (q < 0.0f) u = -u;
        s = -1.0f / u;
        c = -t * s;
    }
}

RPW_HD float wilkinson_mu(float dm1, float dend, float e)
{
    const float td = (dm1 - dend) * 0.5f;
    float mu = dend;
    if (td == 0.0f) {
        mu -= fabs_(e);
    } else if (e != 0.0f) {
        const float e2 = e * e;
        const float h = hypot_eigen(td, e);
        if (e2 == 0.0f) mu -= e / ((td + (td > 0.0f ? h : -h)) / e);
        else            mu -= e2 / (td + (td > 0.0f ? h : -h));
    }
    return mu;
}

// one Givens step on rows/cols (k, k+1) of the tridiagonal matrix and on columns (k, k+1) of Q
#define RPW_ROT_COLS(QA0, QA1, QA2, QB0, QB1, QB2)                                  \
    if (!(c == 1.0f && s == 0.0f)) {                                                \
        float xi, yi;                                                               \
        xi = QA0; yi = QB0; QA0 = c * xi - s * yi; QB0 = s * xi + c * yi;           \
        xi = QA1; yi = QB1; QA1 = c * xi - s * yi; QB1 = s * xi + c * yi;           \
        xi = QA2; yi = QB2; QA2 = c * xi - s * yi; QB2 = s * xi + c * yi;           \
    }

// tridiagonal_qr_step(start = 0, end = 2)
RPW_HD void qr_step_02(Eig3& m)
{
    const float mu = wilkinson_mu(m.d1, m.d2, m.e1);
    float x = m.d0 - mu;
    float z = m.e0;
    if (z != 0.0f) {                                   // k = 0
        float c, s;
        givens(x, z, c, s);
        const float sdk = s * m.d0 + c * m.e0;
        const float dkp1 = s * m.e0 + c * m.d1;
        m.d0 = c * (c * m.d0 - s * m.e0) - s * (c * m.e0 - s * m.d1);
        m.d1 = s * sdk + c * dkp1;
        m.e0 = c * sdk - s * dkp1;
        x = m.e0;
        z = -s * m.e1;                                 // k < end - 1
        m.e1 = c * m.e1;
        RPW_ROT_COLS(m.q00, m.q10, m.q20, m.q01, m.q11, m.q21)
        if (z != 0.0f) {                               // k = 1
            givens(x, z, c, s);
            const float sdk1 = s * m.d1 + c * m.e1;
            const float dkp2 = s * m.e1 + c * m.d2;
            m.d1 = c * (c * m.d1 - s * m.e1) - s * (c * m.e1 - s * m.d2);
            m.d2 = s * sdk1 + c * dkp2;
            m.e1 = c * sdk1 - s * dkp2;
            m.e0 = c * m.e0 - s * z;                   // k > start
            RPW_ROT_COLS(m.q01, m.q11, m.q21, m.q02, m.q12, m.q22)
        }
    }
}

// tridiagonal_qr_step(start = 1, end = 2)
RPW_HD void qr_step_12(Eig3& m)
{
    const float mu = wilkinson_mu(m.d1, m.d2, m.e1);
    const float x = m.d1 - mu;
    const float z = m.e1;
    if (z != 0.0f) {
        float c, s;
        givens(x, z, c, s);
        const float sdk = s * m.d1 + c * m.e1;
        const float dkp1 = s * m.e1 + c * m.d2;
        m.d1 = c * (c * m.d1 - s * m.e1) - s * (c * m.e1 - s * m.d2);
        m.d2 = s * sdk + c * dkp1;
        m.e1 = c * sdk - s * dkp1;
        RPW_ROT_COLS(m.q01, m.q11, m.q21, m.q02, m.q12, m.q22)
    }
}

// tridiagonal_qr_step(start = 0, end = 1)
RPW_HD void qr_step_01(Eig3& m)
{
    const float mu = wilkinson_mu(m.d0, m.d1, m.e0);
    const float x = m.d0 - mu;
    const float z = m.e0;
    if (z != 0.0f) {
        float c, s;
        givens(x, z, c, s);
        const float sdk = s * m.d0 + c * m.e0;
        const float dkp1 = s * m.e0 + c * m.d1;
        m.d0 = c * (c * m.d0 - s * m.e0) - s * (c * m.e0 - s * m.d1);
        m.d1 = s * sdk + c * dkp1;
        m.e0 = c * sdk - s * dkp1;
        RPW_ROT_COLS(m.q00, m.q10, m.q20, m.q01, m.q11, m.q21)
    }
}
#undef RPW_ROT_COLS

RPW_HD bool deflate(float& e, float da, float db)
{
    const float considerAsZero = 1.17549435e-38f;          // FLT_MIN
    const float precision_inv = 8388608.0f;                 // 1 / FLT_EPSILON = 2^23
    if (fabs_(e) < considerAsZero) { e = 0.0f; }
    else {
        const float sc = precision_inv * e;
        if (sc * sc <= (fabs_(da) + fabs_(db))) e = 0.0f;
    }
    return e == 0.0f;
}

// Input: the six lower-triangle entries of the covariance. Output: eigenvector of the
// smallest eigenvalue = eigenvectors().col(0), flipped so that z >= 0 as
// recursive_patchwork.cpp:93-95 does (strictly negative z only).
RPW_HD void smallest_eigenvector(float a00, float a10, float a11, float a20, float a21, float a22,
                                 float& nx, float& ny, float& nz)
{
    float scale = fabs_(a00);
    { float t;
      t = fabs_(a10); if (t > scale) scale = t;
      t = fabs_(a20); if (t > scale) scale = t;
      t = fabs_(a11); if (t > scale) scale = t;
      t = fabs_(a21); if (t > scale) scale = t;
      t = fabs_(a22); if (t > scale) scale = t; }
    if (scale == 0.0f) scale = 1.0f;
    a00 /= scale; a10 /= scale; a11 /= scale; a20 /= scale; a21 /= scale; a22 /= scale;

    Eig3 m;
    m.d0 = a00;
    const float v1norm2 = a20 * a20;
    if (v1norm2 <= 1.17549435e-38f) {
        m.d1 = a11; m.d2 = a22; m.e0 = a10; m.e1 = a21;
        m.q00 = 1; m.q10 = 0; m.q20 = 0; m.q01 = 0; m.q11 = 1; m.q21 = 0; m.q02 = 0; m.q12 = 0; m.q22 = 1;
    } else {
        const float beta = sqrt_rn(a10 * a10 + v1norm2);
        const float invBeta = 1.0f / beta;
        const float m01 = a10 * invBeta;
        const float m02 = a20 * invBeta;
        const float q = 2.0f * m01 * a21 + m02 * (a22 - a11);
        m.d1 = a11 + m02 * q;
        m.d2 = a22 - m02 * q;
        m.e0 = beta;
        m.e1 = a21 - m01 * q;
        m.q00 = 1; m.q01 = 0;   m.q02 = 0;
        m.q10 = 0; m.q11 = m01; m.q12 = m02;
        m.q20 = 0; m.q21 = m02; m.q22 = -m01;
    }

    int end = 2, start = 0, iter = 0;
    bool converged = true;
    while (end > 0) {
        // deflation test for i in [start, end)
        if (start == 0) { deflate(m.e0, m.d0, m.d1); if (end == 2) deflate(m.e1, m.d1, m.d2); }
        else if (end == 2) { deflate(m.e1, m.d1, m.d2); }      // start == 1: only i = 1 is in [start, end)
        while (end > 0 && (end == 2 ? m.e1 : m.e0) == 0.0f) end--;
        if (end <= 0) break;
        iter++;
        if (iter > 90) { converged = false; break; }
        start = end - 1;
        while (start > 0 && (start == 1 ? m.e0 : 0.0f) != 0.0f) start--;
        if (end == 2) { if (start == 0) qr_step_02(m); else qr_step_12(m); }
        else          { qr_step_01(m); }
    }
    if (converged) {
        // selection sort, first minimum, swapping eigenvalues and columns; only column 0 is needed
        // i = 0: k = argmin(d0, d1, d2)
        int k = 0; float best = m.d0;
        if (m.d1 < best) { best = m.d1; k = 1; }
        if (m.d2 < best) { best = m.d2; k = 2; }
        if (k == 1) { nx = m.q01; ny = m.q11; nz = m.q21; }
        else if (k == 2) { nx = m.q02; ny = m.q12; nz = m.q22; }
        else { nx = m.q00; ny = m.q10; nz = m.q20; }
    } else {
        nx = m.q00; ny = m.q10; nz = m.q20;            // NoConvergence: no sort
    }
    if (nz < 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
}

// ---------------------------------------------------------------------------------------
// polar classification (recursive_patchwork.cpp:315-378 per point)
//
// The reference compares d = sqrtf(x*x + y*y) with the ring edges and with the radius. sqrtf is
// correctly rounded and monotone, so every such comparison is equivalent to a comparison of
// s = x*x + y*y (same two products, same add) with a threshold the host derives exactly:
//   d >= e   <=>   s >= ge[e]      with ge[e] = min { s : sqrtf(s) >= e }
//   d <= R   <=>   s <  gt_radius  with gt_radius = min { s : sqrtf(s) > R }
// For the sector, a cheap angle (error < 2e-5 rad) decides every point that is farther than 1e-4 rad
// from a sector bound; the others (a few per 10^4) take the exact libm-model path.
// ---------------------------------------------------------------------------------------
struct PolarTables {
    float edges[9];          // ring_edges[0..8] (used by the literal fallback)
    float ge[9];             // ge[i] = min { s : sqrtf(s) >= edges[i] }
    float gt_radius;         // min { s : sqrtf(s) > filtering_radius }
    float radius;            // filtering_radius
    float inv_sector;        // 1 / sector_angle, only a search hint
    int   num_sectors;
    int   edges_monotone;    // edges non-decreasing (always, unless filtering_radius < 1)
    // cheap angle in SECTOR UNITS (angle / sector_angle): odd degree-9 polynomial of min/max, coefficients
    // premultiplied by inv_sector; quarter / half / full turn in the same units; half-width of the sure band
    float ac[5];
    float uq, uh, uf;
    float usure;             // a point is decided by the cheap angle iff |frac(u) - 0.5| < usure  (1e-4 rad from either bound)
    // sector bounds bounds[k] = (float)k * sector_angle live in a separate array (device memory / LDS)
};

RPW_HD float fract_(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fractf(x);
#else
    return x - __builtin_floorf(x);
#endif
}

RPW_HD float fast_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

// exact sector of angle a, or -1: the unique k with bounds[k] <= a < bounds[k+1]
RPW_HD int sector_exact(float a, int S, float inv_sector, const float* __restrict__ bounds)
{
    int k = (int)(a * inv_sector);
    if (k > S - 1) k = S - 1;
    if (k < 0) k = 0;
    while (k > 0 && a < bounds[k]) --k;
    while (k < S && !(a < bounds[k + 1])) ++k;
    if (k >= S || !(a >= bounds[k] && a < bounds[k + 1])) return -1;
    return k;
}

// returns ring*S + sector, or P + 0 (in zone, no patch), P + 1 (outside radius), P + 2 (non-finite).
// `uncertain` is set when the cheap angle cannot decide the sector; the caller then calls
// classify_point_exact for that point. Straight-line code (selects, no branches).
RPW_HD int classify_point_fast(float x, float y, float z, const PolarTables& t, const float* __restrict__ bounds, bool& uncertain)
{
    const int S = t.num_sectors;
    const int P = 8 * S;
    const bool fin = finite_(x) && finite_(y) && finite_(z);
    const float s = x * x + y * y;
    const bool inzone = s < t.gt_radius;
    int ring;
    if (t.edges_monotone) {
        int c = 0;                      // ring = #{i : s >= ge[i]} - 1, valid if it is in 0..7
#pragma unroll
        for (int i = 0; i < 9; ++i) c += (s >= t.ge[i]) ? 1 : 0;
        ring = (c >= 1 && c <= 8) ? c - 1 : -1;
    } else {
        ring = -1;
#pragma unroll
        for (int i = 0; i < 8; ++i) ring = (s >= t.ge[i] && s < t.ge[i + 1]) ? i : ring;
    }
    // cheap angle, in sector units u = angle / sector_angle in [0, S]: |error| < 2e-5 rad (degree-9 odd polynomial on
    // [0,1], 1-ulp reciprocal). Sector = floor(u); the point is decided here only if u is farther than 1e-4 rad from
    // both neighbouring integers (the reference's bounds[k] = k * sector_angle are within 2e-6 rad of them).
    const float ax = fabs_(x), ay = fabs_(y);
    const bool steep = ay > ax;
    const float mx = steep ? ay : ax, mn = steep ? ax : ay;
    const float q = mn * fast_rcp(mx);
    const float q2 = q * q;
    // this path is approximate by design, so fused multiply-adds are welcome here
    float u = q * __builtin_fmaf(q2, __builtin_fmaf(q2, __builtin_fmaf(q2, __builtin_fmaf(q2, t.ac[4], t.ac[3]), t.ac[2]), t.ac[1]), t.ac[0]);
    u = steep ? t.uq - u : u;
    u = x < 0.0f ? t.uh - u : u;
    u = y < 0.0f ? t.uf - u : u;
    int k = (int)u;
    k = k > S - 1 ? S - 1 : k;
    k = k < 0 ? 0 : k;                   // also maps a NaN angle (x = y = 0 or non-finite input) to 0
    const bool sure = fabs_(fract_(u) - 0.5f) < t.usure;      // false for NaN and for u >= S (then frac ~ 0)
    const bool inpatch = fin && inzone && ring >= 0;
    uncertain = inpatch && !sure;
    return !fin ? P + 2 : (!inzone ? P + 1 : ((ring < 0 || !sure) ? P : ring * S + k));
}

// the literal evaluation (libm-model angle), used for the few uncertain points and by the tests
RPW_HD int classify_point_exact(float x, float y, float z, const PolarTables& t, const float* __restrict__ bounds)
{
    const int S = t.num_sectors;
    const int P = 8 * S;
    if (!(finite_(x) && finite_(y) && finite_(z))) return P + 2;
    const float d = sqrt_rn(x * x + y * y);
    if (!(d <= t.radius)) return P + 1;
    int ring = -1;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (d >= t.edges[i] && d < t.edges[i + 1]) ring = i;
    if (ring < 0) return P;
    const int k = sector_exact(polar_angle(x, y), S, t.inv_sector, bounds);
    return k < 0 ? P : ring * S + k;
}

} // namespace rpw
