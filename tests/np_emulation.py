"""Independent numpy/float32 restatement of the reference path (test infrastructure).

Written separately from oracle/rpw_oracle.cpp, in another language, to catch coding slips in the
C++ oracle: same reference lines (recursive_patchwork.cpp:77-426, point_cloud_processor.cpp:58-86,
cuda_interface.cu:582-654) and the same Eigen 3.4.0 solver, but every sum is
np.add.accumulate over float32 (a sequential loop, like the reference's) and libm's atan2f is called
through ctypes. It pins the oracle's control flow, not the reference's bits ("parity unpinned").
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

F = np.float32
_libm = ctypes.CDLL("libm.so.6")
_libm.atan2f.restype = ctypes.c_float
_libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
_libm.powf.restype = ctypes.c_float
_libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
FLT_MIN = F(1.17549435e-38)
FLT_MAX = F(3.402823466e+38)


def seqsum(a):
    a = np.asarray(a, dtype=F)
    if a.size == 0:
        return F(0)
    return np.add.accumulate(a, dtype=F)[-1]


def _hypot(a, b):
    x, y = F(abs(a)), F(abs(b))
    p = x if x > y else y
    if p == 0:
        return F(0)
    q = (y if y < x else x) / p
    return p * np.sqrt(F(1) + q * q, dtype=F)


def _givens(p, q):
    if q == 0:
        return (F(-1) if p < 0 else F(1)), F(0)
    if p == 0:
        return F(0), (F(1) if q < 0 else F(-1))
    if abs(p) > abs(q):
        t = q / p
        u = np.sqrt(F(1) + t * t, dtype=F)
        if p < 0:
            u = -u
        c = F(1) / u
        return c, -t * c
    t = p / q
    u = np.sqrt(F(1) + t * t, dtype=F)
    if q < 0:
        u = -u
    s = F(-1) / u
    return -t * s, s


def eig3(A):
    """Eigen 3.4.0 SelfAdjointEigenSolver<Matrix3f>; returns (info, evals, Q) with Q[:, k] eigenvectors."""
    A = np.asarray(A, dtype=F)
    m = np.zeros((3, 3), F)
    for i in range(3):
        for j in range(i + 1):
            m[i, j] = A[i, j]
    scale = F(np.abs(m).max())
    if scale == 0:
        scale = F(1)
    for i in range(3):
        for j in range(i + 1):
            m[i, j] = m[i, j] / scale
    d = np.zeros(3, F); e = np.zeros(2, F)
    d[0] = m[0, 0]
    v1 = m[2, 0] * m[2, 0]
    if v1 <= FLT_MIN:
        d[1], d[2], e[0], e[1] = m[1, 1], m[2, 2], m[1, 0], m[2, 1]
        Q = np.eye(3, dtype=F)
    else:
        beta = np.sqrt(m[1, 0] * m[1, 0] + v1, dtype=F)
        inv = F(1) / beta
        m01, m02 = m[1, 0] * inv, m[2, 0] * inv
        q = F(2) * m01 * m[2, 1] + m02 * (m[2, 2] - m[1, 1])
        d[1] = m[1, 1] + m02 * q
        d[2] = m[2, 2] - m02 * q
        e[0] = beta
        e[1] = m[2, 1] - m01 * q
        Q = np.array([[1, 0, 0], [0, m01, m02], [0, m02, -m01]], dtype=F)
    end, start, it = 2, 0, 0
    pinv = F(1) / F(np.finfo(np.float32).eps)
    while end > 0:
        for i in range(start, end):
            if abs(e[i]) < FLT_MIN:
                e[i] = F(0)
            else:
                sc = pinv * e[i]
                if sc * sc <= (abs(d[i]) + abs(d[i + 1])):
                    e[i] = F(0)
        while end > 0 and e[end - 1] == 0:
            end -= 1
        if end <= 0:
            break
        it += 1
        if it > 90:
            break
        start = end - 1
        while start > 0 and e[start - 1] != 0:
            start -= 1
        td = (d[end - 1] - d[end]) * F(0.5)
        ee = e[end - 1]
        mu = d[end]
        if td == 0:
            mu = mu - F(abs(ee))
        elif ee != 0:
            e2 = ee * ee
            h = _hypot(td, ee)
            den = td + (h if td > 0 else -h)
            mu = mu - (ee / (den / ee) if e2 == 0 else e2 / den)
        x = d[start] - mu
        z = e[start]
        k = start
        while k < end and z != 0:
            c, s = _givens(x, z)
            sdk = s * d[k] + c * e[k]
            dkp1 = s * e[k] + c * d[k + 1]
            d[k] = c * (c * d[k] - s * e[k]) - s * (c * e[k] - s * d[k + 1])
            d[k + 1] = s * sdk + c * dkp1
            e[k] = c * sdk - s * dkp1
            if k > start:
                e[k - 1] = c * e[k - 1] - s * z
            x = e[k]
            if k < end - 1:
                z = -s * e[k + 1]
                e[k + 1] = c * e[k + 1]
            if not (c == 1 and s == 0):
                xi = Q[:, k].copy(); yi = Q[:, k + 1].copy()
                Q[:, k] = c * xi - s * yi
                Q[:, k + 1] = s * xi + c * yi
            k += 1
    info = 0 if it <= 90 else 1
    if info == 0:
        for i in range(2):
            k = int(np.argmin(d[i:]))          # first minimum
            if k > 0:
                d[i], d[i + k] = d[i + k], d[i]
                Q[:, [i, i + k]] = Q[:, [i + k, i]]
    return info, d * scale, Q


def plane_dist(P, c, n):
    e0 = (P[:, 0] - c[0]) * n[0]
    e1 = (P[:, 1] - c[1]) * n[1]
    e2 = (P[:, 2] - c[2]) * n[2]
    return np.abs(e0 + (e1 + e2))


def fit_plane(P):
    if len(P) < 3:
        return np.zeros(3, F), np.array([0, 0, 1], F), FLT_MAX
    n = F(len(P))
    c = np.array([seqsum(P[:, k]) / n for k in range(3)], F)
    D = (P - c).astype(F)
    cov = np.zeros((3, 3), F)
    for i in range(3):
        for j in range(3):
            cov[i, j] = seqsum(D[:, i] * D[:, j]) / F(len(P) - 1)
    _, _, Q = eig3(cov)
    nrm = Q[:, 0].copy()
    if nrm[2] < 0:
        nrm = -nrm
    res = seqsum(plane_dist(P, c, nrm)) / n
    return c, nrm, res


def three_lowest(z):
    """libstdc++ partial_sort(idx, idx+3, idx+n, z[a] < z[b]) as a set: __heap_select on a 3-heap."""
    h = [0, 1, 2]

    def adjust(v):
        hole = 1 if z[h[2]] < z[h[1]] else 2
        h[0] = h[hole]
        if z[h[0]] < z[v]:
            h[hole] = h[0]; h[0] = v
        else:
            h[hole] = v
    adjust(h[0])
    for i in range(3, len(z)):
        if z[i] < z[h[0]]:
            adjust(i)
    return h


def fit_and_split(P, mean_dist, depth, cfg, stats):
    n = len(P)
    stats["nodes"] += 1
    if n < 3 or depth > cfg["max_split_depth"]:
        return np.zeros(n, bool)
    area = (P[:, 0].max() - P[:, 0].min()) * (P[:, 1].max() - P[:, 1].min())
    if area < F(25) and depth > 0:
        return np.ones(n, bool)
    if (P[:, 2].max() - P[:, 2].min()) < F(0.05) and n > 10:
        return np.ones(n, bool)
    rel = mean_dist / F(cfg["filtering_radius"])
    if cfg["adaptive_seed_height"]:
        z_th = F(cfg["sensor_height"]) + F(0.2) * rel
    else:
        z_th = np.sort(P[:, 2])[int(F(0.1) * F(n))] + F(cfg["th_seeds"])
    g = P[:, 2] < z_th
    if g.sum() < 3:
        g = np.zeros(n, bool); g[three_lowest(P[:, 2])] = True
    thr = F(cfg["th_dist"]) * (F(1) + F(0.2) * rel)
    for _ in range(cfg["max_iter"]):
        if g.sum() < 3:
            break
        c, nrm, _ = fit_plane(P[g])
        stats["iters"] += 1
        new = plane_dist(P, c, nrm) < thr
        if np.array_equal(new, g):
            break
        g = new
    _, _, res = fit_plane(P[g])
    split_thr = F(cfg["th_dist"]) * (F(1) + F(1.5) * F(depth))
    if res > split_thr and depth < cfg["max_split_depth"] and n >= 50 + 10 * depth:
        cx = seqsum(P[:, 0]) / F(n); cy = seqsum(P[:, 1]) / F(n)
        vx = seqsum((P[:, 0] - cx) * (P[:, 0] - cx)) / F(n)
        vy = seqsum((P[:, 1] - cy) * (P[:, 1] - cy)) / F(n)
        vals = P[:, 0] if vx > vy else P[:, 1]
        med = np.sort(vals)[n // 2]
        left = vals <= med
        stats["splits"] += 1
        rl = fit_and_split(P[left], mean_dist, depth + 1, cfg, stats)
        rr = fit_and_split(P[~left], mean_dist, depth + 1, cfg, stats)
        return np.concatenate([rl, rr])          # positional, as recursive_patchwork.cpp:290-304
    return g


def segment(xyz, cfg):
    """-> (labels uint8[n], stats) with the oracle's label values."""
    xyz = np.asarray(xyz, F).reshape(-1, 3)
    labels = np.zeros(len(xyz), np.uint8)
    fin = np.isfinite(xyz).all(axis=1)
    labels[~fin] = 2
    idx_c = np.nonzero(fin)[0]
    C = xyz[idx_c]
    d = np.sqrt(C[:, 0] * C[:, 0] + C[:, 1] * C[:, 1], dtype=F)
    R = F(cfg["filtering_radius"])
    inz = d <= R
    labels[idx_c[~inz]] = 3
    stats = {"nodes": 0, "iters": 0, "splits": 0}
    if inz.sum() < 3:
        return labels, stats
    Z, dz, idx_z = C[inz], d[inz], idx_c[inz]
    S = cfg["num_sectors"]
    edges = [F(1.0) * F(_libm.powf(float(R / F(1.0)), float(F(i) / F(8)))) for i in range(9)]
    sa = F(2.0 * math.pi / S) if False else F(np.float64(np.float32(2.0)) * math.pi / S)
    ang = np.array([_libm.atan2f(float(y), float(x)) for x, y in zip(Z[:, 0], Z[:, 1])], F)
    neg = ang < 0
    ang[neg] = (ang[neg].astype(np.float64) + 2.0 * math.pi).astype(F)
    ground = np.zeros(len(Z), bool)
    for ring in range(8):
        for sec in range(S):
            a0, a1 = F(sec) * sa, F(sec + 1) * sa
            m = (dz >= edges[ring]) & (dz < edges[ring + 1]) & (ang >= a0) & (ang < a1)
            if not m.any():
                continue
            pi = np.nonzero(m)[0]
            mean = seqsum(dz[pi]) / F(len(pi))
            mask = fit_and_split(Z[pi], mean, 0, cfg, stats)
            ground[pi[mask]] = True
    labels[idx_z[ground]] = 1
    return labels, stats
