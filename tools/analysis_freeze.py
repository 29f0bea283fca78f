"""Analysis aid (CPU, numpy): how much of the refit loop's work lies on slots that provably cannot become inliers?

For every patch of a few bench frames the refit loop of the reference (recursive_patchwork.cpp:186-217) is replayed with
tests/np_emulation.py and, for a reference plane taken after k0 refits and a guard band G, it reports
  * the share of the node's slots farther than thr + G from the reference plane ("frozen"), and
  * for how many of the following refits the plane stays within the band:
        rho * min(|n_k - n_0|, |n_k + n_0|) + |c_k - c_0| < G,   rho = max |p - c_0| over the node,
    which guarantees | |(p - c_k).n_k| - |(p - c_0).n_0| | < G for every slot of the node.
Weighted by slots x refits it says what an active-set compaction of patch_fit would save.
Usage: python tools/analysis_freeze.py [workload] [n_frames]
       python tools/analysis_freeze.py --speculative [workload] [n_frames]
The second mode drops the provable bound: a candidate set chosen after R0 refits (slots within thr + G of that plane), the
rest checked afterwards against every later plane in one pass (exact, with a redo when a left-out slot turns out to have
been an inlier). It reports the slot-refits saved, the patches that would have to be redone, and the ceiling: slots that are
an inlier of NO refit (c2: 21.6 % of the slot-refits; the realistic selections save 1-4 % and fail on 10-60 % of the work).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib

import np_emulation as E

wl = importlib.import_module("ros2-recursive-patchwork-implementation_amd.workloads")
F = np.float32


def patches(xyz, S=10, R=150.0):
    xyz = xyz[np.isfinite(xyz).all(axis=1)]
    d = np.sqrt(xyz[:, 0] * xyz[:, 0] + xyz[:, 1] * xyz[:, 1], dtype=F)
    inz = d <= F(R)
    Z, dz = xyz[inz], d[inz]
    edges = np.array([F(R) ** (F(i) / F(8)) for i in range(9)], F)
    ang = np.arctan2(Z[:, 1], Z[:, 0]).astype(F)
    ang = np.where(ang < 0, ang + F(2 * np.pi), ang)
    ring = np.searchsorted(edges, dz, side="right") - 1
    sec = np.floor(ang / F(2 * np.pi / S)).astype(int)
    ok = (ring >= 0) & (ring < 8) & (sec < S)
    pid = ring * S + sec
    for p in np.unique(pid[ok]):
        m = ok & (pid == p)
        yield int(p), Z[m], dz[m]


def replay(P, dz, cfg):
    n = len(P)
    mean = E.seqsum(dz) / F(n)
    rel = mean / F(cfg["filtering_radius"])
    z_th = F(cfg["sensor_height"]) + F(0.2) * rel
    g = P[:, 2] < z_th
    if g.sum() < 3:
        return None
    thr = F(cfg["th_dist"]) * (F(1) + F(0.2) * rel)
    planes, counts = [], []
    for _ in range(cfg["max_iter"]):
        if g.sum() < 3:
            break
        c, nrm, _ = E.fit_plane(P[g])
        planes.append((c, nrm)); counts.append(int(g.sum()))
        new = E.plane_dist(P, c, nrm) < thr
        if np.array_equal(new, g):
            break
        g = new
    return thr, planes, counts


def speculative(name, nframes):
    cfg = dict(sensor_height=1.2, filtering_radius=150.0, th_dist=0.2, max_iter=100)
    Gs, R0s = [0.2, 0.4, 0.8, 1.6], [1, 2, 4]
    tot = never_w = 0.0
    sav = {(r0, G): 0.0 for r0 in R0s for G in Gs}
    fail = {k: 0 for k in sav}
    failw = {k: 0.0 for k in sav}
    for seed in range(nframes):
        for pid, P, dz in patches(wl.make(name, seed)):
            r = replay(P, dz, cfg)
            if r is None:
                continue
            thr, planes, _ = r
            n, Rf = len(P), len(planes)
            tot += n * Rf
            if Rf < 3:
                continue
            D = np.stack([E.plane_dist(P, c, nr) for c, nr in planes])      # [refit][slot]
            inl = D < thr
            never_w += (~inl.any(axis=0)).sum() * Rf
            for r0 in R0s:
                if r0 >= Rf - 1:
                    continue
                later = inl[r0:].any(axis=0)
                for G in Gs:
                    left_out = D[r0] >= thr + F(G)
                    if (left_out & later).any():
                        fail[(r0, G)] += 1; failw[(r0, G)] += n * Rf
                    else:
                        sav[(r0, G)] += left_out.sum() * (Rf - r0)
    print(f"{name}: slot-refits {tot:.3g}; on slots that are an inlier of no refit: {never_w / tot:.1%}")
    for k in sav:
        print(f"R0={k[0]} G={k[1]}: saved {sav[k] / tot:.1%} of the slot-refits; patches to redo {fail[k]} ({failw[k] / tot:.1%} of the work)")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--speculative":
        return speculative(sys.argv[2] if len(sys.argv) > 2 else "c2", int(sys.argv[3]) if len(sys.argv) > 3 else 2)
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    cfg = dict(sensor_height=1.2, filtering_radius=150.0, th_dist=0.2, max_iter=100)
    Gs = [0.1, 0.2, 0.4]
    K0s = [1, 2, 3]
    tot = 0.0
    tot_big = 0.0
    saved = {(k0, G): 0.0 for k0 in K0s for G in Gs}
    rows = []
    for seed in range(nframes):
        xyz = wl.make(name, seed)
        for pid, P, dz in patches(xyz):
            n = len(P)
            r = replay(P, dz, cfg)
            if r is None:
                continue
            thr, planes, counts = r
            Rf = len(planes)
            work = n * Rf
            tot += work
            if n < 1024 or Rf < 6:
                continue
            tot_big += work
            line = [seed, pid, n, Rf, counts[-1] / n]
            for k0 in K0s:
                c0, n0 = planes[k0]
                rho = float(np.sqrt(((P - c0) ** 2).sum(axis=1)).max())
                d0 = E.plane_dist(P, c0, n0)
                for G in Gs:
                    frozen = float((d0 >= thr + F(G)).mean())
                    held = 0
                    for k in range(k0 + 1, Rf):
                        ck, nk = planes[k]
                        dn = min(np.linalg.norm(nk - n0), np.linalg.norm(nk + n0))
                        if rho * dn + np.linalg.norm(ck - c0) < G:
                            held += 1
                        else:
                            break
                    saved[(k0, G)] += n * frozen * held
                    line += [frozen, held]
            rows.append(line)
    print("node: seed patch n refits inl | per k0 in", K0s, "per G in", Gs, ": frozen share, refits the band held")
    for r in sorted(rows, key=lambda r: -r[2] * r[3])[:40]:
        print(" ".join(f"{v:.2f}" if isinstance(v, float) else str(v) for v in r))
    print(f"slot-refits total {tot:.3g}, in nodes n>=1024 & refits>=6: {tot_big:.3g} ({tot_big / tot:.1%})")
    for k, v in saved.items():
        print(f"k0={k[0]} G={k[1]}: slot-refits on frozen slots while the band holds: {v / tot:.1%} of all")


if __name__ == "__main__":
    main()
