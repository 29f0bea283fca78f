"""Which half bounds a lone patch's fused pass / covariance sweep -- the feeding half (time per SLOT, whatever the share of
inliers) or the chain (time per INLIER)? Runs patch_fit's loops on one synthetic node with 5 ... 100 % inliers.
  python tools/loop_pct_probe.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpw_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
eng = rpw_amd.Engine(rpw_amd.PatchworkConfig(), max_points=1024, max_frames=1)
FORM = {1: "one wave", 3: "chain server (lone feeder)", 4: "one server per node"}
for waves in (1, 3, 4):
    for what, nm in ((17, "fused pass (plane)"), (1, "covariance sweep")):
        row = []
        for pct in (5, 20, 40, 56, 80, 100):
            reps = 20
            ms = eng.loop_bench(what + 32 * (waves - 1), n, pct, reps, 1)
            m = max(1, n * pct // 100)
            per_slot = ms * 1e6 / (reps * n) * 2.4
            per_inl = ms * 1e6 / (reps * m) * 2.4
            row.append(f"{pct:3d}%: {per_slot:5.2f}/slot {per_inl:6.2f}/inlier")
        print(f"{FORM[waves]:28s} {nm:20s} " + " | ".join(row))
eng.close()
