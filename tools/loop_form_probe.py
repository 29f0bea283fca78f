"""Cycles per element (shader clock, s_memtime) of patch_fit's loops on ONE synthetic node, every form of the kernel:
what a lone patch pays -- the figure behind the latency of a single frame.   python tools/loop_form_probe.py [n] [inlier %]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpw_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
pct = int(sys.argv[2]) if len(sys.argv) > 2 else 56
eng = rpw_amd.Engine(rpw_amd.PatchworkConfig(), max_points=1024, max_frames=1)
names = {1: "cov", 0: "xyz", 2: "dist", 3: "range", 17: "fused plane", 16: "fused seed"}
FORM = {1: "one wave", 3: "server of 7 (lone feeder)", 4: "server per patch"}
for waves in (1, 3, 4):
    row = []
    for what, nm in names.items():
        reps = 20
        ms, ticks = eng.loop_bench_ticks(what + 32 * (waves - 1), n, pct, reps, 1)
        elems = n if what in (3, 16, 17) else n * pct // 100
        row.append(f"{nm} {ticks / (reps * elems):5.2f}")
    print(f"{FORM[waves]:26s} cycles per element: " + " | ".join(row), flush=True)
eng.close()
