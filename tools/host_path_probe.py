"""Latency of the drop-in entry points on one C2 frame (host pointers in, host results out): the C-ABI call
rpw_segment (labels only / with residuals) and the Python mirror of RecursivePatchwork.filterGroundPoints."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rpw_amd
pts = rpw_amd.workloads.c2_hdl64(0)
eng = rpw_amd.Engine(rpw_amd.PatchworkConfig(max_split_depth=3), max_points=len(pts), max_frames=1)
def t(fn, reps=50):
    for _ in range(5): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return 1e3 * (time.perf_counter() - t0) / reps
print("rpw_segment, labels only      : %.3f ms" % t(lambda: eng.segment(pts)))
print("rpw_segment, labels+residuals : %.3f ms" % t(lambda: eng.segment(pts, want_residual=True)))
rp = rpw_amd.RecursivePatchwork(rpw_amd.PatchworkConfig(max_split_depth=3), max_points=len(pts))
print("RecursivePatchwork.filterGroundPoints (python mirror): %.3f ms" % t(lambda: rp.filterGroundPoints(pts)))
