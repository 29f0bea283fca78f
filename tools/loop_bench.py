"""Cycles per element of the loops patch_fit1 lives in (chain sweeps, mask passes), alone on the chip and at full occupancy.
  python tools/loop_bench.py [n]      (assumes 2.4 GHz for the cycle figures; ns are exact)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpw_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
eng = rpw_amd.Engine(rpw_amd.PatchworkConfig(), max_points=1024, max_frames=1)
names = {0: "sweep xyz (3 rows)", 1: "sweep cov (6 rows)", 2: "sweep |d| (1 row)", 3: "sweep range (1 row, all slots)",
         17: "fused pass plane (mask + centroid chain)", 16: "fused pass seed (mask + centroid chain)"}
pct = 56
FORM = {1: "one wave per node", 3: "chain-server form (7 nodes per workgroup of 8 waves)", 4: "one server wave per node"}
for blocks, waves in ((1, 1), (1, 4), (1, 3), (7, 3), (256 * 4, 1), (256 * 4, 4), (256 * 7, 3), (256 * 16, 1), (256 * 8, 4), (256 * 14, 3), (256 * 28, 3)):
    print(f"--- {blocks} nodes ({blocks / 256:.1f} per CU), {FORM[waves]}, node of {n} slots, {pct} % inliers")
    for what, nm in names.items():
        reps = 20
        ms = eng.loop_bench(what + 32 * (waves - 1), n, pct, reps, blocks)
        elems = n if what in (3, 16, 17) else n * pct // 100
        ns = ms * 1e6 / (reps * elems)
        print(f"  {nm:32s} {ms:9.3f} ms  {ns:7.3f} ns/element/wave = {ns * 2.4:6.2f} cycles;  chip rate {blocks * reps * elems / ms / 1e6:9.1f} G elements/s")
eng.close()
