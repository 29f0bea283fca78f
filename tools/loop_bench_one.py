"""one-wave and saturated cost of selected loops (experiment helper): python tools/loop_bench_one.py what[,what..] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpw_amd
whats = [int(x) for x in sys.argv[1].split(",")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
eng = rpw_amd.Engine(rpw_amd.PatchworkConfig(), max_points=1024, max_frames=1)
for blocks in (1, 4096):
    for what in whats:
        ms = eng.loop_bench(what, n, 56, 20, blocks)
        elems = n if what in (3, 16, 17, 18) else n * 56 // 100
        print(os.environ.get("RPW_LIB", "default"), "blocks", blocks, "what", what, "%.3f ms  %.2f cycles/element/wave" % (ms, ms * 1e6 / (20 * elems) * 2.4))
