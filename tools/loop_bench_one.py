"""One loop of tools/loop_bench.py, for a profiler run:  python tools/loop_bench_one.py <what> <nodes> [n]   (what + 64: chain-server form, + 96: a server per node)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpw_amd
what = int(sys.argv[1]); nodes = int(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
eng = rpw_amd.Engine(rpw_amd.PatchworkConfig(), max_points=1024, max_frames=1)
ms = eng.loop_bench(what, n, 56, 20, nodes)
print("what", what, "nodes", nodes, "n", n, "%.3f ms" % ms, "%.1f G pts/s" % (nodes * 20 * n / ms / 1e6))
