"""K4 capacity probe: B copies of ONE frame in one launch (no long straggler relative to the total work)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rpw_amd
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_lib as O
seed = int(os.environ.get("SEED", "0"))
f = rpw_amd.workloads.c2_hdl64(seed)
r = O.segment(f, O.make_config(max_split_depth=3), want_residual=False)
w = (r.nodes["len"].astype(np.int64) * np.maximum(r.nodes["iterations"], 1))
print("frame work pt-iters", int(w.sum()), "max patch", int(w.max()), "iterations", int(r.stats["total_iterations"]))
for B in (1, 64, 256, 512):
    off = (np.arange(B + 1, dtype=np.uint64) * len(f)).astype(np.uint32)
    eng = rpw_amd.Engine(rpw_amd.PatchworkConfig(max_split_depth=3), max_points=B * len(f), max_frames=B)
    d = torch.from_numpy(np.tile(f, (B, 1))).cuda(); lab = torch.empty(B * len(f), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    eng.segment_batch_device(d.data_ptr(), off, lab.data_ptr(), None, st); torch.cuda.synchronize()
    eng.set_profiling(True); eng.kernel_ms(reset=True)
    for _ in range(3): eng.segment_batch_device(d.data_ptr(), off, lab.data_ptr(), None, st)
    torch.cuda.synchronize()
    k = {n: v[0] / max(1, v[1]) for n, v in eng.kernel_ms().items()}
    print(B, "K4 ms %.3f  -> %.2f us/frame; chain-floor %.2f us/frame (8.2 cyc/add, 2 sweeps, 1024 WGs, 2.4 GHz)" % (k["patch_fit"], 1e3 * k["patch_fit"] / B, w.sum() * 2 * 8.2 / 2.4e3 / 1024))
    eng.close(); del d, lab
