"""Binning-pass probe: C5 dense frames (1.09 M points each), B distinct frames per launch so that the working
set exceeds the 256 MiB Infinity Cache. Reports per-kernel ms and the algorithmic GB/s of the binning pass
(28 B/pt: 12 in + 12 sorted SoA + 4 index; the counting sort reads the input twice: 40 B/pt actually moved)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rpw_amd
B = int(os.environ.get("B", "32")); reps = int(os.environ.get("REPS", "5"))
base = rpw_amd.workloads.c5_dense(0)
frames = []
for i in range(B):
    a = np.float32(0.05 * i); c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
    r = np.empty_like(base); r[:, 0] = base[:, 0] * c - base[:, 1] * s; r[:, 1] = base[:, 0] * s + base[:, 1] * c; r[:, 2] = base[:, 2]
    frames.append(r)
off = np.zeros(B + 1, np.uint32); off[1:] = np.cumsum([len(f) for f in frames]); N = int(off[-1])
eng = rpw_amd.Engine(rpw_amd.PatchworkConfig(max_split_depth=6, max_iter=int(os.environ.get("MAX_ITER", "100"))), max_points=N, max_frames=B)
d = torch.from_numpy(np.concatenate(frames)).cuda(); lab = torch.empty(N, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
eng.segment_batch_device(d.data_ptr(), off, lab.data_ptr(), None, st); torch.cuda.synchronize()
eng.set_profiling(True); eng.kernel_ms(reset=True)
for _ in range(reps): eng.segment_batch_device(d.data_ptr(), off, lab.data_ptr(), None, st)
torch.cuda.synchronize()
# wall time of the binning pass alone: whole call minus the patch_fit kernel (MAX_ITER=1 keeps it short), events on the launch stream
eng.set_profiling(False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): eng.segment_batch_device(d.data_ptr(), off, lab.data_ptr(), None, st)
e1.record(); torch.cuda.synchronize()
call_ms = e0.elapsed_time(e1) / reps
k = {n: v[0] / max(1, v[1]) for n, v in eng.kernel_ms().items()}
binms = k["bin_count"] + k["scan_offsets"] + k["bin_scatter"]
print(json.dumps({"call_ms": round(call_ms, 4), "call_minus_fit_ms": round(call_ms - k["patch_fit"], 4),
                  "binning_wall_GBps_alg28": round(28 * N / max(1e-9, call_ms - k["patch_fit"]) / 1e6, 1), "frames": B, "points": N, "input_MB": round(N * 12 / 1e6, 1), "ms": {n: round(v, 4) for n, v in k.items()},
                  "binning_ms": round(binms, 4), "binning_GBps_alg28": round(28 * N / binms / 1e6, 1),
                  "frac_of_8TBps": round(28 * N / binms / 1e6 / 8000, 4),
                  "frac_of_8TBps_actual40": round(40 * N / binms / 1e6 / 8000, 4)}))
