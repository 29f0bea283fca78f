"""Single-frame latency probe (device-resident), optional RPW_LIB override; prints kernel ms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rpw_amd
frame = rpw_amd.workloads.c2_hdl64(int(os.environ.get("SEED", "0")))
eng = rpw_amd.Engine(rpw_amd.PatchworkConfig(max_split_depth=3), max_points=len(frame), max_frames=1)
d = torch.from_numpy(frame).cuda(); lab = torch.empty(len(frame), dtype=torch.uint8, device="cuda")
off = np.array([0, len(frame)], np.uint32); st = torch.cuda.current_stream().cuda_stream
for _ in range(3): eng.segment_batch_device(d.data_ptr(), off, lab.data_ptr(), None, st)
torch.cuda.synchronize(); eng.set_profiling(True); eng.kernel_ms(reset=True)
for _ in range(10): eng.segment_batch_device(d.data_ptr(), off, lab.data_ptr(), None, st)
torch.cuda.synchronize()
print(os.environ.get("RPW_LIB", "default"), {k: round(v[0] / max(1, v[1]), 4) for k, v in eng.kernel_ms().items()})
