"""Diagnostic: per-phase cycles of block 0 (the largest patch) of the single-wave patch_fit kernel.
Needs the diagnostic build:  hipcc ... -DRPW_STAMPS -shared -o tools/exp/stamps.so csrc/rpw_kernels.hip csrc/rpw_capi.hip
and RPW_LIB=tools/exp/stamps.so. The stamps sit in the last 8 records of the trace buffer (that build returns the whole buffer)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rpw_amd, ctypes as C
frame = rpw_amd.workloads.c2_hdl64(int(os.environ.get("SEED", "0")))
eng = rpw_amd.Engine(rpw_amd.PatchworkConfig(max_split_depth=3), max_points=len(frame), max_frames=1)
eng.set_trace(True, 4096)
d = torch.from_numpy(frame).cuda(); lab = torch.empty(len(frame), dtype=torch.uint8, device="cuda")
off = np.array([0, len(frame)], np.uint32); st = torch.cuda.current_stream().cuda_stream
for _ in range(3): eng.segment_batch_device(d.data_ptr(), off, lab.data_ptr(), None, st)
torch.cuda.synchronize()
out = np.zeros(4096, rpw_amd.NODE_DTYPE); cnt = C.c_uint32(0)
eng._L.rpw_get_nodes(eng._h, out.ctypes.data, 4096, C.byref(cnt))
raw = out.view(np.uint8)[(4096 - 8) * 68:(4096 - 8) * 68 + 128].view(np.uint64)
names = ["gather", "sweep_range", "mask_pass", "sweep_xyz", "sweep_cov", "eigen", "sweep_dist", "kernel"]
cyc = {k: int(v) for k, v in zip(names, raw[:8])}
work = {k: int(v) for k, v in zip(names, raw[8:16])}
print("seed", os.environ.get("SEED", "0"), "patch points", work["kernel"], "kernel cycles", cyc["kernel"])
acc = 0
for k in names[:-1]:
    per = cyc[k] / max(1, work[k])
    acc += cyc[k]
    print(f"  {k:12s} {cyc[k]:10d} cycles ({100.0 * cyc[k] / cyc['kernel']:5.1f} %)  work {work[k]:9d}  -> {per:8.2f} cycles per unit")
print(f"  other        {cyc['kernel'] - acc:10d} cycles ({100.0 * (cyc['kernel'] - acc) / cyc['kernel']:5.1f} %)")
