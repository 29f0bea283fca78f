"""Diagnostic: per-phase cycles of block 0 (largest patch). Needs RPW_LIB=tools/exp/stamps.so.
The stamps overwrite the head of the trace buffer, so trace is enabled but emit is disabled (trace=1 with tiny cap)."""
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
import ctypes
# copy the whole trace buffer (stamps live in its last 8 records)
eng._L.rpw_set_trace  # noqa
lib = eng._L
class H(ctypes.Structure): pass
st_ = lib.rpw_get_nodes(eng._h, out.ctypes.data, 4096, C.byref(cnt))
# rpw_get_nodes copies only `count` records; read the tail through a second call with a fake count is not possible,
# so the probe build reports count = cap (see RPW_STAMPS in rpw_capi.hip)
raw = out.view(np.uint8)[(4096 - 8) * 68:(4096 - 8) * 68 + 128].view(np.uint64).reshape(2, 8)
names = ["kernel", "loop", "barrier_wait", "chunks", "passes", "prologue", "epilogue", "n"]
for w, nm in ((0, "consumer"), (1, "producer")):
    print(nm, {k: int(v) for k, v in zip(names, raw[w])})
c = raw[0]
print("consumer cycles/chunk in loop: %.1f  (barrier wait %.1f);  loop share of kernel %.2f; pro+epi per pass %.0f" % (c[1] / c[3], c[2] / c[3], c[1] / c[0], (c[5] + c[6]) / c[4]))
