"""Throughput and latency of the drop-in (host pointers in, host results out) entry points on C2 frames:
the synchronous rpw_segment, and the rpw_submit / rpw_wait pipeline with K frames in flight (K slots, frames
submitted round-robin; a slot is waited right before it is reused, so at most K frames are in flight).

  python tools/host_path_probe.py [--slots 1,2,4,8,16] [--seconds 1.5] [--frames 8]
Prints one JSON object.
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rpw_amd


def pipeline_fps(eng, frames, K, seconds, labels_out):
    """frames/s with K slots in flight, host memory in -> labels out; also returns per-frame latency stats."""
    eng.set_pipeline(K)
    n = len(frames)
    t_sub = [0.0] * K
    lat = []
    for s in range(K):                                   # warm-up round: every slot once
        eng.submit(s, frames[s % n])
    for s in range(K):
        eng.wait(s, labels_out[s])
    i = 0
    pending = [False] * K
    t0 = time.perf_counter()
    while True:
        s = i % K
        if pending[s]:
            eng.wait(s, labels_out[s])
            lat.append(time.perf_counter() - t_sub[s])
        now = time.perf_counter()
        if now - t0 >= seconds:
            break
        t_sub[s] = now
        eng.submit(s, frames[i % n])
        pending[s] = True
        i += 1
    done = len(lat)
    for k in range(1, K):                                # drain the other slots (counted: they were submitted in time)
        s = (i + k) % K
        if pending[s] and s != i % K:
            eng.wait(s, labels_out[s]); done += 1
    dt = time.perf_counter() - t0
    lat_ms = np.array(lat) * 1e3
    return {"slots": K, "frames_per_sec": round(done / dt, 1), "frames": done,
            "latency_ms_median": round(float(np.median(lat_ms)), 3) if len(lat_ms) else None,
            "latency_ms_p95": round(float(np.percentile(lat_ms, 95)), 3) if len(lat_ms) else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", default="1,2,4,8,16")
    ap.add_argument("--seconds", type=float, default=1.5)
    ap.add_argument("--frames", type=int, default=8)
    args = ap.parse_args()
    frames = [rpw_amd.workloads.c2_hdl64(s) for s in range(args.frames)]
    cap = max(len(f) for f in frames)
    eng = rpw_amd.Engine(rpw_amd.PatchworkConfig(max_split_depth=3), max_points=cap, max_frames=1)
    out = {"workload": "c2_hdl64 seeds 0..%d, host memory in -> labels in host memory out" % (args.frames - 1),
           "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "(library default 16)")}
    for _ in range(5):
        eng.segment(frames[0])
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < args.seconds:
        eng.segment(frames[k % len(frames)]); k += 1
    dt = time.perf_counter() - t0
    out["rpw_segment_sync"] = {"frames_per_sec": round(k / dt, 1), "ms_per_frame": round(1e3 * dt / k, 3)}
    Kmax = max(int(x) for x in args.slots.split(","))
    labels_out = [np.empty(cap, np.uint8) for _ in range(Kmax)]
    out["pipeline"] = [pipeline_fps(eng, frames, int(K), args.seconds, labels_out) for K in args.slots.split(",")]
    eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
