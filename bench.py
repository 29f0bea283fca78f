#!/usr/bin/env python3
"""bench.py -- frames/sec of the Recursive Patchwork hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (binning: count, scan, scatter into patch-sorted arrays -> recursive plane fit ->
labels) over one batch of 2048 synthetic HDL-64-like frames that is already resident in HBM (BASELINE.json
configs[1]/[2]: ~126 k points per frame, max_split_depth = 3). Frames are independent, so N GPUs run N such batches with no data-path collective ("weak"
scaling); the only RCCL call is the gather of every batch's per-frame result record.

`value` is that device-resident throughput. The same JSON line also carries what the DROP-IN path does, host memory in,
labels in host memory out (`host_path`: the synchronous one-frame call and the rpw_submit / rpw_wait pipeline with 8
frames in flight = BASELINE.json configs[2]), the latency of one frame, the recursion the workload performs, the two
rooflines and the CPU baseline.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

# Batches in flight run on separate HIP streams; ROCm multiplexes streams onto 4 hardware queues by
# default, which would serialise streams that share one. Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_POINT_FIT = 17        # 12 sorted xyz + 4 index + 1 label
BYTES_PER_POINT_BIN = 28        # 12 xyz in + 12 sorted SoA out + 4 index out
BYTES_PER_POINT_BIN_ACTUAL = 40 # what the counting sort moves: the input is read twice (count pass, scatter pass)
PROFILE_ROUND = "r03"


WORKLOADS = {   # name: (generator, config, default frames per step, description) -- BASELINE.json `configs`
    "c2": ("c2_hdl64", dict(max_split_depth=3), 2048,
           "c2_hdl64 (BASELINE.json configs[1]/[2]): seeded HDL-64-like ray-cast frames"),
    "c1": ("c1_test_cloud", dict(), 8192, "c1_test_cloud (configs[0]): the reference test's 20 k-point clouds"),
    "c4": ("c4_iac_fused", dict(max_split_depth=1000), 1024,
           "c4_iac_fused (configs[3]): 3 fused solid-state sensors, ~296 k points, deep recursion allowed"),
    "c5": ("c5_dense", dict(max_split_depth=6), 256, "c5_dense (configs[4]): dense 1.09 M-point frames"),
    "rough3": ("rough_terrain", dict(max_split_depth=3, th_dist=0.0005), 1024,
               "rough_terrain: HDL-64-like scan + far-field clutter, th_dist 0.0005: fits split NATURALLY down to depth 3"),
    "rough6": ("rough_terrain", dict(max_split_depth=6, th_dist=0.0005), 1024,
               "rough_terrain: the same scan, max_split_depth 6: natural splits down to depth 6"),
}


def make_frames(n_frames: int, rank: int, workload: str = "c2"):
    """8 seeded base frames of the workload (C2/C3 by default), further frames are yaw-rotated copies."""
    import rpw_amd
    W = rpw_amd.workloads
    gen = getattr(W, WORKLOADS[workload][0])
    base = [gen(seed) for seed in range(min(8 if workload != "c5" else 4, n_frames))]
    frames = []
    for i in range(n_frames):
        b = base[i % len(base)]
        k = i // len(base) + 8 * rank
        if k == 0:
            frames.append(b)
        else:
            a = np.float32(0.173 * k)
            c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
            r = np.empty_like(b)
            r[:, 0] = b[:, 0] * c - b[:, 1] * s
            r[:, 1] = b[:, 0] * s + b[:, 1] * c
            r[:, 2] = b[:, 2]
            frames.append(r)
    return frames


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(frames, ocfg_kw, seconds: float = 8.0):
    """The CPU oracle (a port of the reference algorithm) timed on this box's host cores by its own native harness
    (oracle/rpw_oracle.cpp rpwo_bench_threads: no Python in the timed loops): single-thread median ms/frame, then
    one frame per std::thread on all hardware threads. Silent, as the reference is NOT: its as-shipped stdout chatter
    (about 5 flushed lines per refit iteration, cuda_interface.cu:637-651) is timed separately, into /dev/null."""
    import oracle_lib as O
    cfg = O.make_config(**ocfg_kw)
    sample = frames[: min(len(frames), 64)]
    # all hardware threads, and one thread per two of them (physical cores on an SMT-2 host): the better one is reported
    r_all = O.bench_threads(sample, cfg, n_threads=0, seconds=seconds / 2, single_reps=24)
    hw = int(r_all["hardware_concurrency"]) or 1
    r_half = O.bench_threads(sample, cfg, n_threads=max(1, hw // 2), seconds=seconds / 2, single_reps=4) if hw > 1 else r_all
    fps_all, fps_half = r_all["frames_done"] / r_all["elapsed_s"], r_half["frames_done"] / r_half["elapsed_s"]
    r = r_all if fps_all >= fps_half else r_half
    r = dict(r, single_thread_median_ms=r_all["single_thread_median_ms"], single_thread_frames=r_all["single_thread_frames"])
    # as shipped: the same single-thread run with the reference's prints, stdout -> /dev/null
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    times = []
    try:
        os.dup2(devnull, 1)
        O.set_chatter(True)
        for i in range(10):
            t0 = time.perf_counter()
            O.segment(sample[i % len(sample)], cfg, want_residual=False, want_nodes=False)
            times.append(time.perf_counter() - t0)
    finally:
        O.set_chatter(False)
        os.dup2(saved, 1)
        os.close(saved); os.close(devnull)
    fps = r["frames_done"] / r["elapsed_s"]
    return {
        "value": round(fps, 2), "unit": "frames/s", "cores": int(r["threads"]), "kind": "port",
        "cpu_model": cpu_model(), "hardware_concurrency": int(r["hardware_concurrency"]),
        "sample": f"{r['frames_done']} frames of the bench batch ({len(sample)} distinct), one frame per std::thread on "
                  f"{r['threads']} threads, {r['elapsed_s']:.1f} s, native harness; single thread: median of "
                  f"{r['single_thread_frames']} frames after 3 warm-ups",
        "frames_per_sec_by_threads": {str(int(r_all["threads"])): round(fps_all, 2), str(int(r_half["threads"])): round(fps_half, 2)},
        "single_thread_ms_per_frame": round(r["single_thread_median_ms"], 3),
        "single_thread_ms_per_frame_as_shipped_chatter_to_devnull": round(1e3 * float(np.median(times)), 3),
    }


def load_pmc_traffic(kernel: str, frames_per_launch: int):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE, separate passes; tools/collect_traffic.sh). Only valid for the batch size it was taken at."""
    name = f"{PROFILE_ROUND}_pmc_traffic_bench.json"
    path = os.path.join(ROOT, "profiles", name)
    try:
        d = json.load(open(path))
        if d.get("_meta", {}).get("frames_per_launch", 256) != frames_per_launch:
            return None, None
        v = d.get(kernel)
        return (int(v["hbm_bytes_per_launch"]), "profiles/" + name) if v else (None, None)
    except Exception:
        return None, None


def gather_records(local_rec, world):
    """The path's only collective: all-gather of the finished batch's per-frame result record (int32 x 4 per frame:
    ground, non-ground, patches, refits). RCCL on the GPU box (backend "nccl"), gloo in the CPU test."""
    import torch
    import torch.distributed as dist
    if world <= 1:
        return local_rec.clone()
    if dist.get_backend() == "gloo" and local_rec.is_cuda:      # one-GPU rehearsal: gloo gathers host tensors
        torch.cuda.current_stream().synchronize()
        host = torch.empty(local_rec.numel() * world, dtype=local_rec.dtype)
        dist.all_gather_into_tensor(host, local_rec.cpu())
        return host.to(local_rec.device)
    out = torch.empty(local_rec.numel() * world, dtype=local_rec.dtype, device=local_rec.device)
    dist.all_gather_into_tensor(out, local_rec)
    return out


def gather_label_slabs(local_labels, world):
    """Optional second half of SURVEY.md section 8(e)'s gather: the batch's labels as fixed-size slabs, one per rank, padded to
    the largest rank's point count (ranks hold different frames) with 0xFF. Returns (slabs [world, slab_bytes], slab_bytes)."""
    import torch
    import torch.distributed as dist
    n = torch.tensor([local_labels.numel()], dtype=torch.int64, device=local_labels.device)
    if world > 1:
        dist.all_reduce(n, op=dist.ReduceOp.MAX)
    slab = int(n.item())
    mine = torch.full((slab,), 0xFF, dtype=torch.uint8, device=local_labels.device)
    mine[: local_labels.numel()] = local_labels
    if world <= 1:
        return mine.reshape(1, slab), slab
    if dist.get_backend() == "gloo" and mine.is_cuda:
        host = torch.empty(world * slab, dtype=torch.uint8)
        dist.all_gather_into_tensor(host, mine.cpu())
        return host.to(mine.device).reshape(world, slab), slab
    out = torch.empty(world * slab, dtype=torch.uint8, device=local_labels.device)
    dist.all_gather_into_tensor(out, mine)
    return out.reshape(world, slab), slab


def max_over_ranks(elapsed: float, device):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return elapsed
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def host_path_legs(rpw_amd, cfg_kw, frames, local_rank, seconds=0.8, slots=8):
    """The drop-in path: frames in host memory in, labels in host memory out (what every caller of the reference does:
    one std::vector<Point3D> per call). Three ways of driving it, same 8 distinct frames:
      sync       rpw_segment, one frame at a time
      in_order   rpw_submit / rpw_wait, `slots` frames in flight, a slot is waited right before it is reused (results
                 come back in submission order: what a ROS2 node that republishes in order does)
      any_order  the same slots, refilled as soon as ANY of them has finished (rpw_poll): throughput of a consumer that
                 does not need the frames in order
      in_order_reorder_buffer  any_order's slot handling, results delivered in submission order through a host-side buffer"""
    base = frames[: min(8, len(frames))]
    cap = max(len(f) for f in base)
    eng = rpw_amd.Engine(rpw_amd.PatchworkConfig(**cfg_kw), device=local_rank, max_points=cap, max_frames=1)
    out = {"frames": f"{len(base)} distinct frames of the workload, ~{cap} points, pageable host memory in, labels in host memory out",
           "slots": slots}
    lab = [np.empty(cap, np.uint8) for _ in range(slots)]
    for f in base[:3]:
        eng.segment(f)
    t0 = time.perf_counter(); k = 0; lat = []
    while time.perf_counter() - t0 < seconds:
        t1 = time.perf_counter()
        eng.segment(base[k % len(base)]); k += 1
        lat.append(time.perf_counter() - t1)
    dt = time.perf_counter() - t0
    out["sync"] = {"frames_per_sec": round(k / dt, 1), "ms_per_frame_median": round(1e3 * float(np.median(lat)), 3),
                   "ms_per_frame_max": round(1e3 * float(np.max(lat)), 3)}
    eng.set_pipeline(slots)
    for s in range(slots):
        eng.submit(s, base[s % len(base)])
    for s in range(slots):
        eng.wait(s, lab[s])
    # in order
    pending = [False] * slots; t_sub = [0.0] * slots; lat = []; i = 0; done = 0
    t0 = time.perf_counter()
    while True:
        s = i % slots
        if pending[s]:
            eng.wait(s, lab[s]); lat.append(time.perf_counter() - t_sub[s]); done += 1; pending[s] = False
        if time.perf_counter() - t0 >= seconds:
            break
        t_sub[s] = time.perf_counter(); eng.submit(s, base[i % len(base)]); pending[s] = True; i += 1
    for s in range(slots):
        if pending[s]:
            eng.wait(s, lab[s]); done += 1; pending[s] = False
    dt = time.perf_counter() - t0
    out["in_order"] = {"frames_per_sec": round(done / dt, 1), "latency_ms_median": round(1e3 * float(np.median(lat)), 3),
                       "latency_ms_max": round(1e3 * float(np.max(lat)), 3)}
    # any order
    lat = []; i = 0; done = 0
    t0 = time.perf_counter()
    for s in range(slots):
        t_sub[s] = time.perf_counter(); eng.submit(s, base[i % len(base)]); pending[s] = True; i += 1
    stop = False
    while any(pending):
        progressed = False
        for s in range(slots):
            if pending[s] and eng.poll(s):
                eng.wait(s, lab[s]); lat.append(time.perf_counter() - t_sub[s]); done += 1; pending[s] = False; progressed = True
                stop = stop or (time.perf_counter() - t0 >= seconds)
                if not stop:
                    t_sub[s] = time.perf_counter(); eng.submit(s, base[i % len(base)]); pending[s] = True; i += 1
        if not progressed:
            time.sleep(0)                                    # yield; the polls themselves are cheap event queries
    dt = time.perf_counter() - t0
    out["any_order"] = {"frames_per_sec": round(done / dt, 1), "latency_ms_median": round(1e3 * float(np.median(lat)), 3),
                        "latency_ms_max": round(1e3 * float(np.max(lat)), 3)}
    # in order again, through rpw_amd.OrderedPipeline: a finished slot is emptied into a host-side reorder buffer and
    # refilled at once; results are handed on strictly in submission order (what a node that must publish in order can do
    # with the same 8 slots)
    pipe = rpw_amd.OrderedPipeline(eng, slots=slots)
    lat = []; t_of = {}; i = 0; done = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        t_of[i] = time.perf_counter()
        rs = pipe.push(base[i % len(base)]); i += 1
        now = time.perf_counter()
        for _ in rs:
            lat.append(now - t_of.pop(done)); done += 1
    rs = pipe.flush()
    now = time.perf_counter()
    for _ in rs:
        lat.append(now - t_of.pop(done)); done += 1
    dt = time.perf_counter() - t0
    out["in_order_reorder_buffer"] = {"frames_per_sec": round(done / dt, 1), "latency_ms_median": round(1e3 * float(np.median(lat)), 3),
                                      "latency_ms_max": round(1e3 * float(np.max(lat)), 3), "max_finished_frames_held": pipe.max_held}
    eng.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=56, help="timed steps (default: a little over one second)")
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2",
                    help="c2 (default) is the configuration the metric is quoted on; the others are informational")
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU per step (one batch); 0 = the workload's default (c2: 2048)")
    ap.add_argument("--streams", type=int, default=1, help="batches in flight per GPU (HIP streams, one engine each). Default 1: one launch of "
                    "every kernel at a time, so a kernel's launch duration is its own (several batches in flight share the chip and "
                    "stretch each other's launches; throughput is the same within 1-2 %)")
    ap.add_argument("--regions", type=int, default=0, help="timed regions of --steps steps each (value = the median region); "
                    "0 = as many as it takes to measure about one second, 1 = a single region")
    ap.add_argument("--max-iter", type=int, default=100,
                    help="refit iterations; 0 = SURVEY.md 8d 'variant B' (every fit ends at once, so the recursion is forced)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-path legs and the single-frame latency "
                    "(extra launches of the same kernels: leave them out when rocprofv3 --stats averages should describe the bench batch only)")
    ap.add_argument("--no-profile-events", action="store_true")
    ap.add_argument("--gather-labels", action="store_true",
                    help="multi-GPU: after the timed region also all-gather the last batch's label slabs (padded, one per rank) "
                         "and check them against the gathered records; informational, never part of `value`")
    ap.add_argument("--end-to-end", action="store_true",
                    help="also time the PCIe-inclusive BATCH pipeline (pinned host xyz in, labels back to pinned host memory, "
                         "copies and kernels of consecutive batches overlapped on the streams); reported beside `value`, never as it")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import rpw_amd

    if not os.path.exists(rpw_amd.LIB_PATH):          # a snapshot without built artefacts: compile in-tree first
        done_file = rpw_amd.LIB_PATH + ".built"
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            rpw_amd.build()
            open(done_file, "w").write("ok")
        else:                                         # the other ranks of the node wait for local rank 0's finished build
            t_wait = time.time()
            while not os.path.exists(done_file) and time.time() - t_wait < 900:
                time.sleep(0.5)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 code path on a box with ONE GPU (RPW_BENCH_REHEARSE_ONE_GPU=1, never set by the driver): every
    # rank uses GPU 0 and the collectives go over gloo, since RCCL refuses two ranks on one device. The line it prints is
    # marked "rehearsal" and is not a measurement of anything.
    rehearse = os.environ.get("RPW_BENCH_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    wl_gen, cfg_kw, wl_frames, wl_desc = WORKLOADS[args.workload]      # c2 = configs[1]: recursion depth = 3
    if args.frames <= 0:
        args.frames = wl_frames
    if args.max_iter != 100:
        cfg_kw = dict(cfg_kw, max_iter=args.max_iter)
    frames = make_frames(args.frames, rank, args.workload)
    offsets = np.zeros(len(frames) + 1, np.uint32)
    offsets[1:] = np.cumsum([len(f) for f in frames])
    total_pts = int(offsets[-1])
    xyz_host = np.concatenate(frames, axis=0)

    # One engine (C-ABI handle: own work buffers) + one HIP stream per batch in flight. The default is ONE batch of 2048
    # frames at a time: large enough that the serial tail of a launch (a few patches that need ~100 sequential refits,
    # started first) is hidden inside it. With --streams k consecutive steps go to consecutive streams and overlap.
    n_streams = max(1, args.streams)
    engines = [rpw_amd.Engine(rpw_amd.PatchworkConfig(**cfg_kw), device=local_rank,
                              max_points=total_pts, max_frames=len(frames)) for _ in range(n_streams)]
    eng = engines[0]
    d_xyz = torch.from_numpy(xyz_host).to(dev)
    d_labels_all = [torch.empty(total_pts, dtype=torch.uint8, device=dev) for _ in range(n_streams)]
    d_labels = d_labels_all[0]
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(n_streams - 1)]
    stream = streams[0]
    F = len(frames)
    # result gather (world > 1): every step's per-frame record {ground, non-ground, patches, refits} is written on the
    # device behind the batch (event dependency on the batch's stream) and all-gathered over RCCL on a stream of its own
    rec = [torch.zeros(4 * F, dtype=torch.int32, device=dev) for _ in range(n_streams)]
    gather_stream = torch.cuda.Stream(device=dev) if world > 1 else None
    gathered = [None]
    step_no = [0]

    def step():
        k = step_no[0] % n_streams
        step_no[0] += 1
        engines[k].segment_batch_device(d_xyz.data_ptr(), offsets, d_labels_all[k].data_ptr(), None,
                                        streams[k].cuda_stream)
        if world > 1:
            # on its own stream: issued on the batch's stream the collective would hold back that stream's next batch
            engines[k].result_record_device(rec[k].data_ptr(), F, gather_stream.cuda_stream)    # waits for THIS batch (event)
            with torch.cuda.stream(gather_stream):
                gathered[0] = gather_records(rec[k], world)

    # set-up, not a step: every engine (one per stream) runs once so that its first-call costs (lazy buffers, code
    # object load) are paid before the W warm-up steps, whatever W is
    for k in range(n_streams):
        engines[k].segment_batch_device(d_xyz.data_ptr(), offsets, d_labels_all[k].data_ptr(), None, streams[k].cuda_stream)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # parity spot check of the warm-up result (not timed): frame 0 against the oracle
    st = eng.batch_stats(F)
    import oracle_lib as O
    if rank == 0:
        ref = O.segment(frames[0], O.make_config(**cfg_kw), want_residual=False, want_nodes=False)
        lab0 = d_labels[: len(frames[0])].cpu().numpy()
        if not np.array_equal(lab0, ref.labels):
            raise SystemExit("bench: HIP labels differ from the oracle on frame 0 -- refusing to report")
    chain_addends_per_batch = int(st["chain_addends"].astype(np.int64).sum())

    step_no[0] = 0
    if not args.no_profile_events:
        for e in engines:
            e.set_profiling(True)
            e.kernel_ms(reset=True)
    def timed_region():
        """EXACTLY args.steps steps between barrier + synchronize on both sides; seconds, the maximum over the ranks"""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        return max_over_ranks(time.perf_counter() - t0, dev)

    # The region is repeated until about a second of GPU time has been measured (a 20-step region lasts ~0.1 s: too short
    # to be insensitive to clock ramp); `value` is quoted on the MEDIAN region, the spread is reported beside it.
    regions = [timed_region()]
    n_regions = 1 if args.regions == 1 else (args.regions if args.regions > 1 else max(1, min(64, int(np.ceil(1.0 / max(regions[0], 1e-4))))))
    while len(regions) < n_regions:
        regions.append(timed_region())
    elapsed = float(np.median(regions))
    for e in engines:                       # patch_fit's error flag (a chain-server wait timed out) is reported here: raises
        e.batch_stats(F)
    gather_check = None
    if world > 1 and rank == 0:
        g = gathered[0].cpu().numpy().reshape(world, F, 4)
        mine = np.stack([st["n_ground"], st["n_clean"] - st["n_ground"], st["n_patches"], st["total_iterations"]], axis=1).astype(np.int32)
        gather_check = {"own_block_equals_stats": bool(np.array_equal(g[0], mine)),
                        "ranks_with_ground_points": int((g[:, :, 0].sum(axis=1) > 0).sum())}
    if world > 1 and args.gather_labels:
        # the labels of the last batch this rank computed, as one padded slab per rank (outside the timed region)
        k_last = (step_no[0] - 1) % n_streams
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        slabs, slab_bytes = gather_label_slabs(d_labels_all[k_last], world)
        torch.cuda.synchronize()
        t_slab = time.perf_counter() - t1
        if rank == 0:
            ground_from_labels = (slabs == 1).sum(dim=1).cpu().numpy()
            ground_from_records = gathered[0].cpu().numpy().reshape(world, F, 4)[:, :, 0].sum(axis=1)
            gather_check.update({"label_slab_bytes_per_rank": slab_bytes, "label_gather_ms": round(t_slab * 1e3, 3),
                                 "ground_counts_from_labels_equal_records": bool(np.array_equal(ground_from_labels, ground_from_records))})
    kms = None
    kms_solo = None
    if not args.no_profile_events:
        kms = {}
        for e in engines:
            for k, v in e.kernel_ms(reset=True).items():
                a = kms.get(k, (0.0, 0))
                kms[k] = (a[0] + v[0], a[1] + v[1])
            e.set_profiling(False)
        # The binning kernels are short; in the timed region they queue behind the resident patch_fit waves
        # of the other streams, so their durations there measure waiting. For the binning roofline they are
        # timed once more, alone on one stream, right after the timed region (same batch, same buffers).
        eng.set_profiling(True)
        for _ in range(3):
            eng.segment_batch_device(d_xyz.data_ptr(), offsets, d_labels.data_ptr(), None, stream.cuda_stream)
        torch.cuda.synchronize()
        kms_solo = eng.kernel_ms(reset=True)
        eng.set_profiling(False)

    # latency of ONE frame, device-resident (batch of 1), over the 8 distinct frames
    lat = None
    host_path = None
    if rank == 0 and not args.no_host_path:
        one = rpw_amd.Engine(rpw_amd.PatchworkConfig(**cfg_kw), device=local_rank, max_points=max(len(f) for f in frames[:8]), max_frames=1)
        lat_ms = []
        for fi in range(min(8, F)):
            off1 = np.array([0, len(frames[fi])], np.uint32)
            src = d_xyz[int(offsets[fi]): int(offsets[fi + 1])]
            for rep in range(4):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                one.segment_batch_device(src.data_ptr(), off1, d_labels.data_ptr(), None, stream.cuda_stream)
                torch.cuda.synchronize()
                if rep:
                    lat_ms.append(1e3 * (time.perf_counter() - t1))
        one.close()
        lat = {"median": round(float(np.median(lat_ms)), 4), "min": round(float(np.min(lat_ms)), 4), "max": round(float(np.max(lat_ms)), 4),
               "what": "one frame per launch, device-resident, 8 distinct frames x 3 runs"}
        eng.segment_batch_device(d_xyz.data_ptr(), offsets, d_labels.data_ptr(), None, stream.cuda_stream)
        torch.cuda.synchronize()
        host_path = host_path_legs(rpw_amd, cfg_kw, frames, local_rank)

    # The same frames as four quarter batches in flight on four streams (how a caller with a continuous supply of batches
    # would drive the library: the binning of one batch and the tail of another overlap a third's plane fits). Informational,
    # never `value`: with launches sharing the chip a kernel's launch duration is no longer its own.
    staggered = None
    if world == 1 and n_streams == 1 and not args.no_host_path and F >= 8:
        q = F // 4
        q_off = np.ascontiguousarray(offsets[: q + 1])
        q_pts = int(q_off[-1])
        q_eng = [rpw_amd.Engine(rpw_amd.PatchworkConfig(**cfg_kw), device=local_rank, max_points=q_pts, max_frames=q) for _ in range(4)]
        q_str = [torch.cuda.Stream(device=dev) for _ in range(4)]
        q_lab = [torch.empty(q_pts, dtype=torch.uint8, device=dev) for _ in range(4)]

        def q_step(i):
            k = i % 4
            q_eng[k].segment_batch_device(d_xyz.data_ptr(), q_off, q_lab[k].data_ptr(), None, q_str[k].cuda_stream)
        for i in range(8):
            q_step(i)
        torch.cuda.synchronize()
        n_q = 4 * max(4, args.steps)
        t1 = time.perf_counter()
        for i in range(n_q):
            q_step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        for e in q_eng:
            e.batch_stats(q)                                        # (raises on patch_fit's error flag)
        same = bool(torch.equal(q_lab[0], d_labels[:q_pts]))
        staggered = {"frames_per_sec": round(q * n_q / dt, 2), "batches_in_flight": 4, "frames_per_batch": q, "steps": n_q,
                     "ms_per_batch": round(1e3 * dt / n_q, 4), "labels_equal_the_single_batch_run": same}
        for e in q_eng:
            e.close()
        del q_lab

    # PCIe-inclusive BATCH pipeline, informational (SURVEY.md 8d "end-to-end"): every batch is copied in from pinned host
    # memory and its labels are copied back, on the batch's own stream, so copies overlap other batches' kernels
    e2e = None
    if world == 1 and args.end_to_end:
        h_xyz = torch.from_numpy(xyz_host).pin_memory()
        h_lab = [torch.empty(total_pts, dtype=torch.uint8).pin_memory() for _ in range(n_streams)]
        d_in = [torch.empty_like(d_xyz) for _ in range(n_streams)]

        def e2e_step(i):
            k = i % n_streams
            with torch.cuda.stream(streams[k]):
                d_in[k].copy_(h_xyz, non_blocking=True)
                engines[k].segment_batch_device(d_in[k].data_ptr(), offsets, d_labels_all[k].data_ptr(), None,
                                                streams[k].cuda_stream)
                h_lab[k].copy_(d_labels_all[k], non_blocking=True)
        for i in range(n_streams):
            e2e_step(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_e2e = min(args.steps, 36)
        for i in range(n_e2e):
            e2e_step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        same = bool(torch.equal(h_lab[0], d_labels_all[0].cpu()))
        e2e = {"frames_per_sec": round(F * n_e2e / dt, 2), "ms_per_step": round(1e3 * dt / n_e2e, 4),
               "host_GBps": round((12 + 1) * total_pts * n_e2e / dt / 1e9, 2), "labels_match_device_run": same,
               "includes": "H2D of the xyz batch + D2H of the labels over PCIe from/to pinned host memory, per batch, on its stream"}

    if rank == 0:
        frames_total = F * world * args.steps
        fps = frames_total / elapsed
        out = {
            "metric": "frames_per_sec", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": wl_desc + ", inputs resident in HBM, labels left in HBM",
                "frames_per_gpu_per_step": F, "batches_in_flight_per_gpu": n_streams,
                "points_per_frame": int(total_pts // F),
                "max_split_depth": cfg_kw.get("max_split_depth", 1000), "max_iter": cfg_kw.get("max_iter", 100), "num_sectors": 10,
                "th_dist": cfg_kw.get("th_dist", 0.2),
            },
            "api": {"value": "rpw_segment_batch_device (device pointers, whole batches)",
                    "host_path": "rpw_segment (sync) and rpw_submit / rpw_wait (pipeline): host pointers, one frame per call"},
            "ms_per_frame": round(1e3 * elapsed / (F * args.steps), 5),
            "timed_region_s": round(elapsed, 3),
            "timed_regions": {"n": len(regions), "steps_each": args.steps, "s_min": round(min(regions), 4), "s_max": round(max(regions), 4),
                              "total_s": round(sum(regions), 3), "value_is": "frames of one region / median region time",
                              "frames_per_sec_min": round(F * world * args.steps / max(regions), 1),
                              "frames_per_sec_max": round(F * world * args.steps / min(regions), 1)},
            # what the recursion did on this workload (a zero here means the headline batch never splits)
            "recursion": {"n_splits_per_frame_mean": round(float(st["n_splits"].mean()), 3),
                          "frames_with_splits": int((st["n_splits"] > 0).sum()),
                          "max_depth_reached": int(st["max_depth"].max()),
                          "nodes_per_frame_mean": round(float(st["n_nodes"].mean()), 2),
                          "refits_per_frame_mean": round(float(st["total_iterations"].mean()), 1)},
            "single_frame_latency_ms": None if lat is None else lat["median"],
            "single_frame_latency": lat,
            "host_path": host_path,
            "four_batches_in_flight": staggered,
        }
        if rehearse:
            out["rehearsal"] = "every rank on GPU 0, collectives over gloo: a run of the N > 1 code path, not a measurement"
        if gather_check is not None:
            out["result_gather"] = gather_check
        if e2e is not None:
            out["end_to_end"] = e2e
        if kms is not None:
            fit_ms, fit_n = kms["patch_fit"]
            dom = max(kms, key=lambda k: kms[k][0])
            per = {k: (round(v[0] / max(1, v[1]), 5)) for k, v in kms.items()}
            out["kernel_ms_per_launch"] = per
            avg_s = (kms[dom][0] / max(1, kms[dom][1])) * 1e-3
            bpp = BYTES_PER_POINT_FIT if dom == "patch_fit" else BYTES_PER_POINT_BIN
            ach = bpp * total_pts / avg_s / 1e9
            # (a batch runs patch_fit's chain-server kernel; the other two forms are patch_fit1_kernel<1|2>)
            knames = {"patch_fit": ("patch_fit_srv_kernel", "patch_fit1_kernel"), "bin_count": ("bin_count_kernel",),
                      "bin_scatter": ("bin_scatter_kernel",), "scan_offsets": ("scan_offsets_kernel",)}[dom]
            traffic, traffic_src = None, None
            for kname in knames:
                traffic, traffic_src = load_pmc_traffic(kname, F)
                if traffic is not None:
                    break
            out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                               "traffic_source": traffic_src, "algorithmic_bytes_per_point": bpp,
                               "algorithmic_bytes_per_launch": bpp * total_pts,
                               "launches_in_flight": n_streams,
                               "note": "algorithmic bytes assume a patch is read once; the exact refit loop re-reads it every "
                                       "iteration (sequential binary32 sums, patches larger than LDS), so the traffic is what "
                                       "matters: see memory_system below and DESIGN.md section 4. The duration is the average "
                                       "launch in the timed region, where launches_in_flight launches share the chip."}
            # what the memory system moves per step (L2-miss traffic of the three kernels from the PMC passes) over the
            # step time: how far the whole pipeline is from HBM's 8 TB/s
            t_all = [load_pmc_traffic(k, F)[0] for k in ("bin_count_kernel", "scan_offsets_kernel", "bin_scatter_kernel")] + [traffic if dom == "patch_fit" else None]
            if all(t is not None for t in t_all):
                step_s = elapsed / args.steps
                rate = sum(t_all) / step_s / 1e9
                out["memory_system"] = {"traffic_bytes_per_step": int(sum(t_all)), "traffic_source": traffic_src,
                                        "ms_per_step": round(step_s * 1e3, 4), "achieved": round(rate, 1), "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": round(rate / HBM_PEAK_GBS, 4),
                                        # what this chip streams from HBM at full occupancy (tools/microbench_loads.hip,
                                        # profiles/r02_microbench_loads.txt: 6.24-6.29 TB/s)
                                        "frac_of_measured_streaming_rate": round(rate / 6260.0, 4),
                                        "note": "FETCH_SIZE x2 + WRITE_SIZE of bin_count + scan_offsets + bin_scatter + patch_fit per launch "
                                                "(requests that leave L2; Infinity Cache hits are included) / time per step"}
            # the bound that governs patch_fit: sequential binary32 adds. A dependent v_add_f32 retires every
            # 3.4 ns (8.2 cycles, tools/microbench_chain.hip); the chip tops out at 256 CUs x 8 chains / 3.4 ns.
            chain_peak = 256 * 8 / 3.4e-9
            chain_rate = chain_addends_per_batch * world * args.steps / elapsed
            out["chain_roofline"] = {"kernel": "patch_fit", "bound": "dependent fp32 add latency",
                                     "achieved": round(chain_rate / 1e9, 2), "peak": round(chain_peak / 1e9, 1),
                                     "unit": "G sequential addends/s (per accumulator row)",
                                     "frac": round(chain_rate / chain_peak, 4),
                                     "addends_per_frame": chain_addends_per_batch // F}
            solo = {k: kms_solo[k][0] / max(1, kms_solo[k][1]) for k in kms_solo}
            bin_kernels = ("bin_count", "scan_offsets", "bin_scatter")
            bin_s = sum(solo[k] for k in bin_kernels) * 1e-3
            achb = BYTES_PER_POINT_BIN * total_pts / bin_s / 1e9
            tb = [load_pmc_traffic(k + "_kernel", F) for k in bin_kernels]
            # the WHOLE binning pass of north_star ("radial-sector scatter + per-bin prefix-sum compaction"): a stable counting
            # sort by patch -- count, scan, scatter -- that leaves every patch contiguous, in input order
            out["roofline_binning"] = {"kernels": "+".join(bin_kernels), "bound": "hbm",
                                       "measured": "one stream, right after the timed region (in it these kernels queue "
                                                   "behind other streams' patch_fit waves: see in_timed_region_ms)",
                                       "ms_per_launch": round(bin_s * 1e3, 4),
                                       "ms_per_kernel": {k: round(solo[k], 4) for k in bin_kernels},
                                       "in_timed_region_ms": round(sum(per[k] for k in bin_kernels), 4),
                                       "achieved": round(achb, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(achb / HBM_PEAK_GBS, 5),
                                       "traffic": None if any(t[0] is None for t in tb) else sum(t[0] for t in tb),
                                       "traffic_source": tb[0][1],
                                       "algorithmic_bytes_per_point": BYTES_PER_POINT_BIN,
                                       "algorithmic_bytes_per_launch": BYTES_PER_POINT_BIN * total_pts,
                                       "actual_bytes_per_point": f"{BYTES_PER_POINT_BIN_ACTUAL}: 12 read (count pass) + 12 read again + 16 written (scatter pass)",
                                       "frac_of_actual_bytes": round(BYTES_PER_POINT_BIN_ACTUAL * total_pts / bin_s / 1e9 / HBM_PEAK_GBS, 5),
                                       "previous_design": "rounds 1-2: one pass to tile-sorted records (28 B/pt) + a gather inside patch_fit "
                                                          "(32 B/pt): 0.36 + 0.68 ms per 512-frame launch, 22 % of peak for the whole pass"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames, cfg_kw)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for e in engines:
        e.close()


if __name__ == "__main__":
    main()
