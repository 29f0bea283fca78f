#!/usr/bin/env python3
"""bench.py -- frames/sec of the Recursive Patchwork hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (classify+bin -> stable scatter -> per-patch recursive plane
fit -> labels) over one batch of synthetic HDL-64-like frames that is already resident in HBM
(BASELINE.json configs[1]/[2]: ~127 k points per frame, max_split_depth = 3). Frames are independent,
so N GPUs run N such batches with no data-path collective ("weak" scaling); the only RCCL call is
the trivial gather of the per-frame result records.

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
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_POINT_FIT = 17        # 12 sorted xyz + 4 index + 1 label
BYTES_PER_POINT_BIN = 28        # 12 xyz in + 12 sorted SoA out + 4 index out


WORKLOADS = {   # name: (generator, config, default frames per step, description) -- BASELINE.json `configs`
    "c2": ("c2_hdl64", dict(max_split_depth=3), 512,
           "c2_hdl64 (BASELINE.json configs[1]/[2]): seeded HDL-64-like ray-cast frames"),
    "c1": ("c1_test_cloud", dict(), 2048, "c1_test_cloud (configs[0]): the reference test's 20 k-point clouds"),
    "c4": ("c4_iac_fused", dict(max_split_depth=1000), 256,
           "c4_iac_fused (configs[3]): 3 fused solid-state sensors, ~296 k points, deep recursion allowed"),
    "c5": ("c5_dense", dict(max_split_depth=6), 64, "c5_dense (configs[4]): dense 1.09 M-point frames"),
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


def cpu_baseline(frames, ocfg_kw, budget_s: float = 16.0):
    """The CPU oracle (a port of the reference algorithm, stdout silent) timed on this box's host
    cores: single-thread ms/frame and frame-parallel frames/s on all cores (ctypes releases the GIL)."""
    import oracle_lib as O
    from concurrent.futures import ThreadPoolExecutor
    cfg = O.make_config(**ocfg_kw)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    O.segment(frames[0], cfg, want_residual=False, want_nodes=False)      # warm-up
    # single thread
    t0 = time.perf_counter(); n1 = 0
    while time.perf_counter() - t0 < budget_s * 0.4 and n1 < 400:
        O.segment(frames[n1 % len(frames)], cfg, want_residual=False, want_nodes=False); n1 += 1
    t1 = time.perf_counter() - t0
    # all cores, one frame per thread
    def work(i):
        O.segment(frames[i % len(frames)], cfg, want_residual=False, want_nodes=False)
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, range(cores)))                                   # warm-up
        t0 = time.perf_counter(); nmt = 0
        while time.perf_counter() - t0 < budget_s * 0.6:
            list(ex.map(work, range(nmt, nmt + cores)))
            nmt += cores
        tp = time.perf_counter() - t0
    return {
        "value": round(nmt / tp, 2), "unit": "frames/s", "cores": cores, "kind": "port",
        "sample": f"{nmt} frames of the bench batch, one frame per thread on {cores} threads "
                  f"({tp:.1f} s); single thread: {1e3 * t1 / n1:.2f} ms/frame over {n1} frames",
        "single_thread_ms_per_frame": round(1e3 * t1 / n1, 3),
    }


def load_pmc_traffic(kernel: str, frames_per_launch: int):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE, separate passes; tools/collect_traffic.sh). Only valid for the batch size it was taken at."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic_bench.json")
    try:
        d = json.load(open(path))
        if d.get("_meta", {}).get("frames_per_launch", 256) != frames_per_launch:
            return None, None
        v = d.get(kernel + "_kernel")
        return (int(v["hbm_bytes_per_launch"]), "profiles/r01_pmc_traffic_bench.json") if v else (None, None)
    except Exception:
        return None, None


def gather_counts(local_counts, world):
    """The path's only collective: all-gather of the per-frame result record (ground-point counts).
    RCCL on the GPU box (backend "nccl"), gloo in the CPU test."""
    import torch
    import torch.distributed as dist
    if world <= 1:
        return local_counts.clone()
    out = torch.empty(local_counts.numel() * world, dtype=local_counts.dtype, device=local_counts.device)
    dist.all_gather_into_tensor(out, local_counts)
    return out


def max_over_ranks(elapsed: float, device):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return elapsed
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=18)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2",
                    help="c2 (default) is the configuration the metric is quoted on; the others are informational")
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU per step (one batch); 0 = the workload's default (c2: 512)")
    ap.add_argument("--streams", type=int, default=3, help="batches in flight per GPU (HIP streams, one engine each)")
    ap.add_argument("--max-iter", type=int, default=100,
                    help="refit iterations; 0 = SURVEY.md 8d 'variant B' (every fit ends at once, so the recursion is forced)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-frame-latency", action="store_true",
                    help="also time batches of ONE frame (extra launches of the same kernels; off by default so that "
                         "rocprofv3 --stats averages describe the bench batch only)")
    ap.add_argument("--no-profile-events", action="store_true")
    ap.add_argument("--end-to-end", action="store_true",
                    help="also time the PCIe-inclusive pipeline (pinned host xyz in, labels back to pinned host memory, "
                         "copies and kernels of consecutive batches overlapped on the streams); reported beside `value`, never as it")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import rpw_amd

    if not os.path.exists(rpw_amd.LIB_PATH):          # a snapshot without built artefacts: compile in-tree first
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            rpw_amd.build()
        else:                                         # the other ranks of the node wait for local rank 0's build
            t_wait = time.time()
            while not os.path.exists(rpw_amd.LIB_PATH) and time.time() - t_wait < 600:
                time.sleep(1.0)
            time.sleep(2.0)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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

    # One engine (C-ABI handle: own work buffers) + one HIP stream per batch in flight. Consecutive
    # steps go to consecutive streams, so the serial tail of one batch (a few patches that need
    # ~100 sequential refits) overlaps with the next batches instead of idling the chip.
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
    gather_in = torch.zeros(F, dtype=torch.int32, device=dev)
    gather_stream = torch.cuda.Stream(device=dev) if world > 1 else None
    step_no = [0]

    def step():
        k = step_no[0] % n_streams
        step_no[0] += 1
        engines[k].segment_batch_device(d_xyz.data_ptr(), offsets, d_labels_all[k].data_ptr(), None,
                                        streams[k].cuda_stream)
        if world > 1:
            # trivial result gather: ground-point count per frame (RCCL all-gather of F int32 per rank), on a stream
            # of its own: issued on a batch's stream it would make that stream's next batch wait for the collective
            with torch.cuda.stream(gather_stream):
                gather_counts(gather_in, world)

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
    gather_in.copy_(torch.from_numpy(st["n_ground"].astype(np.int32)))
    chain_addends_per_batch = int(st["chain_addends"].astype(np.int64).sum())

    step_no[0] = 0
    if not args.no_profile_events:
        for e in engines:
            e.set_profiling(True)
            e.kernel_ms(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, dev)
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

    # single-frame latency (batch of 1), informational
    lat_ms = None
    if rank == 0 and args.single_frame_latency:
        off1 = offsets[:2].copy()
        for _ in range(3):
            eng.segment_batch_device(d_xyz.data_ptr(), off1, d_labels.data_ptr(), None, stream.cuda_stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            eng.segment_batch_device(d_xyz.data_ptr(), off1, d_labels.data_ptr(), None, stream.cuda_stream)
            torch.cuda.synchronize()
        lat_ms = 1e3 * (time.perf_counter() - t1) / reps

    # PCIe-inclusive pipeline, informational (SURVEY.md 8d "end-to-end"): every batch is copied in from pinned host
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
        for i in range(args.steps):
            e2e_step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        same = bool(torch.equal(h_lab[0], d_labels_all[0].cpu()))
        e2e = {"frames_per_sec": round(F * args.steps / dt, 2), "ms_per_step": round(1e3 * dt / args.steps, 4),
               "host_GBps": round((12 + 1) * total_pts * args.steps / dt / 1e9, 2), "labels_match_device_run": same,
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
            },
            "ms_per_frame": round(1e3 * elapsed / (F * args.steps), 5),
            "single_frame_latency_ms": None if lat_ms is None else round(lat_ms, 4),
        }
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
            traffic, traffic_src = load_pmc_traffic(dom, F)
            out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                               "traffic_source": traffic_src, "algorithmic_bytes_per_point": bpp,
                               "algorithmic_bytes_per_launch": bpp * total_pts,
                               "note": "patch_fit is bound by sequential binary32 add chains (8.2 cycles per dependent "
                                       "v_add_f32), not by HBM; see DESIGN.md section 4"}
            # the bound that actually governs patch_fit: sequential binary32 adds. A dependent v_add_f32 retires every
            # 3.4 ns (8.2 cycles, tools/microbench_chain.hip); a SIMD sustains two such chains at full rate, so the chip
            # tops out at 256 CUs x 8 chains / 3.4 ns, whatever the lanes beside the chain lanes do.
            chain_peak = 256 * 8 / 3.4e-9
            chain_rate = chain_addends_per_batch * world * args.steps / elapsed
            out["chain_roofline"] = {"kernel": "patch_fit", "bound": "dependent fp32 add latency",
                                     "achieved": round(chain_rate / 1e9, 2), "peak": round(chain_peak / 1e9, 1),
                                     "unit": "G sequential addends/s (per accumulator row)",
                                     "frac": round(chain_rate / chain_peak, 4),
                                     "addends_per_frame": chain_addends_per_batch // F}
            bin_s = sum(kms_solo[k][0] / max(1, kms_solo[k][1]) for k in ("bin_tile", "scan_offsets")) * 1e-3
            achb = BYTES_PER_POINT_BIN * total_pts / bin_s / 1e9
            tb1, tsrc1 = load_pmc_traffic("bin_tile", F)
            tb2, _ = load_pmc_traffic("scan_offsets", F)
            out["roofline_binning"] = {"kernels": "bin_tile+scan_offsets", "bound": "hbm",
                                       "measured": "one stream, right after the timed region (in it these kernels queue "
                                                   "behind other streams' patch_fit waves)",
                                       "ms_per_launch": round(bin_s * 1e3, 4),
                                       "achieved": round(achb, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(achb / HBM_PEAK_GBS, 5),
                                       "traffic": None if tb1 is None or tb2 is None else tb1 + tb2,
                                       "traffic_source": tsrc1,
                                       "algorithmic_bytes_per_point": BYTES_PER_POINT_BIN,
                                       "algorithmic_bytes_per_launch": BYTES_PER_POINT_BIN * total_pts}
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
