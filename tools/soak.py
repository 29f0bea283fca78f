#!/usr/bin/env python3
"""One-off soak: the randomised HIP-vs-oracle sweep of tests/test_gpu_more.py with other seeds and many more cases.
Test infrastructure (it calls the oracle): run on the GPU box,  python tools/soak.py [cases] [seed]  -> prints a summary."""
import os, sys, time
import numpy as np
import torch  # noqa: F401  (before librpw_hip.so)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rpw_amd
from rpw_amd import workloads as W
from parity_util import cfg_pair, assert_frame_parity


def batch_soak(batches, seed):
    """Random batches of 1..24 frames of mixed sizes (empty and tiny ones included) through the device-resident batch
    call, every frame compared with the oracle."""
    rs = np.random.RandomState(seed)
    t0 = time.time(); nfr = 0
    for b in range(batches):
        F = int(rs.randint(1, 25)); frames = []
        for _ in range(F):
            kind = rs.randint(0, 6)
            if kind == 0: x = np.zeros((int(rs.randint(0, 4)), 3), np.float32) + np.float32(rs.uniform(1, 30))
            elif kind == 1: x = W.c1_test_cloud(int(rs.randint(0, 10000)), n=int(rs.randint(3, 9000)))
            elif kind == 2: x = W.c2_hdl64(int(rs.randint(0, 10000)))[:: int(rs.randint(1, 12))]
            elif kind == 3: x = W.c4_iac_fused(int(rs.randint(0, 10000)), rays_per_sensor=int(rs.randint(2000, 30000)))
            elif kind == 4: x = W.c1_test_cloud(int(rs.randint(0, 10000)), n=int(rs.choice([2047, 2048, 2049, 4096, 6143])))
            else: x = W.c2_hdl64(int(rs.randint(0, 10000)))[: int(rs.randint(1, 60000))]
            frames.append(np.ascontiguousarray(x, np.float32))
        kw = dict(num_sectors=int(rs.choice([1, 4, 10, 16, 33, 128])), max_split_depth=int(rs.choice([0, 2, 3, 1000])),
                  max_iter=int(rs.choice([0, 3, 100])), filtering_radius=float(rs.choice([20.0, 50.0, 150.0])))
        pc, oc = cfg_pair(**kw)
        off = np.zeros(F + 1, np.uint32); off[1:] = np.cumsum([len(f) for f in frames]); N = int(off[-1])
        eng = rpw_amd.Engine(pc, max_points=max(N, 1), max_frames=F)
        d = torch.from_numpy(np.concatenate(frames) if N else np.zeros((0, 3), np.float32)).cuda()
        lab = torch.empty(max(N, 1), dtype=torch.uint8, device="cuda"); res = torch.empty(max(N, 1), dtype=torch.float32, device="cuda")
        eng.segment_batch_device(d.data_ptr(), off, lab.data_ptr(), res.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        st = eng.batch_stats(F); L, R = lab.cpu().numpy(), res.cpu().numpy()
        for f, xyz in enumerate(frames):
            sl = slice(int(off[f]), int(off[f + 1]))
            assert_frame_parity(xyz, L[sl], R[sl], {k: st[k][f] for k in st.dtype.names}, None, oc, check_nodes=False, tag=f"batch {b} frame {f} {kw}")
        eng.close(); nfr += F
        if b % 25 == 24: print(f"{b + 1} batches ok, {nfr} frames, {time.time() - t0:.0f} s", flush=True)
    print(f"batch soak: {batches} batches, {nfr} frames: labels bit-exact, stats and residuals equal to the oracle ({time.time() - t0:.0f} s)")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--batch":
        return batch_soak(int(sys.argv[2]) if len(sys.argv) > 2 else 100, int(sys.argv[3]) if len(sys.argv) > 3 else 4242)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
    gens = [lambda s: W.c1_test_cloud(s, n=int(rs.randint(500, 30000))), lambda s: W.c2_hdl64(s)[:: int(rs.randint(1, 6))],
            lambda s: W.c4_iac_fused(s, rays_per_sensor=20000)]
    t0 = time.time(); npts = 0
    for it in range(cases):
        xyz = gens[it % 3](int(rs.randint(0, 100000))).copy()
        if rs.rand() < 0.5:
            xyz *= np.float32(rs.choice([0.1, 0.25, 0.5, 2.0, 3.0, 7.0]))
        if rs.rand() < 0.5:
            a = np.float32(rs.uniform(0, 6.28)); c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
            xyz = np.column_stack([xyz[:, 0] * c - xyz[:, 1] * s, xyz[:, 0] * s + xyz[:, 1] * c, xyz[:, 2]]).astype(np.float32)
        if rs.rand() < 0.3:
            xyz = xyz[rs.permutation(len(xyz))]
        if rs.rand() < 0.4:
            idx = rs.randint(0, len(xyz), 20)
            xyz[idx, rs.randint(0, 3, 20)] = rs.choice([np.nan, np.inf, -np.inf], 20)
        kw = dict(
            num_sectors=int(rs.choice([1, 2, 3, 5, 7, 10, 12, 16, 31, 37, 64, 100, 128])),
            filtering_radius=float(rs.choice([2.0, 5.0, 30.0, 50.0, 80.0, 150.0, 1000.0])),
            max_iter=int(rs.choice([0, 1, 2, 3, 5, 20, 50, 100])),
            max_split_depth=int(rs.choice([0, 1, 2, 3, 6, 10, 1000])),
            th_dist=float(rs.choice([0.02, 0.05, 0.1, 0.2, 0.5])),
            th_seeds=float(rs.choice([0.05, 0.15, 0.5])),
            sensor_height=float(rs.choice([-2.0, -1.73, -1.5, 0.0, 0.6, 1.2, 3.0])),
            adaptive_seed_height=bool(rs.rand() < 0.7),
        )
        pc, oc = cfg_pair(**kw)
        eng = rpw_amd.Engine(pc, max_points=len(xyz), max_frames=1)
        eng.set_trace(True, max(1 << 16, len(xyz)))
        labels, res, st = eng.segment(xyz, want_residual=True)
        assert_frame_parity(xyz, labels, res, st, eng.nodes(max(1 << 16, len(xyz))), oc, tag=f"soak case {it} {kw}")
        eng.close(); npts += len(xyz)
        if it % 25 == 24:
            print(f"{it + 1} cases ok, {npts} points, {time.time() - t0:.0f} s", flush=True)
    print(f"soak: {cases} cases, {npts} points: labels bit-exact, stats, residuals and trees equal to the oracle ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
