"""More GPU parity: batches, edge cases, full-size configs, golden fixtures, the C++ drop-in class."""
import json
import os
import subprocess
import zlib

import numpy as np
import pytest

import oracle_lib as O
import rpw_amd
from rpw_amd import workloads as W
from parity_util import cfg_pair, assert_frame_parity, sort_nodes, nodes_crc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=lambda c: f"{c['workload']}-{c['seed']}-{len(c['config'])}")
def test_hip_against_golden_fixtures(engine_factory, case):
    """BASELINE.json's full sizes (C1 20k ... C5 1.09M points) against the committed fixtures."""
    xyz = W.make(case["workload"], case["seed"])
    assert crc(xyz) == case["input_crc32"]
    kw = dict(case["config"])
    if "adaptive_seed_height" in kw:
        kw["adaptive_seed_height"] = bool(kw["adaptive_seed_height"])
    eng = engine_factory(rpw_amd.PatchworkConfig(**kw), max_points=len(xyz), max_frames=1)
    eng.set_trace(True, max(1 << 16, len(xyz)))
    labels, res, st = eng.segment(xyz, want_residual=True)
    assert crc(labels) == case["labels_crc32"]
    for k, v in case["stats"].items():
        assert int(st[k]) == v, k
    nodes = eng.nodes(max(1 << 16, len(xyz)))
    assert len(nodes) == case["stats"]["n_nodes"] and nodes_crc(nodes) == case["nodes_crc32"]      # the whole recursion tree
    # residuals: within 1e-5 of the oracle is the bar; bit equality is expected and reported
    if crc(res) != case["residual_crc32"]:
        ref = O.segment(xyz, O.make_config(**case["config"]), want_nodes=False).residual
        ok = ref >= 0
        assert np.array_equal(res < 0, ~ok) and np.abs(res[ok] - ref[ok]).max() <= 1e-5


@pytest.mark.parametrize("name,kw", [("c5", dict(max_split_depth=6)), ("c5", dict(max_split_depth=6, max_iter=0)),
                                     ("c4", dict(max_iter=3, max_split_depth=8, th_dist=0.05)),
                                     # NATURAL recursion (no max_iter = 0): fits lose their inliers and split to the configured depth
                                     ("rough", dict(max_split_depth=3, th_dist=0.0005)), ("rough", dict(max_split_depth=6, th_dist=0.0005)),
                                     ("rough", dict(max_split_depth=1000, th_dist=0.002, num_sectors=16)),
                                     # patches of more than 65536 points: the index list switches to 32-bit entries
                                     ("c5", dict(num_sectors=1, max_split_depth=2)), ("c5", dict(num_sectors=2, max_iter=4, max_split_depth=1000, th_dist=0.01))])
def test_full_size_parity_with_trace(engine_factory, name, kw):
    xyz = W.make(name, 0)
    pc, oc = cfg_pair(**kw)
    eng = engine_factory(pc, max_points=len(xyz), max_frames=1)
    eng.set_trace(True, 1 << 18)
    labels, res, st = eng.segment(xyz, want_residual=True)
    assert_frame_parity(xyz, labels, res, st, eng.nodes(1 << 18), oc, tag=f"{name}/{kw}")
    if name == "rough" and kw["max_split_depth"] <= 6:
        assert st["max_depth"] == kw["max_split_depth"] and st["n_splits"] > 100      # the recursion did reach the configured depth


def test_result_record_on_device(engine_factory):
    """rpw_result_record_device: the per-frame record the multi-GPU gather sends, written behind the batch on ANOTHER
    stream (event dependency), equals the batch stats."""
    import torch
    # (the last frame reaches beyond filtering_radius: its outside points belong to the reference's non_ground_points)
    frames = [W.c2_hdl64(1), W.c1_test_cloud(2), W.c2_hdl64(2)[::2], W.c1_test_cloud(4) * np.array([4, 4, 1], np.float32)]
    off = np.zeros(len(frames) + 1, np.uint32); off[1:] = np.cumsum([len(f) for f in frames])
    pc, oc = cfg_pair(max_split_depth=3)
    eng = engine_factory(pc, max_points=int(off[-1]), max_frames=len(frames))
    dev = torch.device("cuda", 0)
    d_xyz = torch.from_numpy(np.concatenate(frames)).to(dev)
    d_lab = torch.empty(int(off[-1]), dtype=torch.uint8, device=dev)
    rec = torch.full((4 * len(frames),), -1, dtype=torch.int32, device=dev)
    s_batch, s_gather = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    eng.segment_batch_device(d_xyz.data_ptr(), off, d_lab.data_ptr(), None, s_batch.cuda_stream)
    eng.result_record_device(rec.data_ptr(), len(frames), s_gather.cuda_stream)
    s_gather.synchronize()
    got = rec.cpu().numpy().reshape(-1, 4)
    st = eng.batch_stats(len(frames))
    for f, xyz in enumerate(frames):
        ref = O.segment(xyz, oc, want_residual=False, want_nodes=False)
        # non-ground as the reference returns it (recursive_patchwork.cpp:402-419): size of its second vector
        _, ng = O.filter_ground_points(xyz, oc)
        assert got[f].tolist() == [ref.stats["n_ground"], len(ng), ref.stats["n_patches"], ref.stats["total_iterations"]]
        assert len(ng) == ref.stats["n_clean"] - ref.stats["n_ground"]
        if f == len(frames) - 1:
            assert ref.stats["n_clean"] > ref.stats["n_zone"]        # the frame does have points beyond the radius
        assert int(st["n_ground"][f]) == ref.stats["n_ground"]
    torch.cuda.synchronize()
    lab = d_lab.cpu().numpy()
    for f, xyz in enumerate(frames):
        assert np.array_equal(lab[off[f]: off[f + 1]], O.segment(xyz, oc, want_residual=False, want_nodes=False).labels)


def test_batch_of_mixed_frames_device_resident(engine_factory):
    """One launch over a batch of different frames (incl. an empty, a tiny and a non-finite frame)."""
    import torch
    frames = [W.c2_hdl64(5), W.c1_test_cloud(3), np.zeros((0, 3), np.float32), W.c4_iac_fused(1)[:50000],
              np.array([[1, 2, 0], [np.nan, 0, 0]], np.float32), W.c2_hdl64(6)[::3],
              np.full((100, 3), np.inf, np.float32), W.c1_test_cloud(4, n=2049)]
    pc, oc = cfg_pair(max_split_depth=3)
    off = np.zeros(len(frames) + 1, np.uint32); off[1:] = np.cumsum([len(f) for f in frames])
    eng = engine_factory(pc, max_points=int(off[-1]), max_frames=len(frames))
    eng.set_trace(True, 1 << 17)
    d = torch.from_numpy(np.concatenate(frames)).cuda()
    lab = torch.empty(int(off[-1]), dtype=torch.uint8, device="cuda")
    res = torch.empty(int(off[-1]), dtype=torch.float32, device="cuda")
    for rep in range(2):       # second run reuses every work buffer
        eng.segment_batch_device(d.data_ptr(), off, lab.data_ptr(), res.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = eng.batch_stats(len(frames))
    nodes = eng.nodes(1 << 17)
    L, R = lab.cpu().numpy(), res.cpu().numpy()
    for f, xyz in enumerate(frames):
        sl = slice(int(off[f]), int(off[f + 1]))
        assert_frame_parity(xyz, L[sl], R[sl], {k: st[k][f] for k in st.dtype.names}, nodes[nodes["frame"] == f], oc, tag=f"frame{f}")


def test_edge_cases(engine_factory):
    pc, oc = cfg_pair()
    eng = engine_factory(pc, max_points=1 << 16, max_frames=1)
    rs = np.random.RandomState(0)
    cases = {
        "empty": np.zeros((0, 3), np.float32),
        "one": np.array([[3, 4, 0]], np.float32),
        "two": np.array([[3, 4, 0], [5, 1, 0.1]], np.float32),
        "all_nan": np.full((10, 3), np.nan, np.float32),
        "zone_lt_3": np.array([[300, 0, 0], [2, 2, 0], [np.nan, 1, 1], [400, 3, 0]], np.float32),
        "inside_rmin": (rs.uniform(-0.5, 0.5, (200, 3))).astype(np.float32),
        "on_radius": np.array([[150, 0, 0], [0, 150, 0], [-150, 0, 0], [90, 120, 0], [1, 0, 0], [0, -1, 0]], np.float32),
        "axes": np.array([[x, y, 0.01 * i] for i, (x, y) in enumerate([(5, 0), (0, 5), (-5, 0), (0, -5), (5, -0.0), (5, 1e-30), (5, -1e-30)] * 5)], np.float32),
        "identical": np.tile(np.array([[10, 10, -1]], np.float32), (500, 1)),
        "flat": np.column_stack([rs.uniform(5, 6, 300), rs.uniform(5, 6, 300), np.full(300, 0.01)]).astype(np.float32),
        "ragged_tile": W.c1_test_cloud(8, n=2047),
        "quantized": (np.round(W.c1_test_cloud(9, n=5000) * 4) / 4).astype(np.float32),
        "huge_coords": (W.c1_test_cloud(10, n=3000) * np.float32(1e18)).astype(np.float32),
        "denormal": (W.c1_test_cloud(12, n=1000) * np.float32(1e-42)).astype(np.float32),
    }
    for name, xyz in cases.items():
        labels, res, st = eng.segment(xyz, want_residual=True)
        assert_frame_parity(xyz, labels, res, st, None, oc, check_nodes=False, tag=name)
        g, ng = rpw_amd.split_by_labels(xyz, labels, st["n_zone"])
        g2, ng2 = O.filter_ground_points(xyz, oc)
        assert np.array_equal(g, g2, equal_nan=True) and np.array_equal(ng, ng2, equal_nan=True), name


def test_forced_deep_recursion_and_ties(engine_factory):
    """Quantised coordinates give many equal split values (left child may keep every point)."""
    xyz = (np.round(W.c2_hdl64(7)[::2] * 2) / 2).astype(np.float32)
    total_splits = 0
    for kw in (dict(max_iter=0, max_split_depth=40), dict(max_iter=2, th_dist=0.02, max_split_depth=12),
               dict(max_iter=1, adaptive_seed_height=False, th_dist=0.05, max_split_depth=6)):
        pc, oc = cfg_pair(**kw)
        eng = engine_factory(pc, max_points=len(xyz), max_frames=1)
        eng.set_trace(True, 1 << 18)
        labels, res, st = eng.segment(xyz, want_residual=True)
        assert_frame_parity(xyz, labels, res, st, eng.nodes(1 << 18), oc, tag=str(kw))
        total_splits += int(st["n_splits"])
    assert total_splits > 0


def test_period_two_shortcut_is_exact(engine_factory):
    """max_iter parity decides which of two alternating masks survives: odd and even max_iter."""
    xyz = W.c2_hdl64(2)
    for mi in (7, 8, 30, 31, 100):
        pc, oc = cfg_pair(max_split_depth=3, max_iter=mi, th_dist=0.05)
        eng = engine_factory(pc, max_points=len(xyz), max_frames=1)
        labels, res, st = eng.segment(xyz, want_residual=True)
        assert_frame_parity(xyz, labels, res, st, None, oc, check_nodes=False, tag=f"max_iter={mi}")


def test_run_to_run_determinism_and_python_mirror(engine_factory):
    xyz = W.c2_hdl64(4)
    rp = rpw_amd.RecursivePatchwork(rpw_amd.PatchworkConfig(max_split_depth=3), max_points=len(xyz))
    g1, n1 = rp.filterGroundPoints(xyz)
    g2, n2 = rp.filterGroundPoints(xyz)
    assert np.array_equal(g1, g2) and np.array_equal(n1, n2)
    og, on = O.filter_ground_points(xyz, O.make_config(max_split_depth=3))
    assert np.array_equal(g1, og) and np.array_equal(n1, on)
    assert len(rp.cleanPoints(np.array([[1, 2, 3], [np.nan, 0, 0]], np.float32))) == 1


def test_cpp_dropin_program(tmp_path):
    """Compile and run tests/cpp/test_dropin.cpp against the in-tree libraries."""
    pkg = rpw_amd.PKG_DIR
    subprocess.check_call(["make", "-C", pkg, "-s", "librecursive_patchwork.so"])
    exe = str(tmp_path / "test_dropin")
    orc = os.path.join(ROOT, "oracle")
    O.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(pkg, "cpp"), "-I", os.path.join(ROOT, "include"), "-I", orc,
                           "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "test_dropin.cpp"), "-o", exe,
                           "-L", pkg, "-lrecursive_patchwork", "-lrpw_hip", "-L", orc, "-lrpw_oracle",
                           f"-Wl,-rpath,{pkg}", f"-Wl,-rpath,{orc}"])
    out = subprocess.check_output([exe], text=True, timeout=300)
    assert "drop-in OK" in out


def test_boundary_torture(engine_factory):
    """Points on / one ulp around every ring edge and sector bound: the sqrt-free ring thresholds and the
    fast sector test (exact libm-model fallback near a bound) must classify exactly like the oracle."""
    rs = np.random.RandomState(5)
    for kw in (dict(), dict(num_sectors=8, filtering_radius=50.0), dict(num_sectors=31, filtering_radius=80.0),
               dict(num_sectors=40, filtering_radius=0.5), dict(num_sectors=3, filtering_radius=1.0)):
        pc, oc = cfg_pair(**kw)
        e, b = O.tables(oc)
        pts = []
        ups = lambda v, k: np.nextafter(np.float32(v), np.float32(np.inf if k > 0 else -np.inf)) if k else np.float32(v)
        for edge in list(e) + [oc.filtering_radius]:
            for k in (-2, -1, 0, 1, 2):
                d = np.float32(edge)
                for _ in range(abs(k)):
                    d = ups(d, k)
                for th in rs.uniform(0, 2 * np.pi, 6):
                    pts.append((d * np.cos(th), d * np.sin(th), rs.normal(0, 0.05)))
                pts += [(d, 0.0, 0.0), (0.0, d, 0.0), (-d, 0.0, 0.0), (0.0, -d, 0.0), (d, -0.0, 0.0)]
        for ang in b:
            for k in range(-3, 4):
                a = np.float32(ang)
                for _ in range(abs(k)):
                    a = ups(a, k)
                for r in rs.uniform(1.0, max(1.5, float(oc.filtering_radius)), 8):
                    pts.append((r * np.cos(np.float64(a)), r * np.sin(np.float64(a)), rs.normal(0, 0.05)))
        # a dense ring of ordinary points so that patches are non-trivial
        base = W.c1_test_cloud(21, n=6000) * np.float32(min(1.0, float(oc.filtering_radius) / 50.0))
        xyz = np.concatenate([np.array(pts, np.float32), base])
        eng = engine_factory(pc, max_points=len(xyz), max_frames=1)
        eng.set_trace(True, 1 << 16)
        labels, res, st = eng.segment(xyz, want_residual=True)
        assert_frame_parity(xyz, labels, res, st, eng.nodes(1 << 16), oc, tag=f"torture {kw}")


def test_randomised_configs_and_clouds(engine_factory):
    """Seeded sweep over configurations and perturbed clouds: any divergence from the oracle in labels, stats,
    residuals or the recursion tree fails. Covers odd sector counts, tiny / huge radii, every threshold knob,
    shallow and deep recursion limits, both seed modes, NaN/inf injection, scaling, rotation, shuffled order."""
    rs = np.random.RandomState(2024)
    gens = [lambda s: W.c1_test_cloud(s, n=int(rs.randint(500, 30000))), lambda s: W.c2_hdl64(s)[:: int(rs.randint(1, 6))],
            lambda s: W.c4_iac_fused(s, rays_per_sensor=20000)]
    for it in range(36):
        xyz = gens[it % 3](int(rs.randint(0, 1000))).copy()
        if rs.rand() < 0.5:
            xyz *= np.float32(rs.choice([0.25, 0.5, 2.0, 3.0]))
        if rs.rand() < 0.5:
            a = np.float32(rs.uniform(0, 6.28)); c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
            xyz = np.column_stack([xyz[:, 0] * c - xyz[:, 1] * s, xyz[:, 0] * s + xyz[:, 1] * c, xyz[:, 2]]).astype(np.float32)
        if rs.rand() < 0.3:
            xyz = xyz[rs.permutation(len(xyz))]
        if rs.rand() < 0.4:
            idx = rs.randint(0, len(xyz), 20)
            xyz[idx, rs.randint(0, 3, 20)] = rs.choice([np.nan, np.inf, -np.inf], 20)
        kw = dict(
            num_sectors=int(rs.choice([1, 2, 3, 7, 10, 16, 31, 37, 64])),
            filtering_radius=float(rs.choice([5.0, 30.0, 50.0, 150.0, 1000.0])),
            max_iter=int(rs.choice([0, 1, 2, 5, 50, 100])),
            max_split_depth=int(rs.choice([0, 1, 3, 10, 1000])),
            th_dist=float(rs.choice([0.02, 0.1, 0.2, 0.5])),
            th_seeds=float(rs.choice([0.05, 0.15, 0.5])),
            sensor_height=float(rs.choice([-2.0, -1.5, 0.0, 1.2, 3.0])),
            adaptive_seed_height=bool(rs.rand() < 0.7),
        )
        pc, oc = cfg_pair(**kw)
        eng = engine_factory(pc, max_points=len(xyz), max_frames=1)
        eng.set_trace(True, max(1 << 16, len(xyz)))
        labels, res, st = eng.segment(xyz, want_residual=True)
        assert_frame_parity(xyz, labels, res, st, eng.nodes(max(1 << 16, len(xyz))), oc, tag=f"case {it} {kw}")
        eng.close()


def test_pointcloud2_records_read_in_place(engine_factory):
    """PointCloud2 wire format (pcl::PointXYZ 16-byte records, Velodyne-style 32-byte records, odd field
    order): the binning kernel reads the records in place; results must equal the packed-float3 path."""
    xyz = W.c2_hdl64(9)[::2].copy()
    xyz[5] = np.nan
    pc, oc = cfg_pair(max_split_depth=3, max_iter=4, th_dist=0.05)        # a few splits, so the post-split residual gather runs too
    eng = engine_factory(pc, max_points=len(xyz), max_frames=1)
    want_l, want_r, want_s = eng.segment(xyz, want_residual=True)
    assert_frame_parity(xyz, want_l, want_r, want_s, None, oc, check_nodes=False, tag="packed")
    n = len(xyz)
    for step, (ox, oy, oz) in ((16, (0, 4, 8)), (32, (0, 4, 8)), (24, (12, 4, 20))):
        rec = np.full((n, step), 0xAB, np.uint8)
        rec[:, ox:ox + 4] = xyz[:, 0:1].copy().view(np.uint8)
        rec[:, oy:oy + 4] = xyz[:, 1:2].copy().view(np.uint8)
        rec[:, oz:oz + 4] = xyz[:, 2:3].copy().view(np.uint8)
        l, r, s = eng.segment_pointcloud2(rec.reshape(-1), n, step, ox, oy, oz, want_residual=True)
        assert np.array_equal(l, want_l) and np.array_equal(r, want_r, equal_nan=True), (step, ox, oy, oz)
        for k in ("n_ground", "n_nodes", "n_splits", "total_iterations"):
            assert s[k] == want_s[k]
    with pytest.raises(rpw_amd.RpwError):
        eng.segment_pointcloud2(np.zeros(64, np.uint8), 4, 16, 0, 4, 14)      # misaligned field


def _fusion_case(rs, kind):
    if kind == 0:      # IAC-style: three sensors, 0 / +120 / -120 degrees (reference lidar_fusion.cpp:20-36)
        clouds = [W.c2_hdl64(s)[:: 3 + s] for s in (11, 12, 13)]
        return clouds, [0.0, 120.0, -120.0], [2.5, 2.5, 2.5]
    if kind == 1:      # no rotation at all, different ego radii, a few non-finite points
        clouds = [W.c1_test_cloud(31, n=8000), W.c1_test_cloud(32, n=3000)]
        clouds[1][7] = np.nan; clouds[0][100, 1] = np.inf
        return clouds, [0.0, 1e-7], [4.0, 0.0]
    if kind == 2:      # one sensor, odd angle, big ego radius
        return [W.c4_iac_fused(3, rays_per_sensor=30000)], [33.3], [6.0]
    clouds = [W.c1_test_cloud(40 + k, n=int(rs.randint(1, 4000))) for k in range(8)]
    clouds[3] = clouds[3][:0]            # an empty sensor cloud
    return clouds, list(rs.uniform(-180, 180, 8)), list(rs.uniform(0, 8, 8))


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_lidar_fusion_prestep_on_device(engine_factory, kind):
    """LidarFusion pre-step (rotation + ego-radius removal + concatenation) fused into the binning kernel:
    labels of the surviving points == oracle run on the oracle-fused cloud; removed points carry LABEL_EGO."""
    rs = np.random.RandomState(77 + kind)
    clouds, rot, ego = _fusion_case(rs, kind)
    fused, src, raw, off = O.fuse(clouds, rot, ego)
    pc, oc = cfg_pair(max_split_depth=3, max_iter=6 if kind == 0 else 100, th_dist=0.05 if kind == 0 else 0.2)
    eng = engine_factory(pc, max_points=len(raw) + 8, max_frames=2)
    eng.set_sensors(rot, ego)
    eng.set_trace(True, 1 << 17)
    labels, res, st = eng.segment_fused(clouds, want_residual=True)
    removed = np.ones(len(raw), bool); removed[src] = False
    assert np.all(labels[removed] == rpw_amd.LABEL_EGO) and not np.any(labels[src] == rpw_amd.LABEL_EGO)
    assert np.all(res[removed] == -1)
    assert_frame_parity(fused, labels[src], res[src], st, eng.nodes(1 << 17), oc, tag=f"fusion kind {kind}")
    # the plain call still works on the same handle, and so does switching the pre-step off
    l2, _, s2 = eng.segment(fused)
    assert np.array_equal(l2, labels[src]) and s2["n_ground"] == st["n_ground"]
    if kind == 0:      # device-resident batch of two fused frames
        import torch
        frames = [raw, raw[::-1].copy()]
        foff = np.array([0, len(raw), 2 * len(raw)], np.uint32)
        so = np.stack([off, np.array([0, len(raw) - off[2], len(raw) - off[1], len(raw)], np.uint32)])
        eng2 = engine_factory(pc, max_points=2 * len(raw), max_frames=2)
        eng2.set_sensors([0.0, 120.0, -120.0], ego)
        d = torch.from_numpy(np.concatenate(frames)).cuda()
        lab = torch.empty(2 * len(raw), dtype=torch.uint8, device="cuda")
        eng2.segment_batch_device_fused(d.data_ptr(), foff, so, lab.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        L = lab.cpu().numpy()
        assert np.array_equal(L[: len(raw)], labels)
        # frame 1: the reversed cloud, sensors in reversed sizes (rotations 0 / 120 / -120 applied to its three parts)
        parts = [frames[1][so[1][k]: so[1][k + 1]] for k in range(3)]
        f1, src1, _, _ = O.fuse(parts, [0.0, 120.0, -120.0], ego)
        ref1 = O.segment(f1, oc, want_nodes=False)
        assert np.array_equal(L[len(raw):][src1], ref1.labels)
    eng.set_sensors([], [])
    l3, _, _ = eng.segment(raw)
    assert np.array_equal(l3, O.segment(raw, oc, want_nodes=False).labels)


def test_huge_frame_and_huge_patches(engine_factory):
    """A 2.3 M-point frame (more tiles than the gather keeps in LDS: global binary-search fallback) cut into few,
    very large patches (num_sectors = 2 -> 16 patches, the largest > 300 k points)."""
    a = W.c5_dense(0)
    ang = np.float32(0.7); c, s = np.float32(np.cos(ang)), np.float32(np.sin(ang))
    b = np.column_stack([a[:, 0] * c - a[:, 1] * s, a[:, 0] * s + a[:, 1] * c, a[:, 2] + np.float32(0.01)]).astype(np.float32)
    xyz = np.concatenate([a, b, a[:200000] * np.float32(0.5)])
    assert len(xyz) > 1024 * 2048
    pc, oc = cfg_pair(num_sectors=2, max_iter=8, max_split_depth=2, th_dist=0.1)
    eng = engine_factory(pc, max_points=len(xyz), max_frames=1)
    eng.set_trace(True, 1 << 16)
    labels, res, st = eng.segment(xyz, want_residual=True)
    ref = assert_frame_parity(xyz, labels, res, st, eng.nodes(1 << 16), oc, tag="huge")
    assert ref.nodes["len"].max() > 300000


def test_maximum_sector_count(engine_factory):
    """num_sectors = 128 (the limit): 1024 patches per frame, histogram rows of 1028 classes; a batch of two frames."""
    import torch
    frames = [W.c2_hdl64(11), W.c1_test_cloud(12)]
    pc, oc = cfg_pair(num_sectors=128, max_split_depth=2)
    off = np.zeros(len(frames) + 1, np.uint32); off[1:] = np.cumsum([len(f) for f in frames])
    eng = engine_factory(pc, max_points=int(off[-1]), max_frames=len(frames))
    d = torch.from_numpy(np.concatenate(frames)).cuda()
    lab = torch.empty(int(off[-1]), dtype=torch.uint8, device="cuda")
    res = torch.empty(int(off[-1]), dtype=torch.float32, device="cuda")
    eng.segment_batch_device(d.data_ptr(), off, lab.data_ptr(), res.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = eng.batch_stats(len(frames))
    L, R = lab.cpu().numpy(), res.cpu().numpy()
    for f, xyz in enumerate(frames):
        sl = slice(int(off[f]), int(off[f + 1]))
        assert_frame_parity(xyz, L[sl], R[sl], {k: st[k][f] for k in st.dtype.names}, None, oc, check_nodes=False, tag=f"S128-frame{f}")


def test_batch_mixing_a_huge_frame_with_small_ones(engine_factory):
    """Frames of more than kGatherTiles tiles take the global run directory (offsets made by scan_offsets), the others
    scan their run lengths in LDS: both kinds in ONE batch, the huge frame in the middle."""
    import torch
    a = W.c5_dense(1)
    huge = np.concatenate([a, a[::-1] * np.float32(0.999), a[:150000] * np.float32(0.7)])
    assert len(huge) > 1024 * 2048
    frames = [W.c1_test_cloud(21), huge, W.c2_hdl64(22)]
    pc, oc = cfg_pair(max_iter=6, max_split_depth=2)
    off = np.zeros(len(frames) + 1, np.uint32); off[1:] = np.cumsum([len(f) for f in frames])
    eng = engine_factory(pc, max_points=int(off[-1]), max_frames=len(frames))
    d = torch.from_numpy(np.concatenate(frames)).cuda()
    lab = torch.empty(int(off[-1]), dtype=torch.uint8, device="cuda")
    eng.segment_batch_device(d.data_ptr(), off, lab.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = eng.batch_stats(len(frames))
    L = lab.cpu().numpy()
    for f, xyz in enumerate(frames):
        sl = slice(int(off[f]), int(off[f + 1]))
        assert_frame_parity(xyz, L[sl], None, {k: st[k][f] for k in st.dtype.names}, None, oc, check_nodes=False, tag=f"mixed-frame{f}")


def test_large_batch_equals_frames_processed_alone(engine_factory):
    """Size-independent property at bench scale: in a batch of 96 HDL-64 frames (12 M points, one launch) every frame
    gets exactly the labels and stats it gets when it is processed alone; three of them are also checked against the
    oracle. Exercises frame_of_tile, per-frame class totals and the size-bucket lists with thousands of patches."""
    import torch
    import bench
    frames = bench.make_frames(96, 0)
    pc, oc = cfg_pair(max_split_depth=3)
    off = np.zeros(len(frames) + 1, np.uint32); off[1:] = np.cumsum([len(f) for f in frames]); N = int(off[-1])
    big = engine_factory(pc, max_points=N, max_frames=len(frames))
    d = torch.from_numpy(np.concatenate(frames)).cuda()
    lab = torch.empty(N, dtype=torch.uint8, device="cuda")
    big.segment_batch_device(d.data_ptr(), off, lab.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = big.batch_stats(len(frames)); L = lab.cpu().numpy()
    one = engine_factory(pc, max_points=max(len(f) for f in frames), max_frames=1)
    for f in range(0, len(frames), 5):
        l1, _, s1 = one.segment(frames[f])
        sl = slice(int(off[f]), int(off[f + 1]))
        assert np.array_equal(L[sl], l1), f"frame {f}: batch labels differ from the frame processed alone"
        for k in ("n_zone", "n_ground", "n_patches", "n_nodes", "n_splits", "total_iterations"):
            assert int(st[k][f]) == int(s1[k]), (f, k)
    for f in (0, 47, 95):
        sl = slice(int(off[f]), int(off[f + 1]))
        assert_frame_parity(frames[f], L[sl], None, {k: st[k][f] for k in st.dtype.names}, None, oc, check_nodes=False, tag=f"batch96-frame{f}")
