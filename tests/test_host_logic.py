"""CPU tests of the host logic that sits between the C-ABI and the reference's vector interface,
of the workload generators, and of bench.py's multi-GPU plumbing (gloo, world_size 2)."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
import rpw_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_by_labels_reproduces_reference_order():
    cfg = O.make_config()
    xyz = rpw_amd.workloads.c1_test_cloud(seed=9, n=4000)
    xyz = np.concatenate([xyz, np.array([[400, 0, 0], [np.nan, 0, 0], [0.2, 0.2, 0]], np.float32), xyz[:50] * 10])
    r = O.segment(xyz, cfg)
    g, ng = rpw_amd.split_by_labels(xyz, r.labels, r.stats["n_zone"])
    g2, ng2 = O.filter_ground_points(xyz, cfg)
    assert np.array_equal(g, g2) and np.array_equal(ng, ng2)
    # zone < 3: {{}, cleaned}
    tiny = np.array([[300, 0, 0], [2, 2, 0], [np.nan, 1, 1]], np.float32)
    r = O.segment(tiny, cfg)
    g, ng = rpw_amd.split_by_labels(tiny, r.labels, r.stats["n_zone"])
    assert len(g) == 0 and np.array_equal(ng, tiny[:2])


def test_workloads_are_deterministic_and_shaped():
    W = rpw_amd.workloads
    a, b = W.c2_hdl64(0), W.c2_hdl64(0)
    assert np.array_equal(a, b) and a.dtype == np.float32 and 120000 < len(a) < 135000
    assert not np.array_equal(W.c2_hdl64(1)[:1000], a[:1000])
    c4 = W.c4_iac_fused(0)
    d = np.hypot(c4[:, 0], c4[:, 1])
    assert 280000 < len(c4) < 310000 and d.min() > 2.5        # ego-radius removal, lidar_fusion.cpp:184-187
    assert 19000 < len(W.c1_test_cloud(1)) <= 20000


def test_config_mirror_field_for_field():
    c = rpw_amd.PatchworkConfig(num_sectors=7, adaptive_seed_height=False).to_c()
    o = O.make_config(num_sectors=7, adaptive_seed_height=0)
    for name, _ in rpw_amd.CConfig._fields_:
        assert getattr(c, name) == getattr(o, name), name


def _stub_record(rank, frames):
    """What rpw_result_record_device writes for a batch, from a stub segmenter (no GPU here): four int32 per frame.
    The values depend on the rank AND on the frame, so a gather that mixes up ranks or frames cannot pass."""
    import torch
    rec = torch.empty(4 * len(frames), dtype=torch.int32)
    for f, xyz in enumerate(frames):
        n_ground = int((xyz[:, 2] < -1.5).sum())              # "ground" of the stub: low points
        rec[4 * f: 4 * f + 4] = torch.tensor([n_ground, len(xyz) - n_ground, 50 + f + 10 * rank, 400 + 7 * f + rank], dtype=torch.int32)
    return rec


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = bench.make_frames(3, rank)                      # each rank: its own frames (weak scaling)
    rec = _stub_record(rank, frames)
    out = bench.gather_records(rec, world)                   # the result gather of one finished batch
    t = bench.max_over_ranks(0.5 + rank, torch.device("cpu"))
    # the optional label slabs: the stub's labels (1 = "ground" of _stub_record), ranks of different sizes -> padded slabs
    import numpy as np
    lab = torch.from_numpy(np.concatenate([(f[:, 2] < -1.5).astype(np.uint8) for f in frames])[: 5000 + 1000 * rank].copy())
    slabs, slab_bytes = bench.gather_label_slabs(lab, world)
    q.put((rank, out.tolist(), rec.tolist(), t, [int(rpw_amd.workloads.crc32(f)) for f in frames],
           (slab_bytes, [int((slabs[r] == 1).sum()) for r in range(world)], [int((slabs[r] == 0xFF).sum()) for r in range(world)], int(lab.sum()))))
    dist.barrier()
    dist.destroy_process_group()


def test_multi_gpu_result_gather_gloo_world2():
    """Every rank ends with every rank's per-frame record -- in particular the n_ground of the frames it did NOT process --
    in rank order, and world_size ranks contributed."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    world = 2
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, g0, own0, t0, c0, s0), (r1, g1, own1, t1, c1, s1) = res
    # label slabs: padded to the larger rank, every rank sees both ranks' labels, the padding is 0xFF
    assert s0[0] == s1[0] == 6000 and s0[1] == s1[1] == [s0[3], s1[3]] and s0[2] == s1[2] == [1000, 0]
    assert (r0, r1) == (0, 1)
    assert g0 == g1 and len(g0) == world * 3 * 4             # every rank sees every frame's record
    assert g0[:12] == own0 and g0[12:] == own1               # rank order; rank 0 holds rank 1's n_ground and vice versa
    assert own0 != own1                                      # the two ranks did contribute different records
    n_ground_other = g0[12::4]
    assert all(v > 0 for v in n_ground_other)
    assert t0 == t1 == 1.5                                   # MAX over ranks
    assert c0 != c1                                          # ranks process different frames


def test_pointcloud2_io_cpp(tmp_path):
    """The ROS-free PointCloud2 <-> Point3D conversions used by the ROS2 node source (cpp/ros2/)."""
    import subprocess
    exe = str(tmp_path / "pc2")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(rpw_amd.PKG_DIR, "cpp"),
                           os.path.join(ROOT, "tests", "cpp", "test_pointcloud2_io.cpp"), "-o", exe])
    assert "pointcloud2_io OK" in subprocess.check_output([exe], text=True)


class _ScriptedEngine:
    """Stand-in for rpw_amd.Engine's pipeline calls: a frame is 'done' after a scripted number of polls of its slot."""

    def __init__(self):
        self.rs = np.random.RandomState(11)
        self.slots = []
        self.max_busy = 0

    def set_pipeline(self, n):
        self.slots = [None] * n

    def submit(self, s, xyz, want_residual=False):
        assert self.slots[s] is None
        self.slots[s] = [int(self.rs.randint(0, 23)), float(xyz[0][0])]
        self.max_busy = max(self.max_busy, sum(x is not None for x in self.slots))

    def poll(self, s):
        self.slots[s][0] -= 1
        return self.slots[s][0] < 0

    def wait(self, s):
        tag = self.slots[s][1]
        self.slots[s] = None
        return (np.array([tag]), None, {})


def test_ordered_pipeline_python():
    """rpw_amd.OrderedPipeline: results in submission order although the slots finish in any order, never more frames in
    flight than slots, reorder buffer capped."""
    for slots, cap in ((1, None), (3, None), (8, None), (4, 6)):
        eng = _ScriptedEngine()
        pipe = rpw_amd.OrderedPipeline(eng, slots=slots, max_held=cap)
        got = []
        for k in range(400):
            for labels, _, _ in pipe.push(np.array([[float(k), 0.0, 0.0]], np.float32)):
                got.append(labels[0])
            assert pipe.frames_on_gpu() <= slots
        got += [r[0][0] for r in pipe.flush()]
        assert got == [float(k) for k in range(400)]
        assert eng.max_busy <= slots and pipe.frames_on_gpu() == 0
        assert pipe.max_held <= (cap if cap else 4 * slots)


def test_ordered_pipeline_cpp(tmp_path):
    """The reorder logic of cpp/ordered_pipeline.hpp (results in submission order although the slots finish in any order,
    never more frames on the GPU than slots, a capped reorder buffer) against a scripted stand-in for the engine."""
    import subprocess
    exe = str(tmp_path / "ordered")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(rpw_amd.PKG_DIR, "cpp"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_ordered_pipeline.cpp"), "-o", exe])
    assert "ordered_pipeline OK" in subprocess.check_output([exe], text=True, timeout=120)


def test_node_frame_pipeline_cpp(tmp_path):
    """cpp/frame_pipeline.hpp, the ROS2 node's frame pipeline, on a scripted engine with RecursivePatchwork's capacity rule
    (no growing while frames are in flight): a short first scan followed by full scans, a larger scan into a full pipeline
    and a failing wait must neither lose nor reorder the other frames."""
    import subprocess
    exe = str(tmp_path / "frame_pipeline")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(rpw_amd.PKG_DIR, "cpp"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_frame_pipeline.cpp"), "-o", exe])
    assert "frame_pipeline OK" in subprocess.check_output([exe], text=True, timeout=120)
    node_src = open(os.path.join(rpw_amd.PKG_DIR, "cpp", "ros2", "recursive_patchwork_node.cpp")).read()
    assert "BasicFramePipeline<RecursivePatchwork" in node_src and "pipeline_->push(" in node_src      # the node uses it


def test_exact_block_sum_prototype_matches_sequential_sums(tmp_path):
    """tools/proto/exact_block_sum.cpp (groundwork, not product code): the parallel evaluation of a sequential binary32
    sum -- integer prefix sum inside one binade, ties resolved in order, restart behind a binade change -- must equal the
    real sequential sum bit for bit on its whole adversarial data set (the program exits non-zero on a mismatch)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "exact_block_sum"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-o", str(exe), os.path.join(root, "tools", "proto", "exact_block_sum.cpp")],
                   check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert "MISMATCHES 0" in out and "resume strategy mismatches: 0" in out


def test_ros2_package_glue(tmp_path):
    """The ament package beside the node source: CMake configures WITHOUT ROS2 (rclcpp is found QUIETly and the node target is
    left out), the launch file is valid Python declaring the reference's arguments with the reference's defaults
    (reference launch/recursive_patchwork.launch.py:11-50), package.xml names the package like the reference."""
    import ast
    import subprocess
    import xml.etree.ElementTree as ET
    ros2 = os.path.join(rpw_amd.PKG_DIR, "cpp", "ros2")
    out = subprocess.run(["cmake", "-S", ros2, "-B", str(tmp_path / "b")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "building the drop-in library only" in out.stdout or "ROS2 found" in out.stdout
    src = open(os.path.join(ros2, "launch", "recursive_patchwork.launch.py")).read()
    tree = ast.parse(src)
    args = None
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") == "ARGUMENTS":
            args = ast.literal_eval(node.value)
    want = {"input_topic": "/lidar/points", "ground_topic": "/patchwork/ground", "obstacles_topic": "/patchwork/obstacles",
            "visualization_topic": "/patchwork/visualization", "min_points": "100", "max_iterations": "50",
            "distance_threshold": "0.1", "angle_threshold": "0.1"}
    got = {a[0]: a[1] for a in args}
    for k, v in want.items():
        assert got[k] == v, k
    pkg = ET.parse(os.path.join(ros2, "package.xml")).getroot()
    assert pkg.find("name").text == "recursive_patchwork"
    deps = {d.text for d in pkg.findall("depend")}
    assert {"rclcpp", "sensor_msgs", "visualization_msgs"} <= deps
    node_src = open(os.path.join(ros2, "recursive_patchwork_node.cpp")).read()
    for name in ("input_topic", "ground_topic", "obstacles_topic", "visualization_topic", "min_points", "max_iterations",
                 "distance_threshold", "angle_threshold", "messageFits", "Received empty point cloud"):
        assert name in node_src, name
