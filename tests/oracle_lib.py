"""ctypes binding of the CPU oracle (oracle/librpw_oracle.so) -- test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "librpw_oracle.so")


class OConfig(C.Structure):
    _fields_ = [
        ("sensor_height", C.c_float), ("max_range", C.c_float), ("num_sectors", C.c_int32),
        ("max_iter", C.c_int32), ("adaptive_seed_height", C.c_int32), ("th_seeds", C.c_float),
        ("th_dist", C.c_float), ("th_outlier", C.c_float), ("filtering_radius", C.c_float),
        ("max_split_depth", C.c_int32),
    ]


class ONode(C.Structure):
    _fields_ = [
        ("patch", C.c_int32), ("depth", C.c_int32), ("start", C.c_uint32), ("len", C.c_uint32),
        ("kind", C.c_int32), ("iterations", C.c_int32), ("n_inliers", C.c_uint32),
        ("split_axis", C.c_int32), ("split_value", C.c_float),
        ("centroid", C.c_float * 3), ("normal", C.c_float * 3), ("residual", C.c_float),
    ]


class OStats(C.Structure):
    _fields_ = [
        ("n_in", C.c_uint32), ("n_clean", C.c_uint32), ("n_zone", C.c_uint32), ("n_ground", C.c_uint32),
        ("n_patches", C.c_uint32), ("n_nodes", C.c_uint32), ("n_splits", C.c_uint32),
        ("max_depth", C.c_uint32), ("total_iterations", C.c_uint64),
    ]


class OBenchResult(C.Structure):
    _fields_ = [("single_thread_median_ms", C.c_double), ("single_thread_frames", C.c_uint32), ("threads", C.c_uint32),
                ("hardware_concurrency", C.c_uint32), ("frames_done", C.c_uint64), ("elapsed_s", C.c_double)]


NODE_DTYPE = np.dtype([
    ("patch", "<i4"), ("depth", "<i4"), ("start", "<u4"), ("len", "<u4"), ("kind", "<i4"),
    ("iterations", "<i4"), ("n_inliers", "<u4"), ("split_axis", "<i4"), ("split_value", "<f4"),
    ("centroid", "<f4", 3), ("normal", "<f4", 3), ("residual", "<f4"),
])
assert NODE_DTYPE.itemsize == C.sizeof(ONode)

DEFAULTS = dict(sensor_height=1.2, max_range=150.0, num_sectors=10, max_iter=100,
                adaptive_seed_height=1, th_seeds=0.15, th_dist=0.2, th_outlier=0.08,
                filtering_radius=150.0, max_split_depth=1000)


def make_config(**kw) -> OConfig:
    d = dict(DEFAULTS)
    d.update(kw)
    return OConfig(**d)


_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(ORACLE_DIR, "rpw_oracle.cpp")
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return ORACLE_SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(ORACLE_SO)
        L.rpwo_segment.restype = C.c_int
        L.rpwo_segment.argtypes = [C.POINTER(OConfig), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_uint32, C.POINTER(OStats)]
        L.rpwo_filter_ground_points.restype = C.c_int
        L.rpwo_filter_ground_points.argtypes = [C.POINTER(OConfig), C.c_void_p, C.c_uint32, C.c_void_p,
                                                C.POINTER(C.c_uint32), C.c_void_p, C.POINTER(C.c_uint32)]
        L.rpwo_fuse.restype = C.c_int
        L.rpwo_fuse.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.rpwo_sample_ground_and_obstacles.restype = C.c_int
        L.rpwo_sample_ground_and_obstacles.argtypes = [C.POINTER(OConfig), C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_uint32,
                                                       C.c_void_p, C.POINTER(C.c_uint32)]
        L.rpwo_set_chatter.restype = None
        L.rpwo_set_chatter.argtypes = [C.c_int]
        L.rpwo_bench_threads.restype = C.c_int
        L.rpwo_bench_threads.argtypes = [C.POINTER(OConfig), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_uint32,
                                         C.POINTER(OBenchResult)]
        L.rpwo_eig3.restype = C.c_int
        L.rpwo_eig3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpwo_fit_plane.restype = None
        L.rpwo_fit_plane.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.rpwo_tables.restype = None
        L.rpwo_tables.argtypes = [C.POINTER(OConfig), C.c_void_p, C.c_void_p]
        L.rpwo_angle.restype = C.c_float
        L.rpwo_angle.argtypes = [C.c_float, C.c_float]
        _lib = L
    return _lib


@dataclass
class OracleResult:
    labels: np.ndarray
    residual: np.ndarray | None
    nodes: np.ndarray | None
    stats: dict


def segment(xyz: np.ndarray, cfg: OConfig | None = None, want_residual=True, want_nodes=True,
            nodes_cap: int | None = None) -> OracleResult:
    cfg = cfg or make_config()
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    n = len(xyz)
    labels = np.empty(n, np.uint8)
    res = np.empty(n, np.float32) if want_residual else None
    cap = nodes_cap if nodes_cap is not None else max(4096, n // 8)
    nodes = np.zeros(cap, NODE_DTYPE) if want_nodes else None
    st = OStats()
    rc = lib().rpwo_segment(C.byref(cfg), xyz.ctypes.data, n, labels.ctypes.data,
                            res.ctypes.data if res is not None else None,
                            nodes.ctypes.data if nodes is not None else None,
                            cap if nodes is not None else 0, C.byref(st))
    if rc:
        raise RuntimeError(f"rpwo_segment failed: {rc}")
    stats = {k: getattr(st, k) for k, _ in OStats._fields_}
    if nodes is not None:
        nodes = nodes[: min(cap, st.n_nodes)]
    return OracleResult(labels, res, nodes, stats)


def filter_ground_points(xyz: np.ndarray, cfg: OConfig | None = None):
    cfg = cfg or make_config()
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    n = len(xyz)
    g = np.empty((n, 3), np.float32)
    ng = np.empty((n, 3), np.float32)
    cg, cn = C.c_uint32(0), C.c_uint32(0)
    rc = lib().rpwo_filter_ground_points(C.byref(cfg), xyz.ctypes.data, n, g.ctypes.data, C.byref(cg),
                                         ng.ctypes.data, C.byref(cn))
    if rc:
        raise RuntimeError(f"rpwo_filter_ground_points failed: {rc}")
    return g[: cg.value].copy(), ng[: cn.value].copy()


def eig3(cov: np.ndarray):
    cov = np.ascontiguousarray(cov, dtype=np.float32).reshape(3, 3)
    ev = np.empty(3, np.float32)
    Q = np.empty(9, np.float32)
    info = lib().rpwo_eig3(cov.ctypes.data, ev.ctypes.data, Q.ctypes.data)
    return info, ev, Q.reshape(3, 3).T.copy()  # columns = eigenvectors


def fit_plane(xyz: np.ndarray):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    out = np.empty(7, np.float32)
    lib().rpwo_fit_plane(xyz.ctypes.data, len(xyz), out.ctypes.data)
    return out[:3].copy(), out[3:6].copy(), float(out[6])


def tables(cfg: OConfig):
    e = np.empty(9, np.float32)
    b = np.empty(cfg.num_sectors + 1, np.float32)
    lib().rpwo_tables(C.byref(cfg), e.ctypes.data, b.ctypes.data)
    return e, b


def fuse(clouds, rotation_deg, ego_radius):
    """LidarFusion pre-step on a list of raw sensor clouds -> (fused xyz, index into the concatenated raw cloud)."""
    raw = np.ascontiguousarray(np.concatenate(clouds, axis=0), dtype=np.float32)
    off = np.zeros(len(clouds) + 1, np.uint32); off[1:] = np.cumsum([len(c) for c in clouds])
    rot = np.ascontiguousarray(rotation_deg, dtype=np.float32); ego = np.ascontiguousarray(ego_radius, dtype=np.float32)
    out = np.empty_like(raw); src = np.empty(len(raw), np.uint32); n = C.c_uint32(0)
    rc = lib().rpwo_fuse(raw.ctypes.data, off.ctypes.data, len(clouds), rot.ctypes.data, ego.ctypes.data,
                         out.ctypes.data, src.ctypes.data, C.byref(n))
    assert rc == 0
    return out[: n.value].copy(), src[: n.value].copy(), raw, off


def sample_ground_and_obstacles(xyz: np.ndarray, cfg: OConfig | None = None, target_height=1.1, base_tol=0.5, seed=0):
    """sampleGroundAndObstacles restated (seeded): -> (k, 3) float32."""
    cfg = cfg or make_config()
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    out = np.empty((len(xyz), 3), np.float32)
    k = C.c_uint32(0)
    rc = lib().rpwo_sample_ground_and_obstacles(C.byref(cfg), xyz.ctypes.data, len(xyz), target_height, base_tol, seed,
                                                out.ctypes.data, C.byref(k))
    if rc:
        raise RuntimeError(f"rpwo_sample_ground_and_obstacles failed: {rc}")
    return out[: k.value].copy()


def set_chatter(on: bool):
    lib().rpwo_set_chatter(int(bool(on)))


def bench_threads(frames, cfg: OConfig | None = None, n_threads: int = 0, seconds: float = 5.0, single_reps: int = 24) -> dict:
    """Native CPU-baseline harness (no Python in the timed loops): single-thread median ms/frame, then frames/s with one
    frame per std::thread on n_threads threads (0 = hardware_concurrency)."""
    cfg = cfg or make_config()
    xyz = np.ascontiguousarray(np.concatenate(frames, axis=0), dtype=np.float32)
    off = np.zeros(len(frames) + 1, np.uint32); off[1:] = np.cumsum([len(f) for f in frames])
    r = OBenchResult()
    rc = lib().rpwo_bench_threads(C.byref(cfg), xyz.ctypes.data, off.ctypes.data, len(frames), n_threads, seconds, single_reps, C.byref(r))
    if rc:
        raise RuntimeError(f"rpwo_bench_threads failed: {rc}")
    return {k: getattr(r, k) for k, _ in OBenchResult._fields_}
