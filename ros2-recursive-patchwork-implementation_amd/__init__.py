"""MI355X-native Recursive Patchwork ground segmentation -- Python host side.

This package holds only what the one hot path needs:
  csrc/        hand-written gfx950 HIP kernels + the C-ABI implementation (librpw_hip.so)
  cpp/         the drop-in C++ class recursive_patchwork::RecursivePatchwork over the C-ABI
  workloads.py seeded synthetic LiDAR clouds (BASELINE.json configs)

The Python classes below mirror the reference's C++ interface for this path
(reference include/recursive_patchwork.hpp:25-67): PatchworkConfig field for field,
RecursivePatchwork.filterGroundPoints / cleanPoints / setConfig / getConfig with the same
argument meaning. They call the C-ABI (include/rpw.h) through ctypes. There is NO CPU
fallback: if librpw_hip.so is missing or no MI355X is present the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, asdict

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.environ.get("RPW_LIB", os.path.join(PKG_DIR, "librpw_hip.so"))

LABEL_NONGROUND, LABEL_GROUND, LABEL_DROPPED, LABEL_OUTSIDE, LABEL_EGO = 0, 1, 2, 3, 4
K_BIN_COUNT, K_SCAN, K_BIN_SCATTER, K_FIT, K_COUNT = 0, 1, 2, 3, 4
KERNEL_NAMES = ("bin_count", "scan_offsets", "bin_scatter", "patch_fit")


class RpwError(RuntimeError):
    pass


class CConfig(C.Structure):
    _fields_ = [
        ("sensor_height", C.c_float), ("max_range", C.c_float), ("num_sectors", C.c_int32),
        ("max_iter", C.c_int32), ("adaptive_seed_height", C.c_int32), ("th_seeds", C.c_float),
        ("th_dist", C.c_float), ("th_outlier", C.c_float), ("filtering_radius", C.c_float),
        ("max_split_depth", C.c_int32),
    ]


class CFrameStats(C.Structure):
    _fields_ = [
        ("n_in", C.c_uint32), ("n_clean", C.c_uint32), ("n_zone", C.c_uint32), ("n_ground", C.c_uint32),
        ("n_patches", C.c_uint32), ("n_nodes", C.c_uint32), ("n_splits", C.c_uint32),
        ("max_depth", C.c_uint32), ("total_iterations", C.c_uint64), ("chain_addends", C.c_uint64),
    ]


NODE_DTYPE = np.dtype([
    ("frame", "<i4"), ("patch", "<i4"), ("depth", "<i4"), ("start", "<u4"), ("len", "<u4"), ("kind", "<i4"),
    ("iterations", "<i4"), ("n_inliers", "<u4"), ("split_axis", "<i4"), ("split_value", "<f4"),
    ("centroid", "<f4", 3), ("normal", "<f4", 3), ("residual", "<f4"),
])
STATS_DTYPE = np.dtype([
    ("n_in", "<u4"), ("n_clean", "<u4"), ("n_zone", "<u4"), ("n_ground", "<u4"), ("n_patches", "<u4"),
    ("n_nodes", "<u4"), ("n_splits", "<u4"), ("max_depth", "<u4"), ("total_iterations", "<u8"), ("chain_addends", "<u8"),
])
assert STATS_DTYPE.itemsize == C.sizeof(CFrameStats)


@dataclass
class PatchworkConfig:
    """reference include/recursive_patchwork.hpp:25-36, same names and defaults."""
    sensor_height: float = 1.2
    max_range: float = 150.0
    num_sectors: int = 10
    max_iter: int = 100
    adaptive_seed_height: bool = True
    th_seeds: float = 0.15
    th_dist: float = 0.2
    th_outlier: float = 0.08
    filtering_radius: float = 150.0
    max_split_depth: int = 1000

    def to_c(self) -> CConfig:
        d = asdict(self)
        d["adaptive_seed_height"] = int(bool(d["adaptive_seed_height"]))
        return CConfig(**d)


def build(verbose: bool = False) -> str:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", PKG_DIR, "librpw_hip.so"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RpwError("building librpw_hip.so failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(r.stdout)
    return LIB_PATH


_lib = None


def load_library():
    """dlopen librpw_hip.so; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RpwError(f"{LIB_PATH} not found: run __graft_entry__.build() / make -C {PKG_DIR}. "
                       "There is no CPU fallback for this path.")
    L = C.CDLL(LIB_PATH)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
    L.rpw_abi_version.restype = u32
    L.rpw_default_config.argtypes = [C.POINTER(CConfig)]
    L.rpw_default_config.restype = None
    L.rpw_last_error.argtypes = [vp]
    L.rpw_last_error.restype = C.c_char_p
    L.rpw_create.argtypes = [C.POINTER(CConfig), i32, u32, u32, C.POINTER(vp)]
    L.rpw_create.restype = i32
    L.rpw_destroy.argtypes = [vp]
    L.rpw_destroy.restype = None
    L.rpw_set_config.argtypes = [vp, C.POINTER(CConfig)]
    L.rpw_set_config.restype = i32
    L.rpw_get_config.argtypes = [vp, C.POINTER(CConfig)]
    L.rpw_get_config.restype = i32
    L.rpw_segment.argtypes = [vp, vp, u32, vp, vp, C.POINTER(CFrameStats)]
    L.rpw_segment.restype = i32
    L.rpw_segment_pointcloud2.argtypes = [vp, vp, u32, u32, u32, u32, u32, vp, vp, C.POINTER(CFrameStats)]
    L.rpw_segment_pointcloud2.restype = i32
    L.rpw_set_pipeline.argtypes = [vp, u32]
    L.rpw_set_pipeline.restype = i32
    L.rpw_get_pipeline.argtypes = [vp]
    L.rpw_get_pipeline.restype = u32
    L.rpw_slot_input.argtypes = [vp, u32]
    L.rpw_slot_input.restype = vp
    L.rpw_submit.argtypes = [vp, u32, vp, u32, u32]
    L.rpw_submit.restype = i32
    L.rpw_submit_pointcloud2.argtypes = [vp, u32, vp, u32, u32, u32, u32, u32, u32]
    L.rpw_submit_pointcloud2.restype = i32
    L.rpw_poll.argtypes = [vp, u32, C.POINTER(i32)]
    L.rpw_poll.restype = i32
    L.rpw_wait.argtypes = [vp, u32, vp, vp, C.POINTER(CFrameStats)]
    L.rpw_wait.restype = i32
    L.rpw_set_sensors.argtypes = [vp, u32, vp]
    L.rpw_set_sensors.restype = i32
    L.rpw_segment_fused.argtypes = [vp, vp, vp, vp, vp, C.POINTER(CFrameStats)]
    L.rpw_segment_fused.restype = i32
    L.rpw_segment_batch_device_fused.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp]
    L.rpw_segment_batch_device_fused.restype = i32
    L.rpw_segment_batch_device.argtypes = [vp, vp, vp, u32, vp, vp, vp]
    L.rpw_segment_batch_device.restype = i32
    L.rpw_get_batch_stats.argtypes = [vp, vp, u32]
    L.rpw_get_batch_stats.restype = i32
    L.rpw_result_record_device.argtypes = [vp, vp, u32, vp]
    L.rpw_result_record_device.restype = i32
    L.rpw_set_trace.argtypes = [vp, i32, u32]
    L.rpw_set_trace.restype = i32
    L.rpw_get_nodes.argtypes = [vp, vp, u32, C.POINTER(u32)]
    L.rpw_get_nodes.restype = i32
    L.rpw_set_profiling.argtypes = [vp, i32]
    L.rpw_set_profiling.restype = i32
    L.rpw_get_kernel_ms.argtypes = [vp, vp, vp, i32]
    L.rpw_get_kernel_ms.restype = i32
    L.rpw_debug_angles.argtypes = [vp, vp, vp, vp, u32]
    L.rpw_debug_angles.restype = i32
    L.rpw_debug_eig3.argtypes = [vp, vp, vp, u32]
    L.rpw_debug_eig3.restype = i32
    L.rpw_debug_loop_bench.argtypes = [vp, i32, u32, u32, i32, u32, C.POINTER(C.c_double)]
    L.rpw_debug_loop_bench.restype = i32
    L.rpw_debug_loop_bench_ticks.argtypes = [vp, i32, u32, u32, i32, u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.rpw_debug_loop_bench_ticks.restype = i32
    L.rpw_debug_angles_host.argtypes = [vp, vp, vp, u32]
    L.rpw_debug_angles_host.restype = None
    _lib = L
    return L


EXPORTED_SYMBOLS = (
    "rpw_default_config", "rpw_abi_version", "rpw_create", "rpw_destroy", "rpw_set_config", "rpw_get_config",
    "rpw_last_error", "rpw_segment", "rpw_segment_pointcloud2", "rpw_set_pipeline", "rpw_get_pipeline", "rpw_slot_input",
    "rpw_submit", "rpw_submit_pointcloud2", "rpw_poll", "rpw_wait", "rpw_set_sensors", "rpw_segment_fused",
    "rpw_segment_batch_device_fused", "rpw_segment_batch_device", "rpw_get_batch_stats", "rpw_set_trace",
    "rpw_get_nodes", "rpw_set_profiling", "rpw_get_kernel_ms", "rpw_debug_angles", "rpw_debug_eig3",
    "rpw_debug_angles_host", "rpw_debug_loop_bench", "rpw_debug_loop_bench_ticks", "rpw_result_record_device",
)


class Engine:
    """Thin owner of one rpw_handle (one GPU, one stream at a time)."""

    def __init__(self, config: PatchworkConfig | None = None, device: int = 0,
                 max_points: int = 1 << 21, max_frames: int = 1):
        self._L = load_library()
        self._h = C.c_void_p()
        self.config = config or PatchworkConfig()
        cc = self.config.to_c()
        st = self._L.rpw_create(C.byref(cc), device, max_points, max_frames, C.byref(self._h))
        if st != 0:
            msg = self._L.rpw_last_error(None).decode()
            self._h = C.c_void_p()
            raise RpwError(f"rpw_create failed ({st}): {msg}")
        self.max_points, self.max_frames, self.device = max_points, max_frames, device
        self._slot_n, self._slot_res = [], []          # pipeline slots (set_pipeline)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.rpw_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != 0:
            raise RpwError(f"rpw call failed ({st}): {self._L.rpw_last_error(self._h).decode()}")

    def set_config(self, config: PatchworkConfig):
        cc = config.to_c()
        self._check(self._L.rpw_set_config(self._h, C.byref(cc)))
        self.config = config

    def set_trace(self, enable: bool, max_nodes: int = 1 << 16):
        self._check(self._L.rpw_set_trace(self._h, int(enable), max_nodes))

    def set_profiling(self, enable: bool):
        self._check(self._L.rpw_set_profiling(self._h, int(enable)))

    def kernel_ms(self, reset: bool = False):
        ms = (C.c_double * K_COUNT)()
        cnt = (C.c_uint64 * K_COUNT)()
        self._check(self._L.rpw_get_kernel_ms(self._h, ms, cnt, int(reset)))
        return {KERNEL_NAMES[k]: (ms[k], cnt[k]) for k in range(K_COUNT)}

    def segment(self, xyz: np.ndarray, want_residual: bool = False):
        """One frame from host memory -> (labels uint8[n], residual float32[n] | None, stats dict)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        n = len(xyz)
        labels = np.empty(n, np.uint8)
        res = np.empty(n, np.float32) if want_residual else None
        st = CFrameStats()
        self._check(self._L.rpw_segment(self._h, xyz.ctypes.data, n, labels.ctypes.data,
                                        res.ctypes.data if res is not None else None, C.byref(st)))
        return labels, res, {k: getattr(st, k) for k, _ in CFrameStats._fields_}

    def segment_pointcloud2(self, data: np.ndarray, n: int, point_step: int, off_x: int, off_y: int, off_z: int,
                            want_residual: bool = False):
        """One frame given as PointCloud2 message bytes (uint8 array of n * point_step bytes)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        labels = np.empty(n, np.uint8)
        res = np.empty(n, np.float32) if want_residual else None
        st = CFrameStats()
        self._check(self._L.rpw_segment_pointcloud2(self._h, data.ctypes.data, n, point_step, off_x, off_y, off_z,
                                                    labels.ctypes.data, res.ctypes.data if res is not None else None,
                                                    C.byref(st)))
        return labels, res, {k: getattr(st, k) for k, _ in CFrameStats._fields_}

    # ---- host-pointer pipeline (rpw_submit / rpw_wait): K frames in flight, one slot each ----
    def set_pipeline(self, n_slots: int):
        self._check(self._L.rpw_set_pipeline(self._h, n_slots))
        self._slot_n = [0] * n_slots
        self._slot_res = [False] * n_slots

    def _slot_check(self, slot: int):
        if not 0 <= int(slot) < len(self._slot_n):
            raise RpwError(f"slot {slot} does not exist (set_pipeline made {len(self._slot_n)})")

    def slot_input(self, slot: int, n: int) -> np.ndarray:
        """The slot's pinned input buffer as an (n, 3) float32 view: fill it and pass it to submit() to skip the host copy."""
        ptr = self._L.rpw_slot_input(self._h, slot)
        if not ptr:
            raise RpwError(f"slot {slot} does not exist")
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n, 3))

    def submit(self, slot: int, xyz: np.ndarray, want_residual: bool = False):
        """Enqueue one frame (host memory) on a pipeline slot; returns at once."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        self._slot_check(slot)
        self._check(self._L.rpw_submit(self._h, slot, xyz.ctypes.data, len(xyz), 1 if want_residual else 0))
        self._slot_n[slot] = len(xyz)
        self._slot_res[slot] = want_residual

    def submit_pointcloud2(self, slot: int, data: np.ndarray, n: int, point_step: int, off_x: int, off_y: int, off_z: int,
                           want_residual: bool = False):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        self._slot_check(slot)
        if data.size < int(n) * int(point_step):
            raise ValueError(f"submit_pointcloud2: {n} records of {point_step} bytes need {int(n) * int(point_step)} bytes, data holds {data.size}")
        self._check(self._L.rpw_submit_pointcloud2(self._h, slot, data.ctypes.data, n, point_step, off_x, off_y, off_z,
                                                   1 if want_residual else 0))
        self._slot_n[slot] = n
        self._slot_res[slot] = want_residual

    def poll(self, slot: int) -> bool:
        done = C.c_int(0)
        self._check(self._L.rpw_poll(self._h, slot, C.byref(done)))
        return bool(done.value)

    def wait(self, slot: int, labels: np.ndarray | None = None):
        """Block until the slot's frame is done -> (labels, residual | None, stats dict); frees the slot."""
        self._slot_check(slot)
        n = self._slot_n[slot]
        if labels is None:
            labels = np.empty(n, np.uint8)
        elif not (isinstance(labels, np.ndarray) and labels.dtype == np.uint8 and labels.flags.c_contiguous
                  and labels.flags.writeable and labels.size >= n):
            # rpw_wait copies n bytes to the pointer it is given: anything else than a writable C-contiguous uint8 array
            # of at least n elements would be written past its end (or not where the caller looks)
            raise ValueError(f"wait: labels must be a writable C-contiguous uint8 array of at least {n} elements")
        res = np.empty(n, np.float32) if self._slot_res[slot] else None
        st = CFrameStats()
        self._check(self._L.rpw_wait(self._h, slot, labels.ctypes.data, res.ctypes.data if res is not None else None,
                                     C.byref(st)))
        return labels, res, {k: getattr(st, k) for k, _ in CFrameStats._fields_}

    def set_sensors(self, rotation_deg, ego_radius):
        """LidarFusion pre-step: one (rotation_angle [deg], ego_radius) pair per sensor; empty lists switch it off."""
        cfg = np.ascontiguousarray(np.column_stack([np.asarray(rotation_deg, np.float32), np.asarray(ego_radius, np.float32)]).reshape(-1, 2), np.float32)
        self._check(self._L.rpw_set_sensors(self._h, len(cfg), cfg.ctypes.data if len(cfg) else None))
        self.n_sensors = len(cfg)

    def segment_fused(self, clouds, want_residual: bool = False):
        """One frame = raw clouds of the configured sensors -> (labels per RAW point, residual, stats)."""
        raw = np.ascontiguousarray(np.concatenate([np.asarray(c, np.float32).reshape(-1, 3) for c in clouds], axis=0))
        off = np.zeros(len(clouds) + 1, np.uint32); off[1:] = np.cumsum([len(c) for c in clouds])
        n = len(raw)
        labels = np.empty(n, np.uint8)
        res = np.empty(n, np.float32) if want_residual else None
        st = CFrameStats()
        self._check(self._L.rpw_segment_fused(self._h, raw.ctypes.data, off.ctypes.data, labels.ctypes.data,
                                              res.ctypes.data if res is not None else None, C.byref(st)))
        return labels, res, {k: getattr(st, k) for k, _ in CFrameStats._fields_}

    def segment_batch_device_fused(self, d_xyz_ptr: int, frame_offsets, sensor_offsets, d_labels_ptr: int,
                                   d_residual_ptr: int | None = None, stream: int | None = None):
        off = np.ascontiguousarray(frame_offsets, dtype=np.uint32)
        so = np.ascontiguousarray(sensor_offsets, dtype=np.uint32)
        self._check(self._L.rpw_segment_batch_device_fused(self._h, d_xyz_ptr, off.ctypes.data, so.ctypes.data, len(off) - 1,
                                                           d_labels_ptr, d_residual_ptr or None, stream or None))

    def segment_batch_device(self, d_xyz_ptr: int, frame_offsets: np.ndarray, d_labels_ptr: int,
                             d_residual_ptr: int | None = None, stream: int | None = None):
        """Device-resident batch (raw device pointers, e.g. torch tensor .data_ptr()); asynchronous."""
        off = np.ascontiguousarray(frame_offsets, dtype=np.uint32)
        self._check(self._L.rpw_segment_batch_device(self._h, d_xyz_ptr, off.ctypes.data, len(off) - 1,
                                                     d_labels_ptr, d_residual_ptr or None, stream or None))

    def result_record_device(self, d_out_ptr: int, n_frames: int, stream: int | None = None):
        """int32[4 * n_frames] {ground, non-ground (zone + beyond the radius), patches, refits} per frame of the last batch, written on the device."""
        self._check(self._L.rpw_result_record_device(self._h, d_out_ptr, n_frames, stream or None))

    def batch_stats(self, n_frames: int) -> np.ndarray:
        out = np.zeros(n_frames, STATS_DTYPE)
        self._check(self._L.rpw_get_batch_stats(self._h, out.ctypes.data, n_frames))
        return out

    def nodes(self, cap: int = 1 << 16) -> np.ndarray:
        out = np.zeros(cap, NODE_DTYPE)
        cnt = C.c_uint32(0)
        self._check(self._L.rpw_get_nodes(self._h, out.ctypes.data, cap, C.byref(cnt)))
        if cnt.value > cap:
            raise RpwError(f"trace overflow: {cnt.value} nodes > cap {cap}")
        return out[: cnt.value]

    def loop_bench(self, what: int, n: int, inlier_pct: int = 56, reps: int = 20, blocks: int = 1) -> float:
        """ms of `reps` passes of one of patch_fit's loops over synthetic nodes (see rpw_debug_loop_bench)."""
        ms = C.c_double(0.0)
        self._check(self._L.rpw_debug_loop_bench(self._h, what, n, inlier_pct, reps, blocks, C.byref(ms)))
        return ms.value

    def loop_bench_ticks(self, what: int, n: int, inlier_pct: int = 56, reps: int = 20, blocks: int = 1):
        """-> (ms, shader-clock ticks node 0's owner wave spent in its passes): cycles compare lone-patch loops, whatever the chip's clock."""
        ms, ticks = C.c_double(0.0), C.c_double(0.0)
        self._check(self._L.rpw_debug_loop_bench_ticks(self._h, what, n, inlier_pct, reps, blocks, C.byref(ms), C.byref(ticks)))
        return ms.value, ticks.value

    def debug_angles(self, x, y):
        x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
        a = np.empty_like(x)
        self._check(self._L.rpw_debug_angles(self._h, x.ctypes.data, y.ctypes.data, a.ctypes.data, x.size))
        return a

    def debug_eig3(self, cov):
        cov = np.ascontiguousarray(cov, np.float32).reshape(-1, 9)
        out = np.empty((len(cov), 3), np.float32)
        self._check(self._L.rpw_debug_eig3(self._h, cov.ctypes.data, out.ctypes.data, len(cov)))
        return out


def debug_angles_host(x, y):
    L = load_library()
    x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
    a = np.empty_like(x)
    L.rpw_debug_angles_host(x.ctypes.data, y.ctypes.data, a.ctypes.data, x.size)
    return a


def split_by_labels(xyz: np.ndarray, labels: np.ndarray, n_zone: int):
    """Rebuild the reference's return value {ground, non_ground} (recursive_patchwork.cpp:339-341,402-425)."""
    xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    if n_zone < 3:
        return xyz[:0].copy(), xyz[labels != LABEL_DROPPED].copy()
    g = xyz[labels == LABEL_GROUND]
    ng = np.concatenate([xyz[labels == LABEL_NONGROUND], xyz[labels == LABEL_OUTSIDE]], axis=0)
    return g, ng


class RecursivePatchwork:
    """Python mirror of recursive_patchwork::RecursivePatchwork for this path
    (reference include/recursive_patchwork.hpp:47-67): same method names and meaning."""

    def __init__(self, config: PatchworkConfig | None = None, device: int = 0, max_points: int = 1 << 21):
        self._engine = Engine(config or PatchworkConfig(), device=device, max_points=max_points, max_frames=1)

    def setConfig(self, config: PatchworkConfig):
        self._engine.set_config(config)

    def getConfig(self) -> PatchworkConfig:
        return self._engine.config

    def cleanPoints(self, points: np.ndarray) -> np.ndarray:
        p = np.asarray(points, np.float32).reshape(-1, 3)
        return p[np.isfinite(p).all(axis=1)]

    def filterGroundPoints(self, points: np.ndarray):
        """-> (ground_points, non_ground_points), reference order."""
        p = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        labels, _, st = self._engine.segment(p)
        return split_by_labels(p, labels, st["n_zone"])

    def segment(self, points: np.ndarray, want_residual: bool = False):
        """Label array view of the same call (build-side artefact)."""
        return self._engine.segment(points, want_residual)


class OrderedPipeline:
    """Frames in, (labels, residual | None, stats) out IN SUBMISSION ORDER, with up to `slots` frames on the GPU -- the Python
    twin of cpp/ordered_pipeline.hpp. Waiting on the slots in submission order runs at slots / (latency of the slowest
    frame); here a finished slot is emptied into a reorder buffer and refilled at once, and results leave the buffer in
    order: same slots, same order, the throughput of the any-order loop (bench.py, host_path.in_order_reorder_buffer).
    `engine` needs set_pipeline / submit / poll / wait (rpw_amd.Engine; the CPU test uses a scripted stand-in)."""

    def __init__(self, engine, slots: int = 8, max_held: int | None = None, want_residual: bool = False):
        self._e = engine
        self._e.set_pipeline(slots)
        self._seq = [None] * slots                      # slot -> sequence number of the frame it holds
        self._held = {}                                 # finished, not yet deliverable
        self._next_in = 0
        self._next_out = 0
        self._cap = max_held if max_held else 4 * slots
        self._res = want_residual
        self.max_held = 0

    def frames_on_gpu(self) -> int:
        return sum(q is not None for q in self._seq)

    def _collect(self):
        for s, q in enumerate(self._seq):
            if q is not None and (q == self._next_out or len(self._held) < self._cap) and self._e.poll(s):
                self._held[q] = self._e.wait(s)
                self._seq[s] = None

    def _deliver(self, out):
        while self._next_out in self._held:
            out.append(self._held.pop(self._next_out))
            self._next_out += 1
        self.max_held = max(self.max_held, len(self._held))

    def push(self, xyz) -> list:
        """Submit one frame (blocks only while every slot is busy); returns the results that became deliverable, oldest first."""
        import time
        out = []
        self._collect()
        self._deliver(out)
        while len(self._held) >= self._cap:             # full of frames queued behind the oldest one: wait for that one
            s = self._seq.index(self._next_out)
            self._held[self._next_out] = self._e.wait(s)
            self._seq[s] = None
            self._deliver(out)
        while None not in self._seq:                    # every slot busy
            time.sleep(20e-6)
            self._collect()
            self._deliver(out)
        s = self._seq.index(None)
        self._e.submit(s, xyz, want_residual=self._res)
        self._seq[s] = self._next_in
        self._next_in += 1
        self._deliver(out)
        return out

    def flush(self) -> list:
        """Wait for everything in flight; the remaining results in order."""
        out = []
        for s, q in enumerate(self._seq):
            if q is not None:
                self._held[q] = self._e.wait(s)
                self._seq[s] = None
        self._deliver(out)
        return out
