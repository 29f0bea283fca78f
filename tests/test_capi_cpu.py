"""CPU tests of the boundary: the C-ABI library loads without a GPU, exports every symbol that
include/rpw.h declares, fails LOUDLY (no CPU fallback) when no device exists, and its host-side
hooks match libm. No compute call is made here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import rpw_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "rpw.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rpw_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    rpw_amd.build()
    lib = rpw_amd.load_library()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/rpw.h but not exported"
    assert sorted(rpw_amd.EXPORTED_SYMBOLS) == syms
    out = subprocess.check_output(["nm", "-D", "--defined-only", rpw_amd.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(syms) <= exported
    assert lib.rpw_abi_version() == 3


def test_no_oracle_linked_into_product():
    """The product must not route through the oracle: no oracle symbol or dependency in librpw_hip.so."""
    out = subprocess.check_output(["nm", "-D", rpw_amd.LIB_PATH], text=True)
    assert "rpwo_" not in out
    ldd = subprocess.check_output(["ldd", rpw_amd.LIB_PATH], text=True)
    assert "rpw_oracle" not in ldd
    for root, _, files in os.walk(rpw_amd.PKG_DIR):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(root, f), errors="ignore").read()
                assert "oracle_lib" not in src and "rpw_oracle" not in src, f"{f} references the oracle"


def test_default_config_matches_reference_defaults():
    lib = rpw_amd.load_library()
    c = rpw_amd.CConfig()
    lib.rpw_default_config(C.byref(c))
    want = rpw_amd.PatchworkConfig()      # reference include/recursive_patchwork.hpp:25-36
    assert (c.sensor_height, c.max_range, c.num_sectors, c.max_iter) == (np.float32(1.2), 150.0, 10, 100)
    assert (c.adaptive_seed_height, c.th_seeds, c.th_dist, c.th_outlier) == (1, np.float32(0.15), np.float32(0.2), np.float32(0.08))
    assert (c.filtering_radius, c.max_split_depth) == (150.0, 1000)
    assert want.num_sectors == 10 and want.max_iter == 100 and want.max_split_depth == 1000


def test_fails_loudly_without_device(has_gpu):
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(rpw_amd.RpwError) as ei:
        rpw_amd.Engine()
    assert "no CPU fallback" in str(ei.value) or "no HIP device" in str(ei.value)
    with pytest.raises(rpw_amd.RpwError):
        rpw_amd.RecursivePatchwork()


def test_invalid_config_rejected_before_touching_a_device():
    lib = rpw_amd.load_library()
    h = C.c_void_p()
    bad = rpw_amd.PatchworkConfig(num_sectors=0).to_c()
    assert lib.rpw_create(C.byref(bad), 0, 1024, 1, C.byref(h)) == 1          # RPW_ERR_INVALID
    assert b"num_sectors" in lib.rpw_last_error(None)
    big = rpw_amd.PatchworkConfig(num_sectors=1000).to_c()
    assert lib.rpw_create(C.byref(big), 0, 1024, 1, C.byref(h)) == 5          # RPW_ERR_UNSUPPORTED


def test_atan2f_model_is_libm_bit_for_bit():
    """The restatement of glibc's atan2f used on the device (csrc/rpw_math.h), evaluated on the host,
    against libm itself: random, near-axis, huge/tiny ratio and signed-zero inputs."""
    libm = C.CDLL("libm.so.6")
    libm.atan2f.restype = C.c_float
    libm.atan2f.argtypes = [C.c_float, C.c_float]
    rs = np.random.RandomState(0)
    n = 3_000_000
    x = rs.uniform(-150, 150, n).astype(np.float32)
    y = rs.uniform(-150, 150, n).astype(np.float32)
    # structured: sector boundaries of S = 10 and S = 8, axes, tiny and huge ratios, signed zeros
    ang = np.concatenate([np.arange(11) * np.float32(2 * np.pi / 10), np.arange(9) * np.float32(2 * np.pi / 8)])
    r = rs.uniform(1, 150, (len(ang), 2000)).astype(np.float32)
    bx = (r * np.cos(ang)[:, None]).astype(np.float32).ravel()
    by = (r * np.sin(ang)[:, None]).astype(np.float32).ravel()
    sx = np.array([0.0, -0.0, 1.0, -1.0, 1e-30, -1e-30, 1e30, 1.0, 1.0, 3e-39, 1.0], np.float32)
    sy = np.array([0.0, 0.0, -0.0, 0.0, 1.0, 1.0, 1e-30, 1e-38, -1e-38, 1.0, 1.0], np.float32)
    X = np.concatenate([x, bx, np.nextafter(bx, np.float32(1e9)), sx, sy])
    Y = np.concatenate([y, by, by, sy, sx])
    got = rpw_amd.debug_angles_host(X, Y)
    # reference: libm atan2f, then the reference's wrap (cuda_interface.cu:625-626)
    step = 200003
    idx = np.unique(np.concatenate([np.arange(0, len(X), max(1, len(X) // step)), np.arange(len(x), len(X))]))
    for i in idx:
        a = np.float32(libm.atan2f(float(Y[i]), float(X[i])))
        if a < 0:
            a = np.float32(np.float64(a) + 6.283185307179586)
        assert a.view(np.uint32) == got[i].view(np.uint32) or (a == 0 and got[i] == 0), (i, X[i], Y[i], a, got[i])


def test_capacity_limit_is_checked_before_the_device_is_touched():
    """max_points above 2^30 is refused (32-bit byte offsets inside the kernels) BEFORE any device query: also here, without a GPU."""
    lib = rpw_amd.load_library()
    h = C.c_void_p()
    c = rpw_amd.PatchworkConfig().to_c()
    st = lib.rpw_create(C.byref(c), 0, (1 << 30) + 1, 1, C.byref(h))
    assert st == 4 and not h.value                                              # RPW_ERR_CAPACITY
    assert b"1073741824" in lib.rpw_last_error(None)


def test_pipeline_entry_points_reject_a_null_handle():
    """rpw_submit / rpw_wait / rpw_poll / rpw_set_pipeline: status codes, no crash, without a device."""
    lib = rpw_amd.load_library()
    done = C.c_int(0)
    assert lib.rpw_set_pipeline(None, 4) == 1                                   # RPW_ERR_INVALID
    assert lib.rpw_submit(None, 0, None, 0, 0) == 1
    assert lib.rpw_submit_pointcloud2(None, 0, None, 0, 16, 0, 4, 8, 0) == 1
    assert lib.rpw_poll(None, 0, C.byref(done)) == 1
    assert lib.rpw_wait(None, 0, None, None, None) == 1
    assert lib.rpw_get_pipeline(None) == 0
    assert not lib.rpw_slot_input(None, 0)


def test_library_sets_hw_queue_default_without_overriding(monkeypatch):
    """Loading the library leaves a user-set GPU_MAX_HW_QUEUES alone and sets 16 otherwise (streams of pipeline slots
    need hardware queues of their own). Checked in child processes: the variable is read at load time."""
    import sys
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    # getenv through libc sees what setenv() did (os.environ is a snapshot taken at interpreter start)
    probe = ("import ctypes; ctypes.CDLL(%r); g = ctypes.CDLL(None).getenv; g.restype = ctypes.c_char_p; "
             "print(g(b'GPU_MAX_HW_QUEUES').decode())" % rpw_amd.LIB_PATH)
    assert subprocess.check_output([sys.executable, "-c", probe], env=env, text=True).strip() == "16"
    env["GPU_MAX_HW_QUEUES"] = "5"
    assert subprocess.check_output([sys.executable, "-c", probe], env=env, text=True).strip() == "5"
