"""GPU parity tests proper: HIP path (through the C-ABI) vs the CPU oracle, same seeded inputs.

Bar: labels bit-exact, stats exact, per-point residuals and per-node planes within 1e-5.
"""
import numpy as np
import pytest

import rpw_amd
from rpw_amd import workloads as W
from parity_util import cfg_pair, assert_frame_parity

pytestmark = pytest.mark.gpu


CASES = [
    # (workload, seed, config overrides)
    ("c1", 1, dict()),
    ("c1", 2, dict(sensor_height=1.2, filtering_radius=50.0, num_sectors=8, max_iter=50)),   # the reference test's config
    ("c1", 3, dict(sensor_height=0.3, th_dist=0.1, max_iter=50, th_seeds=0.1)),              # the ROS2 node's thresholds
    ("c1", 4, dict(max_iter=0)),                                                            # forced recursion
    ("c1", 5, dict(adaptive_seed_height=False)),                                            # percentile seeds
    ("c1", 6, dict(sensor_height=-5.0)),                                                    # < 3 seeds -> 3 lowest z
    ("c2", 0, dict(max_split_depth=3)),
    ("c2", 1, dict(max_split_depth=3, max_iter=0)),
    ("c2", 2, dict(max_split_depth=3, sensor_height=-1.5, th_dist=0.1)),
    ("c4", 0, dict()),
    ("c4", 0, dict(max_iter=0)),
]


@pytest.mark.parametrize("name,seed,kw", CASES)
def test_single_frame_parity(engine_factory, name, seed, kw):
    xyz = W.make(name, seed)
    pc, oc = cfg_pair(**kw)
    eng = engine_factory(pc, max_points=len(xyz) + 16, max_frames=1)
    eng.set_trace(True, max(1 << 16, len(xyz)))
    labels, res, st = eng.segment(xyz, want_residual=True)
    nodes = eng.nodes(max(1 << 16, len(xyz)))
    assert_frame_parity(xyz, labels, res, st, nodes, oc, tag=f"{name}/{seed}/{kw}")


def test_device_math_hooks(engine_factory):
    """device atan2f model == libm (via the oracle's angle) and device eigensolver == oracle's."""
    import oracle_lib as O
    eng = engine_factory(rpw_amd.PatchworkConfig(), max_points=1024)
    rs = np.random.RandomState(0)
    n = 2_000_000
    x = rs.uniform(-150, 150, n).astype(np.float32)
    y = rs.uniform(-150, 150, n).astype(np.float32)
    dev = eng.debug_angles(x, y)
    host = rpw_amd.debug_angles_host(x, y)
    assert np.array_equal(dev.view(np.uint32), host.view(np.uint32))
    # eigensolver: random covariances of planar-ish clouds
    m = 20000
    covs = np.empty((m, 9), np.float32)
    for i in range(m):
        pts = rs.normal(size=(16, 3)).astype(np.float32) * np.array([5, 3, 0.05], np.float32)
        c = np.cov(pts.T).astype(np.float32)
        covs[i] = c.reshape(-1)
    nd = eng.debug_eig3(covs)
    for i in range(0, m, 7):
        info, ev, Q = O.eig3(covs[i].reshape(3, 3))
        nref = Q[:, 0].copy()
        if nref[2] < 0:
            nref = -nref
        assert np.array_equal(nd[i].view(np.uint32), nref.view(np.uint32)), (i, nd[i], nref)
