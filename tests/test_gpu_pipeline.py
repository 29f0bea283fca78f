"""GPU tests of the host-pointer pipeline (rpw_submit / rpw_wait) and of the handle's single-occupancy guard.

Every reference caller holds a frame in host memory and calls filterGroundPoints once per frame
(reference src/recursive_patchwork_node.cpp:91, src/main.cpp:268, test/test_recursive_patchwork.cpp:68,155);
the pipeline is that call with K frames in flight. Labels must equal the oracle's whatever the slot, the
order of the waits and the reuse pattern.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import rpw_amd
from rpw_amd import workloads as W
from parity_util import cfg_pair, assert_frame_parity

pytestmark = pytest.mark.gpu


def _frames():
    rs = np.random.RandomState(7)
    base = [W.c2_hdl64(0), W.c2_hdl64(1), W.c1_test_cloud(2), W.c1_test_cloud(3, n=2049), W.c4_iac_fused(0)[:90000],
            np.zeros((0, 3), np.float32), np.array([[1, 2, 0], [np.nan, 0, 0], [3, 4, 0.1]], np.float32)]
    out = []
    for i in range(22):
        b = base[i % len(base)]
        if len(b) > 10:
            b = b[rs.permutation(len(b))[: rs.randint(len(b) // 2, len(b) + 1)]] if i % 3 == 1 else b[:: 1 + (i % 2)]
        out.append(np.ascontiguousarray(b, np.float32))
    return out


def test_submit_wait_equals_oracle_with_slots_reused_out_of_order(engine_factory):
    frames = _frames()
    pc, oc = cfg_pair(max_split_depth=3)
    eng = engine_factory(pc, max_points=max(len(f) for f in frames), max_frames=1)
    K = 8
    eng.set_pipeline(K)
    refs = [O.segment(f, oc, want_residual=True, want_nodes=False) for f in frames]
    rs = np.random.RandomState(3)
    in_slot = {}                                       # slot -> frame index
    todo = list(range(len(frames)))
    done = 0
    while todo or in_slot:
        free = [s for s in range(K) if s not in in_slot]
        rs.shuffle(free)
        while todo and free and (len(in_slot) < K) and (rs.rand() < 0.8 or not in_slot):
            s = free.pop()
            i = todo.pop(0)
            eng.submit(s, frames[i], want_residual=(i % 2 == 0))
            in_slot[s] = i
        # wait a RANDOM pending slot, not the oldest
        s = list(in_slot)[rs.randint(len(in_slot))]
        i = in_slot.pop(s)
        labels, res, st = eng.wait(s)
        assert np.array_equal(labels, refs[i].labels), f"frame {i} in slot {s}"
        for k in ("n_in", "n_clean", "n_zone", "n_ground", "n_patches", "n_nodes", "n_splits", "max_depth", "total_iterations"):
            assert int(st[k]) == int(refs[i].stats[k]), (i, k)
        if res is not None:
            ok = refs[i].residual >= 0
            assert np.array_equal(res < 0, ~ok)
            assert not ok.any() or np.abs(res[ok] - refs[i].residual[ok]).max() <= 1e-5
        done += 1
    assert done == len(frames)


def test_pipeline_status_codes_and_in_place_input(engine_factory):
    xyz = W.c1_test_cloud(5)
    eng = engine_factory(rpw_amd.PatchworkConfig(), max_points=len(xyz), max_frames=1)
    L = eng._L
    lab = np.empty(len(xyz), np.uint8)
    assert L.rpw_submit(eng._h, 0, xyz.ctypes.data, len(xyz), 0) == 1                  # no pipeline yet: RPW_ERR_INVALID
    eng.set_pipeline(2)
    assert L.rpw_get_pipeline(eng._h) == 2
    assert L.rpw_wait(eng._h, 0, lab.ctypes.data, None, None) == 1                     # nothing submitted
    assert L.rpw_submit(eng._h, 2, xyz.ctypes.data, len(xyz), 0) == 1                  # slot out of range
    big = np.zeros((len(xyz) + 1, 3), np.float32)
    assert L.rpw_submit(eng._h, 0, big.ctypes.data, len(big), 0) == 4                  # RPW_ERR_CAPACITY
    # in place: fill the slot's pinned buffer, submit that pointer
    buf = eng.slot_input(1, len(xyz))
    buf[:] = xyz
    eng.submit(1, buf)
    assert L.rpw_submit(eng._h, 1, xyz.ctypes.data, len(xyz), 0) == 6                  # RPW_ERR_BUSY
    assert b"rpw_wait" in L.rpw_last_error(eng._h)
    res = np.empty(len(xyz), np.float32)
    assert L.rpw_wait(eng._h, 1, lab.ctypes.data, res.ctypes.data, None) == 1          # residual was not asked for
    # that failed wait must not have consumed the frame
    labels, _, st = eng.wait(1)
    ref = O.segment(xyz, O.make_config(), want_residual=False, want_nodes=False)
    assert np.array_equal(labels, ref.labels) and st["n_ground"] == ref.stats["n_ground"]
    # poll: eventually done, and the synchronous call still works beside the pipeline
    eng.submit(0, xyz)
    l2, _, _ = eng.segment(xyz)
    while not eng.poll(0):
        pass
    l3, _, _ = eng.wait(0)
    assert np.array_equal(l2, ref.labels) and np.array_equal(l3, ref.labels)
    eng.set_pipeline(0)
    assert L.rpw_get_pipeline(eng._h) == 0


def test_submit_pointcloud2_strided_records(engine_factory):
    xyz = W.c2_hdl64(2)[:60000]
    n = len(xyz)
    step = 32
    rec = np.zeros((n, step), np.uint8)
    rec[:, 4:8] = xyz[:, 0:1].view(np.uint8).reshape(n, 4)
    rec[:, 12:16] = xyz[:, 1:2].view(np.uint8).reshape(n, 4)
    rec[:, 20:24] = xyz[:, 2:3].view(np.uint8).reshape(n, 4)
    pc, oc = cfg_pair(max_split_depth=3)
    eng = engine_factory(pc, max_points=n, max_frames=1)
    eng.set_pipeline(3)
    for s in range(3):
        eng.submit_pointcloud2(s, rec.ravel(), n, step, 4, 12, 20, want_residual=(s == 1))
    ref = O.segment(xyz, oc, want_residual=True, want_nodes=False)
    for s in (2, 0, 1):
        labels, res, st = eng.wait(s)
        assert np.array_equal(labels, ref.labels)
        assert int(st["total_iterations"]) == int(ref.stats["total_iterations"])
    # layout checks must not wrap: an offset near 2^32 is refused, as is a record too short for its fields
    L = eng._L
    lab = np.empty(n, np.uint8)
    for offs in ((0xFFFFFFFC, 4, 8), (0, 0xFFFFFFFC, 8), (0, 4, 0xFFFFFFFC), (0, 4, step), (0, 4, step - 2)):
        assert L.rpw_segment_pointcloud2(eng._h, rec.ctypes.data, n, step, offs[0], offs[1], offs[2], lab.ctypes.data, None, None) == 1
        assert L.rpw_submit_pointcloud2(eng._h, 0, rec.ctypes.data, n, step, offs[0], offs[1], offs[2], 0) == 1
    assert L.rpw_segment_pointcloud2(eng._h, rec.ctypes.data, n, 8, 0, 4, 4, lab.ctypes.data, None, None) == 1


def test_one_handle_on_two_streams_is_serialised_not_corrupted(engine_factory):
    """The handle's workspace holds one batch at a time: a second batch on ANOTHER stream waits (on the device) for
    the first instead of sharing its scratch memory. Both results must equal the oracle."""
    import torch
    fa = [W.c2_hdl64(3), W.c2_hdl64(4)]
    fb = [W.c1_test_cloud(6), W.c2_hdl64(5)[::2]]
    pc, oc = cfg_pair(max_split_depth=3)
    cap = max(sum(len(f) for f in fa), sum(len(f) for f in fb))
    eng = engine_factory(pc, max_points=cap, max_frames=2)
    dev = torch.device("cuda", 0)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def dev_batch(fr):
        off = np.zeros(len(fr) + 1, np.uint32); off[1:] = np.cumsum([len(f) for f in fr])
        return torch.from_numpy(np.concatenate(fr)).to(dev), off, torch.empty(int(off[-1]), dtype=torch.uint8, device=dev)
    xa, oa, la = dev_batch(fa)
    xb, ob, lb = dev_batch(fb)
    torch.cuda.synchronize()
    for _ in range(3):                                 # back to back, no host synchronisation in between
        eng.segment_batch_device(xa.data_ptr(), oa, la.data_ptr(), None, s1.cuda_stream)
        eng.segment_batch_device(xb.data_ptr(), ob, lb.data_ptr(), None, s2.cuda_stream)
    st_b = eng.batch_stats(2)                          # stats of the LAST batch (b)
    torch.cuda.synchronize()
    for fr, off, lab in ((fa, oa, la), (fb, ob, lb)):
        got = lab.cpu().numpy()
        for i, f in enumerate(fr):
            ref = O.segment(f, oc, want_residual=False, want_nodes=False)
            assert np.array_equal(got[off[i]: off[i + 1]], ref.labels)
            if fr is fb:
                assert int(st_b["n_ground"][i]) == int(ref.stats["n_ground"])
    # a rejected call does not become "the last batch"
    bad = np.array([0, 5, 3], np.uint32)
    with pytest.raises(rpw_amd.RpwError):
        eng.segment_batch_device(xb.data_ptr(), bad, lb.data_ptr(), None, s1.cuda_stream)
    st_again = eng.batch_stats(2)
    assert np.array_equal(st_again["n_ground"], st_b["n_ground"])
