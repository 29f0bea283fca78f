"""GPU test: the three forms of patch_fit give the oracle's result on the same frames.

The launcher picks a form by the number of patches in the launch (a chain server per patch for a frame or a few, one chain
server per seven patches for a batch); the single-wave form is the fallback the other two were derived from. RPW_FIT_WAVES=1|3|4 forces a form
for every launch of a process, so each form runs here in a child process of its own over the same frames -- single frames and
one batch -- and every result is compared with the oracle (labels and integers exact, residuals within 1e-5).
Reference semantics: src/recursive_patchwork.cpp:109-308 (fitPlaneAndSplit), :310-426 (filterGroundPoints).
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path[:0] = [%(root)r, %(root)r + "/tests"]
import numpy as np
import torch                                   # first: one HIP runtime for torch and librpw_hip.so
import rpw_amd
from rpw_amd import workloads as W
from parity_util import cfg_pair, assert_frame_parity

frames = [W.c2_hdl64(0), W.c1_test_cloud(1), W.c4_iac_fused(0)[:150000], W.rough_terrain(2, n_clutter=60000)[:140000],
          W.c1_test_cloud(3, n=67), np.zeros((0, 3), np.float32),
          W.c5_dense(0)]             # with num_sectors = 4 its largest node has 106 648 slots: 32-bit index entries in every form
cases = [dict(max_split_depth=3), dict(max_split_depth=3, max_iter=0), dict(max_split_depth=6, th_dist=0.0005),
         dict(max_split_depth=2, adaptive_seed_height=False, num_sectors=4)]
n = 0
for kw in cases:
    pc, oc = cfg_pair(**kw)
    eng = rpw_amd.Engine(pc, max_points=sum(len(f) for f in frames), max_frames=len(frames))
    for i, f in enumerate(frames):
        labels, res, st = eng.segment(f, want_residual=True)
        assert_frame_parity(f, labels, res, st, None, oc, check_nodes=False, tag="form %(form)s %%s frame %%d" %% (kw, i))
        n += 1
    # one batch of all frames, device resident
    off = np.zeros(len(frames) + 1, np.uint32); off[1:] = np.cumsum([len(f) for f in frames])
    xyz = torch.from_numpy(np.concatenate(frames)).cuda()
    lab = torch.empty(int(off[-1]), dtype=torch.uint8, device="cuda")
    res = torch.empty(int(off[-1]), dtype=torch.float32, device="cuda")
    eng.segment_batch_device(xyz.data_ptr(), off, lab.data_ptr(), res.data_ptr(), torch.cuda.current_stream().cuda_stream)
    stats = eng.batch_stats(len(frames))
    torch.cuda.synchronize()
    lab, res = lab.cpu().numpy(), res.cpu().numpy()
    for i, f in enumerate(frames):
        a, b = int(off[i]), int(off[i + 1])
        assert_frame_parity(f, lab[a:b], res[a:b], stats[i], None, oc, check_nodes=False, tag="form %(form)s %%s batch frame %%d" %% (kw, i))
        n += 1
    eng.close()
print("FORM_OK", %(form)s, n)
"""


@pytest.mark.parametrize("form", [1, 3, 4])
def test_every_form_of_patch_fit_equals_the_oracle(form):
    env = dict(os.environ, RPW_FIT_WAVES=str(form))
    out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, form=form)], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "FORM_OK %d" % form in out.stdout
