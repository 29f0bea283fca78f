"""Shared helpers for the HIP-vs-oracle parity tests."""
from __future__ import annotations

import numpy as np

import oracle_lib as O
import rpw_amd

RESIDUAL_TOL = 1e-5   # north_star: per-point float residuals within 1e-5


def cfg_pair(**kw):
    """Same knobs for the product (PatchworkConfig) and for the oracle (OConfig)."""
    pc = rpw_amd.PatchworkConfig(**kw)
    okw = dict(kw)
    if "adaptive_seed_height" in okw:
        okw["adaptive_seed_height"] = int(bool(okw["adaptive_seed_height"]))
    return pc, O.make_config(**okw)


def sort_nodes(nodes, with_frame=False):
    # (len breaks the tie between the two children of a split that puts every point on one side: both start at the same
    # slot at the same depth, one of them with no points -- found by tools/soak.py, seed 73 case 492)
    keys = [nodes["len"], nodes["depth"], nodes["start"], nodes["patch"]]
    if with_frame:
        keys.append(nodes["frame"])
    return nodes[np.lexsort(keys)]


NODE_INT_FIELDS = ("patch", "depth", "start", "len", "kind", "iterations", "n_inliers", "split_axis")


def nodes_crc(nodes):
    """CRC32 of the recursion tree's integer fields, nodes sorted by (patch, start, depth), field by field: a multi-field
    selection of a structured array keeps its padding, and hashing uninitialised padding bytes is not reproducible."""
    import zlib
    s = sort_nodes(nodes)
    c = 0
    for f in NODE_INT_FIELDS:
        c = zlib.crc32(np.ascontiguousarray(s[f]).astype("<i8").tobytes(), c)
    return c & 0xFFFFFFFF


def assert_frame_parity(xyz, labels, residual, stats, nodes, ocfg, check_nodes=True, tag=""):
    """Compare one frame of HIP output with the oracle run on the same input."""
    ref = O.segment(xyz, ocfg, want_residual=residual is not None, want_nodes=check_nodes,
                    nodes_cap=max(1 << 16, len(xyz)))
    # 1. integer labels: bit exact
    nbad = int(np.count_nonzero(labels != ref.labels))
    assert nbad == 0, f"{tag}: {nbad} label mismatches of {len(xyz)}"
    # 2. stats: exact integers
    for k in ("n_in", "n_clean", "n_zone", "n_ground", "n_patches", "n_nodes", "n_splits", "max_depth",
              "total_iterations"):
        assert int(stats[k]) == int(ref.stats[k]), f"{tag}: stats[{k}] {stats[k]} != {ref.stats[k]}"
    # 3. per-point residuals within 1e-5 (sentinel -1 must agree exactly)
    if residual is not None:
        a, b = residual, ref.residual
        assert np.array_equal(a < 0, b < 0), f"{tag}: residual sentinel sets differ"
        big = np.isfinite(b) & (b >= 0)
        err = np.abs(a[big].astype(np.float64) - b[big].astype(np.float64))
        assert err.size == 0 or err.max() <= RESIDUAL_TOL, f"{tag}: residual max err {err.max()}"
    # 4. recursion tree: same nodes, same iteration counts, planes within 1e-5
    if check_nodes and nodes is not None:
        assert len(nodes) == len(ref.nodes), f"{tag}: node count {len(nodes)} != {len(ref.nodes)}"
        a, b = sort_nodes(nodes), sort_nodes(ref.nodes)
        for f in ("patch", "depth", "start", "len", "kind", "iterations", "n_inliers", "split_axis"):
            assert np.array_equal(a[f], b[f]), f"{tag}: node field {f} differs"
        fit = (b["kind"] == 0) | (b["kind"] == 4)
        for f in ("centroid", "normal"):
            d = np.abs(a[f][fit].astype(np.float64) - b[f][fit].astype(np.float64))
            assert d.size == 0 or d.max() <= RESIDUAL_TOL, f"{tag}: node {f} max err {d.max()}"
        ra, rb = a["residual"][fit], b["residual"][fit]
        finite = rb < 1e30
        assert np.array_equal(finite, ra < 1e30), f"{tag}: FLT_MAX residual sets differ"
        d = np.abs(ra[finite].astype(np.float64) - rb[finite].astype(np.float64))
        assert d.size == 0 or d.max() <= RESIDUAL_TOL, f"{tag}: node residual max err {d.max()}"
        sp = b["kind"] == 4
        assert np.array_equal(a["split_value"][sp], b["split_value"][sp]), f"{tag}: split medians differ"
    return ref
