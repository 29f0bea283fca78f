#!/usr/bin/env python3
"""Regenerates tests/golden/golden.json from the CPU oracle on the seeded workloads.

The reference itself cannot be run in this image (it needs <Eigen/Dense>) and
holds no fixtures for this path, so these vectors are outputs of the build's OWN restatement
(oracle/rpw_oracle.cpp): they pin the oracle against regressions and let the GPU box check the HIP
path without re-deriving anything, but they do not pin it against the reference ("parity unpinned").
Usage:  python tests/golden/make_golden.py
"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O          # noqa: E402
import rpw_amd                  # noqa: E402
from parity_util import nodes_crc   # noqa: E402

CASES = [
    ("c1", 1, {}),
    ("c1", 2, dict(sensor_height=1.2, filtering_radius=50.0, num_sectors=8, max_iter=50)),
    ("c1", 3, dict(sensor_height=0.3, th_dist=0.1, max_iter=50, th_seeds=0.1)),
    ("c1", 4, dict(max_iter=0)),
    ("c1", 5, dict(adaptive_seed_height=0)),
    ("c1", 6, dict(sensor_height=-5.0)),
    ("c2", 0, dict(max_split_depth=3)),
    ("c2", 0, dict(max_split_depth=3, max_iter=0)),
    ("c4", 0, {}),
    ("c4", 0, dict(max_iter=0)),
    ("c5", 0, dict(max_split_depth=6)),
    ("c5", 0, dict(max_split_depth=6, max_iter=0)),
    # natural recursion (no max_iter = 0 forcing): far-field clutter + tight th_dist, fits split to the configured depth
    ("rough", 0, dict(max_split_depth=3, th_dist=0.0005)),
    ("rough", 1, dict(max_split_depth=6, th_dist=0.0005)),
]


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def run_case(name, seed, kw):
    xyz = rpw_amd.workloads.make(name, seed)
    r = O.segment(xyz, O.make_config(**kw), want_residual=True, want_nodes=True, nodes_cap=max(1 << 16, len(xyz)))
    nodes = r.nodes[np.lexsort([r.nodes["depth"], r.nodes["start"], r.nodes["patch"]])]
    head = nodes[:4]
    return {
        "workload": name, "seed": seed, "config": kw, "n": int(len(xyz)), "input_crc32": crc(xyz),
        "labels_crc32": crc(r.labels), "label_hist": np.bincount(r.labels, minlength=4).tolist(),
        "residual_crc32": crc(r.residual),
        "stats": {k: int(v) for k, v in r.stats.items()},
        "nodes_crc32": nodes_crc(r.nodes),      # integer fields of the whole recursion tree, field by field (no padding bytes)
        "first_nodes": [{"patch": int(n["patch"]), "len": int(n["len"]), "kind": int(n["kind"]), "iterations": int(n["iterations"]),
                         "centroid": [float(v) for v in n["centroid"]], "normal": [float(v) for v in n["normal"]],
                         "residual": float(n["residual"])} for n in head],
    }


if __name__ == "__main__":
    out = {"generator": "tests/golden/make_golden.py", "oracle": "oracle/rpw_oracle.cpp",
           "note": "outputs of the build's own CPU restatement; the reference has no fixtures for this path",
           "cases": [run_case(*c) for c in CASES]}
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["cases"]), "cases")
