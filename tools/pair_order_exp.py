#!/usr/bin/env python3
"""How much does the ORDER of the pair list matter for the all-pairs cost of configs[3] (256 merged submaps, 32 640 binary factors)?
Every factor is one block of the general kernel; the ~1000 resident blocks are the next ~1000 factors of the list, so the order decides
which source streams (2.2 MB each) and target maps are in flight together, i.e. what the 256 MiB Infinity Cache can serve.  (GPU box)"""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from glim_amd import api, synth  # noqa: E402

S = int(os.environ.get("SUBMAPS", "256"))
ctx = api.Context(0, 1)
submaps = bench.make_merged_submaps(api, ctx, S, 4, 40, 512)
clouds = [g for _, g in submaps]
poses = [T for T, _ in submaps]
vmaps = [api.GaussianVoxelMapGPU(1.0, ctx=ctx).insert(c) for c in clouds]


def tiled(T):
    out = []
    for ti in range(0, S, T):
        for tj in range(ti, S, T):
            out += [(i, j) for i in range(ti, min(ti + T, S)) for j in range(max(tj, i + 1), min(tj + T, S))]
    return out


orders = {} if os.environ.get("ONLY_FIRST") else None
_all = {
    "target-major (i, j>i)": [(i, j) for i in range(S) for j in range(i + 1, S)],
    "source-major (i<j, j)": [(i, j) for j in range(S) for i in range(j)],
    "tiles 16x16": tiled(16), "tiles 32x32": tiled(32), "tiles 64x64": tiled(64),
}
orders = _all if orders is None else {k: _all[k] for k in list(_all)[:1]}
for name, pairs in orders.items():
    assert len(set(pairs)) == S * (S - 1) // 2
    fset = api.NonlinearFactorSetGPU(ctx)
    for i, j in pairs:
        fset.add(api.IntegratedVGICPFactorGPU(i, j, vmaps[i], clouds[j]))
    deltas = np.stack([api.pose12(synth.relative_pose(poses[i], poses[j])) for i, j in pairs])
    ms = [fset.profile(deltas, iters=5)[0] for _ in range(3)]
    out = fset.linearize_poses(deltas)
    print(json.dumps({"order": name, "kernel_ms": [round(m, 3) for m in ms], "total_error": float(sum(o["error"] for o in out))}), flush=True)
    fset.close()
