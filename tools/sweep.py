#!/usr/bin/env python3
"""Kernel-variant sweep (GPU box): times the fused VGICP kernel for (U, MINW, PPT, XCD map) combinations on the bench workload."""
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from glim_amd import api  # noqa: E402

F = int(os.environ.get("SWEEP_FACTORS", "64"))
ctx = api.Context(0, 1)
from glim_amd import synth  # noqa: E402

_poses = synth.arc_trajectory(F + 1, start=(-12.0, -7.0, 1.8), yaw0_deg=10.0)
_clouds = bench.make_frames(api, ctx, _poses, 128, 1024)
wl = {"clouds": _clouds, "vmaps": [api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(c) for c in _clouds[:F]],
      "deltas": np.stack([api.pose12(synth.relative_pose(_poses[i], _poses[i + 1])) for i in range(F)])}
n_pts = [c.size() for c in wl["clouds"][1:]]
n_vox = [v.voxelmap_info()["num_voxels"] for v in wl["vmaps"]]
algo = float(sum(48 * n + 68 * v + 488 for n, v in zip(n_pts, n_vox)))
combos = json.loads(os.environ.get("SWEEP", "null")) or [
    {"GLIM_AMD_U": u, "GLIM_AMD_MINW": w, "GLIM_AMD_PPT": p}
    for (u, w) in [(1, 3), (1, 4), (2, 2), (2, 3), (2, 4), (4, 1), (4, 2), (4, 3)] for p in (4, 8, 16)
]
ref = None
for c in combos:
    for k, v in c.items():
        os.environ[k] = str(v)
    fset = api.NonlinearFactorSetGPU(ctx)
    for i in range(F):
        fset.add(api.IntegratedVGICPFactorGPU(i, i + 1, wl["vmaps"][i], wl["clouds"][i + 1]))
    ms_k, ms_l = fset.profile(wl["deltas"], iters=20)
    out = fset.linearize_poses(wl["deltas"])[0]
    if ref is None:
        ref = out
    dev = float(np.abs(out["H_ss"] - ref["H_ss"]).max() / np.abs(ref["H_ss"]).max())
    print(json.dumps({**c, "kernel_us": round(ms_k * 1e3, 1), "lin_us": round(ms_l * 1e3, 1), "GBs": round(algo / ms_k / 1e6, 0),
                      "inl": out["num_inliers"], "relH": dev}), flush=True)
    for k in c:
        os.environ.pop(k, None)
    fset.close()
