#!/usr/bin/env python3
"""Kernel experiment driver (GPU box): times the fused VGICP kernel on the bench workload (F x 131 072-pt factors, 0.5 m maps) for a
list of environment settings.  KEXP='[{"GLIM_AMD_BUCKET_FACTOR": 6}, ...]'  (the library itself is chosen with GLIM_AMD_LIB per process).
Environment settings that change the MAP (GLIM_AMD_BUCKET_FACTOR) are applied before the maps are (re)built."""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from glim_amd import api, synth  # noqa: E402

F = int(os.environ.get("KEXP_FACTORS", "128"))
ctx = api.Context(0, 1)
yaw0 = math.radians(10.0)
radius = 0.5 / math.radians(2.0)
poses = synth.arc_trajectory(F + 1, start=(-2.0 + radius * math.sin(yaw0), -0.5 - radius * math.cos(yaw0), 1.8), yaw0_deg=10.0)
clouds = bench.make_frames(api, ctx, poses, 128, 1024)
deltas = np.stack([api.pose12(synth.relative_pose(poses[i], poses[i + 1])) for i in range(F)])
n_pts = [c.size() for c in clouds[1:]]
combos = json.loads(os.environ.get("KEXP", "null")) or [{}]
tag = os.environ.get("KEXP_TAG", "main")
for c in combos:
    for k, v in c.items():
        os.environ[k] = str(v)
    vmaps = [api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(x) for x in clouds[:F]]
    n_vox = [v.voxelmap_info()["num_voxels"] for v in vmaps]
    fset = api.NonlinearFactorSetGPU(ctx)
    for i in range(F):
        fset.add(api.IntegratedVGICPFactorGPU(i, i + 1, vmaps[i], clouds[i + 1]))
    best = None
    for rep in range(3):
        ms_k, ms_l = fset.profile(deltas, iters=40)
        best = ms_k if best is None else min(best, ms_k)
    out = fset.linearize_poses(deltas)[0]
    algo = bench.algorithmic_bytes(n_pts, n_vox)
    print(json.dumps({"lib": tag, **c, "kernel_us": round(best * 1e3, 1), "lin_us": round(ms_l * 1e3, 1), "algo_TBs": round(algo / best / 1e9, 2),
                      "inl": out["num_inliers"], "H00": out["H_ss"][0, 0]}), flush=True)
    for k in c:
        os.environ.pop(k, None)
    fset.close()
    for v in vmaps:
        v.close()
