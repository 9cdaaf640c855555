#!/usr/bin/env python3
"""A/B of the plane-form factor kernel on the batched configs[1] launch (128 factors x 131 072 points): HIP-event kernel time of the library named
by GLIM_AMD_LIB (default: the tree's), five rounds of 40 launches, plus the Gauss-Newton step error of one factor against the oracle.
  GLIM_AMD_LIB=build/ab/sm0/libglim_amd.so python tools/k4_ab.py     (run once per library, alternating, in ONE gpurun call)"""
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glim_amd import api, synth  # noqa: E402

F = int(os.environ.get("K4_AB_FACTORS", "128"))
ctx = api.Context(0, 1)
scene = synth.Scene.default()
dirs = synth.lidar_directions(128, 1024)
yaw0, radius = math.radians(10.0), 0.5 / math.radians(2.0)
poses = synth.arc_trajectory(F + 1, start=(-2.0 + radius * math.sin(yaw0), -0.5 - radius * math.cos(yaw0), 1.8), yaw0_deg=math.degrees(yaw0))
clouds = []
for i, T in enumerate(poses):
    g = api.PointCloudGPU.clone(synth.scan(scene, T, dirs, frame_id=i), ctx=ctx)
    g.find_neighbors(10, download=False)
    g.estimate_covariances(10)
    clouds.append(g)
vmaps = [api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(c) for c in clouds[:F]]
fset = api.NonlinearFactorSetGPU(ctx)
deltas = []
for i in range(F):
    fset.add(api.IntegratedVGICPFactorGPU(i, i + 1, vmaps[i], clouds[i + 1]))
    deltas.append(api.pose12(synth.relative_pose(poses[i], poses[i + 1])))
deltas = np.stack(deltas)
rounds = [fset.profile(deltas, iters=40) for _ in range(6)][1:]
out = {"lib": os.environ.get("GLIM_AMD_LIB", "tree"), "factors": F, "kernel_us": [round(r[0] * 1e3, 2) for r in rounds],
       "kernel_us_mean": float(np.mean([r[0] for r in rounds]) * 1e3), "linearize_us_mean": float(np.mean([r[1] for r in rounds]) * 1e3)}
# synchronous single-factor call, launch per call (the resident session is off in a default context)
single = api.NonlinearFactorSetGPU(ctx)
single.add(api.IntegratedVGICPFactorGPU(0, 1, vmaps[0], clouds[1]))
single.profile_sync(deltas[:1], iters=200)
out["single_dispatch_us"] = single.profile_sync(deltas[:1], iters=2000) * 1e3
if os.environ.get("K4_AB_PARITY", "1") == "1":
    from oracle import oracle as orc

    got = single.linearize_poses(deltas[:1])[0]
    tx, tc, _ = clouds[0].download(normals=False)
    sx, sc, _ = clouds[1].download(normals=False)
    D = np.eye(4)
    D[:3, :4] = deltas[0].reshape(3, 4)
    ref = orc.vgicp_linearize(orc.VoxelMap(0.5).insert(tx, tc.astype(np.float64)), sx, sc.astype(np.float64), D)
    step = np.abs(np.linalg.solve(got["H_ss"], -got["b_s"]) - np.linalg.solve(ref["H_ss"], -ref["b_s"])).max()
    out["parity"] = {"inliers_equal": bool(got["num_inliers"] == ref["num_inliers"]), "gn_step_err": float(step),
                     "H_rel_err": float(np.abs(got["H_ss"] - ref["H_ss"]).max() / np.abs(ref["H_ss"]).max()),
                     "b_rel_err": float(np.abs(got["b_s"] - ref["b_s"]).max() / np.abs(ref["b_s"]).max()),
                     "error_rel_err": float(abs(got["error"] - ref["error"]) / abs(ref["error"]))}
print(json.dumps(out))
