#!/usr/bin/env python3
"""SURVEY 8d config 2 variants (GPU box): VGICP linearize for K = 1, 8, 64 factors per launch, unary / binary, surface validation off / on.
Per variant: HIP-event time of the fused kernel, of kernel + FP64 finalise, and the synchronous host-visible call (pose in, records out).
Writes one JSON document to stdout (and gpurun_out/batch_sweep.json)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from glim_amd import api, synth  # noqa: E402

F = 64
ctx = api.Context(0, 1)
poses = synth.arc_trajectory(F + 1, start=(-12.0, -7.0, 1.8), yaw0_deg=10.0)
t0 = time.time()
clouds = bench.make_frames(api, ctx, poses, 128, 1024)
vmaps = [api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(c) for c in clouds[:F]]
deltas = np.stack([api.pose12(synth.relative_pose(poses[i], poses[i + 1])) for i in range(F)])
gen_s = time.time() - t0
rows = []
for K in (1, 8, 64):
    for binary in (True, False):
        for surf in (False, True):
            fset = api.NonlinearFactorSetGPU(ctx)
            for i in range(K):
                f = api.IntegratedVGICPFactorGPU(i, i + 1, vmaps[i], clouds[i + 1]) if binary else \
                    api.IntegratedVGICPFactorGPU(np.eye(4), i + 1, vmaps[i], clouds[i + 1])
                f.set_enable_surface_validation(surf)
                fset.add(f)
            T = deltas[:K]
            ms_k, ms_l = fset.profile(T, iters=200 if K < 64 else 100)
            sync_ms = fset.profile_sync(T, iters=300 if K < 64 else 100)
            out = fset.linearize_poses(T)
            n_pts = [clouds[i + 1].size() for i in range(K)]
            n_vox = [vmaps[i].voxelmap_info()["num_voxels"] for i in range(K)]
            algo = bench.algorithmic_bytes(n_pts, n_vox) + (12.0 * sum(n_pts) if surf else 0.0)
            rows.append({"factors_per_launch": K, "factor_type": "binary" if binary else "unary", "surface_validation": surf,
                         "kernel_us": round(ms_k * 1e3, 2), "kernel_plus_finalize_us": round(ms_l * 1e3, 2), "sync_call_us": round(sync_ms * 1e3, 2),
                         "linearize_calls_per_s_batched": round(K / (ms_l * 1e-3)), "linearize_calls_per_s_sync": round(K / (sync_ms * 1e-3)),
                         "algorithmic_GBs": round(algo / (ms_k * 1e-3) / 1e9, 1), "mean_inlier_fraction": round(float(np.mean([o["num_inliers"] for o in out]) / np.mean(n_pts)), 4)})
            fset.close()
doc = {"workload": "SURVEY 8d config 2: 131072-pt scans vs 0.5 m voxel maps; K factors per NonlinearFactorSetGPU::linearize", "device": ctx.device_info()["name"],
       "scene_generation_s": round(gen_s, 1), "rows": rows}
s = json.dumps(doc, indent=1)
print(s)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "batch_sweep.json"), "w").write(s)
