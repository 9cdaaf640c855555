import sys, os, numpy as np
sys.path.insert(0, '.')
import bench
from glim_amd import api, synth
ctx = api.Context(0, 1)
poses = synth.arc_trajectory(2)
clouds = bench.make_frames(api, ctx, poses, 128, 1024)
vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(clouds[0])
T = api.pose12(synth.relative_pose(poses[0], poses[1]))[None]
for env in ({}, {"GLIM_AMD_NO_POLL": "1"}, {"GLIM_AMD_NO_INLINE_POSE": "1"}, {"GLIM_AMD_NO_POLL": "1", "GLIM_AMD_NO_INLINE_POSE": "1"}):
    for k in ("GLIM_AMD_NO_POLL", "GLIM_AMD_NO_INLINE_POSE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    fs = api.NonlinearFactorSetGPU(ctx)
    fs.add(api.IntegratedVGICPFactorGPU(0, 1, vm, clouds[1]))
    ms = fs.profile_sync(T, iters=1000)
    r = fs.linearize_poses(T)[0]
    print(env, "ms/call", round(ms, 4), "calls/s", round(1e3 / ms), "inliers", r["num_inliers"])
