import torch; torch.cuda.init()  # torch first: its bundled HIP runtime must be the one in the process
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
import time
for k in ("GLIM_AMD_NO_POLL", "GLIM_AMD_NO_INLINE_POSE"):
    os.environ.pop(k, None)
fs = api.NonlinearFactorSetGPU(ctx)
fs.add(api.IntegratedVGICPFactorGPU(0, 1, vm, clouds[1]))
for _ in range(50): fs.linearize_poses(T)
t0 = time.perf_counter()
for _ in range(500): fs.linearize_poses(T)
print("python loop calls/s", 500 / (time.perf_counter() - t0))
import torch
ctx2 = api.Context(0, 1, external_stream=torch.cuda.current_stream().cuda_stream)
c2 = bench.make_frames(api, ctx2, poses, 128, 1024)
vm2 = api.GaussianVoxelMapGPU(0.5, ctx=ctx2).insert(c2[0])
fs2 = api.NonlinearFactorSetGPU(ctx2)
fs2.add(api.IntegratedVGICPFactorGPU(0, 1, vm2, c2[1]))
print("torch-stream ctx: C loop ms/call", fs2.profile_sync(T, iters=500))
t0 = time.perf_counter()
for _ in range(500): fs2.linearize_poses(T)
print("torch-stream ctx: python loop calls/s", 500 / (time.perf_counter() - t0))
