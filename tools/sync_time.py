"""Where the time of one synchronous single-factor linearize goes when called from Python (GPU box)."""
import os, sys, time, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glim_amd import api, synth
from glim_amd._lib import lib, Linearized6
if len(sys.argv) > 1 and sys.argv[1] == "torch":  # as bench.py does: torch initialised first, our kernels on torch's current stream
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    ctx = api.Context(0, 1, external_stream=torch.cuda.current_stream().cuda_stream)
else:
    ctx = api.Context(0, 1)
scene = synth.Scene.default()
poses = synth.arc_trajectory(2)
gs = []
for i, T in enumerate(poses):
    g = api.PointCloudGPU.clone(synth.scan(scene, T, synth.lidar_directions(128, 1024), i), ctx=ctx)
    g.find_neighbors(10, download=False); g.estimate_covariances(10); gs.append(g)
vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(gs[0])
fs = api.NonlinearFactorSetGPU(ctx)
fs.add(api.IntegratedVGICPFactorGPU(0, 1, vm, gs[1]))
T = api.pose12(synth.relative_pose(poses[0], poses[1])).reshape(1, 12).copy()
out = (Linearized6 * 1)()
Tp = T.ctypes.data_as(C.POINTER(C.c_double))
L = lib()
for _ in range(50): L.glim_amd_factor_set_linearize(fs._h, Tp, out)
n = 2000
t0 = time.perf_counter()
for _ in range(n): L.glim_amd_factor_set_linearize(fs._h, Tp, out)
raw = (time.perf_counter() - t0) / n * 1e6
t0 = time.perf_counter()
for _ in range(n): fs.linearize_poses(T)
wrapped = (time.perf_counter() - t0) / n * 1e6
print(f"C call from Python (ctypes only): {raw:.1f} us;  api.linearize_poses: {wrapped:.1f} us;  in-library loop (profile_sync): {fs.profile_sync(T, iters=500) * 1e3:.1f} us")
