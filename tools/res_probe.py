import sys, time, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from glim_amd import api, synth, _lib
scene = synth.Scene.default(); dirs = synth.lidar_directions(128, 1024); poses = synth.arc_trajectory(2)
tgt = synth.scan(scene, poses[0], dirs, 0); src = synth.scan(scene, poses[1], dirs, 1)
delta = synth.relative_pose(poses[0], poses[1])
ctx = api.Context(0, 1)
tg = api.PointCloudGPU.clone(tgt, ctx=ctx); sg = api.PointCloudGPU.clone(src, ctx=ctx)
for g in (tg, sg):
    g.find_neighbors(10, download=False); g.estimate_covariances(10)
vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
fs = api.NonlinearFactorSetGPU(ctx); fs.add(api.IntegratedVGICPFactorGPU(0, 1, vm, sg))
T = api.pose12(delta)[None]
def stats():
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_int32()
    _lib.lib().glim_amd_debug_resident_stats(0, C.byref(a), C.byref(b), C.byref(c)); return a.value, b.value, c.value
for i in range(16):
    t0 = time.perf_counter(); fs.linearize_poses(T); dt = (time.perf_counter() - t0) * 1e6
    print(i, f"{dt:.1f} us", stats())
print("profile_sync us:", fs.profile_sync(T, iters=200) * 1e3, stats())
# ---- hold a session alive and look at what it costs a concurrent kernel
big = api.NonlinearFactorSetGPU(ctx)
for k in range(8): big.add(api.IntegratedVGICPFactorGPU(0, 1 + k, vm, sg))
Tb = np.repeat(T, 8, axis=0)
api.resident_stop(ctx); print("stopped", stats())
print("8-factor kernel alone (ms):", big.profile(Tb, iters=100), stats())
ctx.set_diag("resident_idle_us=400000"); print(ctx.get_diag() if hasattr(ctx, "get_diag") else "")
for i in range(6):
    fs.linearize_poses(T); print("restart", i, stats())
time.sleep(0.005); print("after 5 ms", stats())
print("8-factor kernel beside the session (ms):", big.profile(Tb, iters=100), stats())
print("sync call itself (us):", fs.profile_sync(T, iters=200) * 1e3, stats())
time.sleep(0.05); print("after 50 ms", stats())
api.resident_stop(ctx); print("stopped", stats())
