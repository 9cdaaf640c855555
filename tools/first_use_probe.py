import sys, time
sys.path.insert(0, ".")
import numpy as np
from glim_amd import api, synth
scene = synth.Scene.default(); dirs = synth.lidar_directions(128, 1024); poses = synth.arc_trajectory(3)
rng = np.random.default_rng(3)
def frame(i):
    pts = synth.scan(scene, poses[i], dirs, 500 + i)
    return pts[np.sort(rng.choice(len(pts), 10000, replace=False))]
ctx = api.Context(0, 1)
a = api.PointCloudGPU.clone(frame(0), ctx=ctx); a.find_neighbors(10, download=False); a.estimate_covariances(10)
vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(a)
b_host = frame(1)
bg = api.PointCloudGPU.clone(b_host, ctx=ctx); bg.find_neighbors(10, download=False); bg.estimate_covariances(10)
xyz, c32, n32 = bg.download()
for diag in ("resident=0", ""):
    ctx.set_diag(diag)
    # keep a resident session alive for another plan, as bench.py's earlier measurements do
    hot = api.NonlinearFactorSetGPU(ctx); hot.add(api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, bg))
    for _ in range(10): hot.linearize({1: np.eye(4)})
    ts = []
    for rep in range(6):
        g = api.PointCloudGPU.clone(xyz, c32, n32, ctx=ctx)
        t0 = time.perf_counter()
        one = api.NonlinearFactorSetGPU(ctx); one.add(api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, g))
        t1 = time.perf_counter(); one.linearize({1: np.eye(4)}); t2 = time.perf_counter(); one.linearize({1: np.eye(4)}); t3 = time.perf_counter()
        one.close(); t4 = time.perf_counter(); g.close(); t5 = time.perf_counter()
        ts.append([(t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6, (t5 - t4) * 1e6])
    print(diag or "default", "[add, first linearize, second, set close, cloud close] us:", np.round(np.array(ts), 1).tolist())
    hot.close()
