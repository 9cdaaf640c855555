"""Per-stage device timings of the per-frame path (upload -> kNN -> covariance -> voxel map), GPU box."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from glim_amd import api, synth
ctx = api.Context(0, 1)
scene = synth.Scene.default()
def t(f, reps=20):
    f(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for name, pts, res in (("lidar131k", synth.scan(scene, synth.arc_trajectory(1)[0], synth.lidar_directions(128, 1024), 0), 0.5),
                       ("rgbd307k", synth.scan(synth.Scene.small_room(), synth.pose(-2.5, -1.5, 1.4, 0.5), synth.pinhole_directions(640, 480, 70, 55), 0, sigma=0.002, max_range=8.0, min_range=0.3), 0.1)):
    g = api.PointCloudGPU.clone(pts, ctx=ctx)
    up = t(lambda: api.PointCloudGPU.clone(pts, ctx=ctx).close())
    kn = t(lambda: g.find_neighbors(10, download=False))
    cv = t(lambda: g.estimate_covariances(10))
    vm = t(lambda: api.GaussianVoxelMapGPU(res, ctx=ctx).insert(g).close())
    v = api.GaussianVoxelMapGPU(res, ctx=ctx).insert(g).voxelmap_info()["num_voxels"]
    print(f"{name}: N={len(pts)} V={v} upload {up:.3f} ms  knn {kn:.3f} ms  cov {cv:.3f} ms  voxelmap {vm:.3f} ms")
