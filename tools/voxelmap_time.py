"""Wall time per GaussianVoxelMapGPU::insert (direct build: tables, keys + sums, records; one synchronise) at the odometry's frame size and at a
full scan; run under rocprofv3 --kernel-trace --stats for the kernel times behind it."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from glim_amd import api, synth

ctx = api.Context(0, 1)
scene = synth.Scene.default()
full = synth.scan(scene, synth.arc_trajectory(1)[0], synth.lidar_directions(128, 1024), 0)
rng = np.random.default_rng(3)
for name, pts, res in (("10000 pts, 0.443 m", full[np.sort(rng.choice(len(full), 10000, replace=False))], 0.443), ("10000 pts, 0.886 m", None, 0.886),
                       ("131072 pts, 0.5 m", full, 0.5), ("131072 pts, 1.0 m", None, 1.0)):
    if pts is not None:
        g = api.PointCloudGPU.clone(pts, ctx=ctx)
        g.find_neighbors(10, download=False)
        g.estimate_covariances(10)
    for _ in range(3):
        api.GaussianVoxelMapGPU(res, ctx=ctx).insert(g).close()
    ts = []
    for _ in range(40):
        m = api.GaussianVoxelMapGPU(res, ctx=ctx)
        t = time.perf_counter(); m.insert(g); ts.append(time.perf_counter() - t)
        nv = m.voxelmap_info()["num_voxels"]
        m.close()
    print(f"{name}: insert p50 {np.median(ts) * 1e6:.1f} us  min {np.min(ts) * 1e6:.1f} us  voxels {nv}", flush=True)
