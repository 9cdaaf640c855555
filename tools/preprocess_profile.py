"""One workload for a rocprofv3 kernel trace of the device scan preprocessing: 40 calls of glim_amd_preprocess on a raw 131 072-pt scan with the shipped
parameters (random grid -> 10 000 points, kNN k = 10).   cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python tools/preprocess_profile.py"""
import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from glim_amd import api, synth
ctx = api.Context(0, 1)
scene = synth.Scene.default()
pts = synth.scan(scene, synth.arc_trajectory(1)[0], synth.lidar_directions(128, 1024), 0).astype(np.float64)
rng = np.random.default_rng(0)
times, inten = np.sort(rng.uniform(0, 0.1, len(pts))), rng.uniform(0, 255, len(pts))
p4 = np.ones((len(pts), 4)); p4[:, :3] = pts
prm = api.preprocess_params()
for _ in range(5):
    api.PointCloudGPU.preprocess(p4, times, inten, prm, ctx=ctx).close()
ts = []
for _ in range(40):
    t0 = time.perf_counter(); api.PointCloudGPU.preprocess(p4, times, inten, prm, ctx=ctx).close(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"preprocess 131072 -> 10000 (+kNN): p50 {np.median(ts):.3f} ms, min {min(ts):.3f} ms")
