import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glim_amd import api, synth
ctx = api.Context(0, 1)
pts = synth.scan(synth.Scene.default(), synth.arc_trajectory(1)[0], synth.lidar_directions(128, 1024), 0)
g = api.PointCloudGPU.clone(pts, ctx=ctx)
g.find_neighbors(10, download=False)
g.find_neighbors(10, download=False)
d = np.fromfile(os.environ["GLIM_AMD_KNN_DEBUG"], dtype=np.int32).reshape(-1, 4)
for name, col in (("candidates", 0), ("probes", 1), ("last ring", 2), ("time us", 3)):
    v = d[:, col] * (0.01 if col == 3 else 1.0)
    print(f"{name}: mean {v.mean():.1f} p50 {np.median(v):.1f} p90 {np.percentile(v,90):.1f} p99 {np.percentile(v,99):.1f} p99.9 {np.percentile(v,99.9):.1f} max {v.max():.1f}")
slow = np.argsort(-d[:, 3])[:8]
print("slowest queries: cand, probes, ring, ticks", d[slow].tolist())
print("corr(time, cand)", np.corrcoef(d[:,3], d[:,0])[0,1], "corr(time, probes)", np.corrcoef(d[:,3], d[:,1])[0,1])
