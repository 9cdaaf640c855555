"""Per-wavefront work counters of the chunk kNN kernel (GLIM_AMD_KNN_DEBUG=<file>), GPU box."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glim_amd import api, synth
ctx = api.Context(0, 1)
which = sys.argv[1] if len(sys.argv) > 1 else "lidar"
if which == "lidar":
    pts = synth.scan(synth.Scene.default(), synth.arc_trajectory(1)[0], synth.lidar_directions(128, 1024), 0)
else:
    pts = synth.scan(synth.Scene.small_room(), synth.pose(-2.5, -1.5, 1.4, 0.5), synth.pinhole_directions(640, 480, 70, 55), 0, sigma=0.002, max_range=8.0, min_range=0.3)
g = api.PointCloudGPU.clone(pts, ctx=ctx)
g.find_neighbors(10, download=False)
g.find_neighbors(10, download=False)
d = np.fromfile(os.environ["GLIM_AMD_KNN_DEBUG"], dtype=np.int32).reshape(-1, 4)
for name, col, sc in (("tiles", 0, 1.0), ("insert rounds", 1, 1.0), ("wave time us", 2, 0.01)):
    v = d[:, col] * sc
    print(f"{name}: mean {v.mean():.1f} p50 {np.median(v):.1f} p90 {np.percentile(v,90):.1f} p99 {np.percentile(v,99):.1f} max {v.max():.1f}")
worst = np.argsort(-d[:, 2])[:6]
print("slowest wavefronts (chunk, tiles, rounds, us, box extent m):", [(int(c), int(d[c, 0]), int(d[c, 1]), round(d[c, 2] * 0.01, 1), d[c, 3] / 1000.0) for c in worst])
print("box extent m: p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(d[:, 3] / 1000.0, [50, 90, 99, 100])))
print("corr(time, tiles) %.2f  corr(time, rounds) %.2f" % (np.corrcoef(d[:, 2], d[:, 0])[0, 1], np.corrcoef(d[:, 2], d[:, 1])[0, 1]))
# where a wavefront's time goes: least-squares fit  time = c0 + a * tiles + b * rounds  (c0: what every wavefront pays -- the walk over the groups)
A = np.stack([np.ones(len(d)), d[:, 0], d[:, 1]], axis=1).astype(np.float64)
t_us = d[:, 2] * 0.01
coef, *_ = np.linalg.lstsq(A, t_us, rcond=None)
res = t_us - A @ coef
print("fit: time_us = %.1f + %.2f * tiles + %.3f * rounds   (R^2 %.3f; mean split: fixed %.1f, tiles %.1f, rounds %.1f us)" %
      (coef[0], coef[1], coef[2], 1.0 - res.var() / t_us.var(), coef[0], coef[1] * d[:, 0].mean(), coef[2] * d[:, 1].mean()))
if len(sys.argv) > 2:
    np.save(sys.argv[2], d)
