"""Wall time of the device scan preprocessing (glim_amd_preprocess) vs the CPU oracle on the same raw scans, GPU box."""
import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from glim_amd import api, synth
from oracle import oracle as orc
ctx = api.Context(0, 1)
def raw(n_rings, n_az, seed=0):
    scene = synth.Scene.default()
    pts = synth.scan(scene, synth.arc_trajectory(1)[0], synth.lidar_directions(n_rings, n_az), seed).astype(np.float64)
    rng = np.random.default_rng(seed)
    return pts, np.sort(rng.uniform(0, 0.1, len(pts))), rng.uniform(0, 255, len(pts))
def t(f, reps=30):
    f(); ctx.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ctx.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.sort(ts)
    return ts[len(ts) // 2], ts[-1]
for name, (pts, times, inten) in (("lidar131k", raw(128, 1024)), ("lidar524k", raw(128, 4096))):
    p4 = np.ones((len(pts), 4)); p4[:, :3] = pts
    for label, kw in (("random 1m -> 10k (shipped)", dict()), ("random 0.5m rate 0.5", dict(downsample_target=0, downsample_rate=0.5, downsample_resolution=0.5)),
                      ("voxelgrid 0.25m", dict(use_random_grid_downsampling=0, downsample_resolution=0.25)),
                      ("voxelgrid 0.25m, no knn", dict(use_random_grid_downsampling=0, downsample_resolution=0.25, k_correspondences=0)),
                      ("random shipped, no knn", dict(k_correspondences=0))):
        prm = api.preprocess_params(**kw)
        out = api.PointCloudGPU.preprocess(p4, times, inten, prm, ctx=ctx)
        ms, mx = t(lambda: api.PointCloudGPU.preprocess(p4, times, inten, prm, ctx=ctx).close())
        oprm = orc.preprocess_params(**kw)
        t0 = time.perf_counter(); ref = orc.preprocess(pts, times, inten, oprm, neighbors=kw.get("k_correspondences", 10) > 0); cpu = (time.perf_counter() - t0) * 1e3
        print(f"{name} N={len(pts)} {label}: out {out.size()} (oracle {len(ref['points'])})  device p50 {ms:.3f} ms max {mx:.3f} ms  cpu-oracle {cpu:.1f} ms")
