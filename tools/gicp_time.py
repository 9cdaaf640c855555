"""Device GICP factor: index build + linearize time (GPU box)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glim_amd import api, synth
import ctypes as C
ctx = api.Context(0, 1)
scene = synth.Scene.default()
for rings, az, label in ((32, 384, "12k-pt frames (between factors, sub_mapping.cpp:202)"), (64, 1024, "65k-pt submaps (global_mapping.cpp:400)")):
    poses = synth.arc_trajectory(2, step=0.6, yaw_step_deg=3.0)
    gs = []
    for i, T in enumerate(poses):
        g = api.PointCloudGPU.clone(synth.scan(scene, T, synth.lidar_directions(rings, az), i), ctx=ctx)
        g.find_neighbors(10, download=False); g.estimate_covariances(10); gs.append(g)
    delta = np.linalg.inv(poses[0]) @ poses[1]
    for max_d in (1.0, 0.5):
        t0 = time.perf_counter()
        for _ in range(10):
            f = api.IntegratedGICPFactor(np.eye(4), 1, gs[0], gs[1], max_correspondence_distance=max_d); f.close()
        build = (time.perf_counter() - t0) / 10 * 1e3
        f = api.IntegratedGICPFactor(np.eye(4), 1, gs[0], gs[1], max_correspondence_distance=max_d)
        L = f.linearize({1: delta})
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); f.linearize({1: delta}); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"{label}: N={gs[1].size()} max_d={max_d}: index build {build:.3f} ms, linearize p50 {np.median(ts):.3f} ms, inliers {L['num_inliers']}")
