"""Wall time of the device submap merge (glim_amd_merge_frames) vs the CPU oracle, GPU box."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glim_amd import api, synth
from oracle import oracle as orc
ctx = api.Context(0, 1)
scene = synth.Scene.default()
for rings, az in ((32, 384), (64, 1024)):
    poses = synth.arc_trajectory(15, step=0.8, yaw_step_deg=3.0)
    origin = np.linalg.inv(poses[7])
    pts, covs, rel = [], [], []
    for i, T in enumerate(poses):
        p = synth.scan(scene, T, synth.lidar_directions(rings, az), i).astype(np.float64)
        g = api.PointCloudGPU.clone(p, ctx=ctx); g.find_neighbors(10, download=False); g.estimate_covariances(10)
        _, c, _ = g.download(covs=True, normals=False)
        pts.append(p); covs.append(c.astype(np.float64)); rel.append(origin @ T)
    for res, target in ((0.1, 50000), (0.25, -1)):
        packed = api._pack_frames(rel, pts, covs)
        out = api.merge_frames(rel, pts, covs, res, target_num_points=target, ctx=ctx)
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); api.merge_frames(None, None, None, res, target_num_points=target, ctx=ctx, packed=packed).close(); ts.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter(); rp, rc = orc.merge_frames(rel, pts, covs, res, target_num_points=target); cpu = (time.perf_counter() - t0) * 1e3
        gp, gc = out.download_merged()
        print(f"15 x {len(pts[0])} pts, res {res}, target {target}: merged {out.size()} (oracle {len(rp)}, equal={np.array_equal(gp, rp) and np.array_equal(gc, rc)})"
              f"  device p50 {np.median(ts):.2f} ms  (C call incl. the host-array upload)  cpu-oracle {cpu:.1f} ms")
