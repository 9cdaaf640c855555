#!/usr/bin/env python3
"""Generates tests/golden/hip_rows.npz ON THE GPU BOX: the compact [n x 29] records the HIP factor kernels produced for the 12 ordered pairs of the
4-scan problem of tests/test_multi_cpu.py (device-estimated covariances, 1.0 m maps, binary factors).  The CPU gloo test shards and reduces THESE
rows -- what the N > 1 path really moves -- instead of rows converted from oracle results.  Run:  python tests/golden/make_golden_hip_rows.py
(writes gpurun_out/hip_rows.npz; copy it to tests/golden/)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glim_amd import api, synth  # noqa: E402


def problem():
    scene = synth.Scene.default()
    dirs = synth.lidar_directions(16, 64)
    poses = synth.arc_trajectory(4)
    scans = [synth.scan(scene, T, dirs, i) for i, T in enumerate(poses)]
    pairs = [(i, j) for i in range(4) for j in range(4) if i != j]
    deltas = np.stack([api.pose12(synth.relative_pose(poses[i], poses[j])) for i, j in pairs])
    return scans, pairs, deltas


def main():
    import ctypes as C

    scans, pairs, deltas = problem()
    ctx = api.Context(0, 1)
    clouds, covs = [], []
    for s in scans:
        g = api.PointCloudGPU.clone(s, ctx=ctx)
        g.find_neighbors(10, download=False)
        g.estimate_covariances(10)
        clouds.append(g)
        covs.append(g.download(normals=False)[1])
    maps = [api.GaussianVoxelMapGPU(1.0, ctx=ctx).insert(c) for c in clouds]
    fset = api.NonlinearFactorSetGPU(ctx)
    for i, j in pairs:
        fset.add(api.IntegratedVGICPFactorGPU(i, j, maps[i], clouds[j]))
    hip = C.CDLL("libamdhip64.so")
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), C.c_size_t(len(pairs) * 29 * 8)) == 0
    fset.linearize_device_async(deltas, d.value, 0)
    ctx.synchronize()
    rows = np.zeros((len(pairs), 29))
    assert hip.hipMemcpy(rows.ctypes.data_as(C.c_void_p), d, C.c_size_t(rows.nbytes), 2) == 0
    full = fset.linearize_poses(deltas)
    out = os.path.join(ROOT, "gpurun_out", "hip_rows.npz")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, rows=rows, deltas=deltas, covs=np.stack(covs), H_tt=np.stack([f["H_tt"] for f in full]), H_ts=np.stack([f["H_ts"] for f in full]),
                        b_t=np.stack([f["b_t"] for f in full]))
    print("wrote", out, rows.shape)


if __name__ == "__main__":
    main()
