#!/usr/bin/env python3
"""Generates tests/golden/vgicp_small.npz from the CPU oracle (oracle/) on a seeded synthetic pair.

NOT a reference pin: the reference (koide3/glim + gtsam_points) ships no golden vectors and cannot be built or imported in this
image, so this fixture pins the oracle against itself (regression) and gives the GPU tests a committed,
oracle-independent-at-run-time set of expected values.  Regenerate with:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glim_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    scene = synth.Scene.default()
    dirs = synth.lidar_directions(16, 96)  # 1536 rays
    poses = synth.arc_trajectory(2)
    tgt = synth.scan(scene, poses[0], dirs, 0)
    src = synth.scan(scene, poses[1], dirs, 1)
    nb_t, nb_s = orc.knn(tgt, 10), orc.knn(src, 10)
    nt, ct = orc.covariances(tgt, nb_t)
    ns, cs = orc.covariances(src, nb_s)
    ct32, cs32 = ct.astype(np.float32), cs.astype(np.float32)
    delta = synth.relative_pose(poses[0], poses[1]) @ orc.se3_exp([0.003, -0.002, 0.004, 0.02, -0.03, 0.01])
    out = {"target_points": tgt, "source_points": src, "target_covs": ct32, "source_covs": cs32, "source_normals": ns.astype(np.float32),
           "source_neighbors": nb_s, "delta": delta}
    for res in (0.5, 1.0):
        vm = orc.VoxelMap(res).insert(tgt, ct32.astype(np.float64))
        coords, counts, means, covs = vm.voxels()
        L = orc.vgicp_linearize(vm, src, cs32.astype(np.float64), delta, num_threads=1, want_corr=True)
        tag = f"r{int(res * 100):03d}"
        out[f"{tag}_voxel_coords"] = coords
        out[f"{tag}_voxel_counts"] = counts
        out[f"{tag}_voxel_means"] = means
        out[f"{tag}_voxel_covs"] = covs
        for k in ("H_tt", "H_ss", "H_ts", "b_t", "b_s"):
            out[f"{tag}_{k}"] = L[k]
        out[f"{tag}_error"] = np.float64(L["error"])
        out[f"{tag}_num_inliers"] = np.int64(L["num_inliers"])
        out[f"{tag}_corr"] = L["corr"]
        out[f"{tag}_overlap"] = np.float64(orc.overlap(vm, src, delta))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "vgicp_small.npz"), **out)
    print("wrote vgicp_small.npz:", {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith("r050")})


if __name__ == "__main__":
    main()
