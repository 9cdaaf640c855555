#!/usr/bin/env python3
"""Generates tests/golden/ref_small.npz with the REFERENCE'S OWN CODE: the outputs come from oracle/_ref/libglim_ref.so, i.e.
/root/reference/src/glim/common/cloud_covariance_estimation.cpp and cloud_deskewing.cpp compiled unmodified (oracle/Makefile target `ref`;
stand-in Eigen / GTSAM headers under oracle/ref_standin/, see there for what that does and does not pin).

This is the reference pin of SURVEY.md 8a row a2 (covariance + normal) and 8f rank 2 (deskewing, and the composed chain
deskew -> IMU frame -> covariance of odometry_estimation_imu.cpp:313-320 through oracle/ref_shim.cpp `ref_frontend`): the restatement in oracle/vgicp_oracle.c and
the HIP kernels are tested against these vectors (tests/test_ref.py).  It can only be regenerated where /root/reference exists:
    python tests/golden/make_golden_ref.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LINEAR_VEL, ANGULAR_VEL = [4.0, -2.0, 0.3], [0.1, -0.2, 1.5]
STAMP = 100.0


def make_inputs(orc):
    from glim_amd import synth

    scene = synth.Scene.default()
    rng = np.random.default_rng(2024)
    dirs = synth.lidar_directions(24, 160)
    pts = synth.scan(scene, synth.arc_trajectory(3)[2], dirs, frame_id=7)  # FP32-representable sensor-frame points
    # a few hard neighbourhoods: exact duplicates (isotropic: identity eigenvectors), a collinear run (two equal eigenvalues), a planar lattice
    extra = np.array([[1.0, 2.0, 0.5]] * 12 + [[3.0 + 0.01 * i, -1.0, 0.25] for i in range(12)] +
                     [[-2.0 + 0.02 * (i % 4), 1.0 + 0.02 * (i // 4), 0.0] for i in range(16)], dtype=np.float32)
    pts = np.concatenate([pts, extra]).astype(np.float32)
    nbrs = orc.knn(pts, 10)
    times = np.sort(rng.uniform(0.0, 0.1, len(pts)))
    imu_times = STAMP + np.array([-0.02, 0.013, 0.031, 0.058, 0.09, 0.13])
    imu_poses = [orc.se3_exp(rng.normal(size=6) * [0.3, 0.3, 0.3, 2, 2, 2])]
    for _ in imu_times[1:]:
        imu_poses.append(imu_poses[-1] @ orc.se3_exp(rng.normal(size=6) * [0.02, 0.02, 0.05, 0.05, 0.05, 0.02]))
    return dict(points=pts, neighbors=nbrs, times=times, T_imu_lidar=orc.se3_exp([0.05, 0.02, -0.1, 0.2, -0.1, 0.05]), imu_times=imu_times,
                imu_poses=np.stack(imu_poses))


def compute(orc, inp, ref):
    out = {}
    for k in (10, 5):
        n, c = orc.covariances(inp["points"], inp["neighbors"], k_neighbors=k, ref=ref)
        out[f"normals_k{k}"], out[f"covs_k{k}"] = n, c
    p, t, Til = inp["points"].astype(np.float64), inp["times"], inp["T_imu_lidar"]
    out["deskew_constvel"] = orc.deskew(p, t, Til, linear_vel=LINEAR_VEL, angular_vel=ANGULAR_VEL, ref=ref)
    out["deskew_constvel_still"] = orc.deskew(p, t, Til, linear_vel=[0.5, 0, 0], angular_vel=[0, 0, 0], ref=ref)  # theta^2 <= eps branch of Expmap
    out["deskew_imu"] = orc.deskew(p, t, Til, imu_times=inp["imu_times"], imu_poses=list(inp["imu_poses"]), stamp=STAMP, ref=ref)
    out["deskew_imu_short_track"] = orc.deskew(p, t, Til, imu_times=inp["imu_times"][:2], imu_poses=list(inp["imu_poses"][:2]), stamp=STAMP, ref=ref)
    # the composed chain between preprocessing and create_frame (odometry_estimation_imu.cpp:313-320): deskew -> pt = T_imu_lidar * pt ->
    # covariances from the raw scan's neighbours, all on FP64 points; the LiDAR-frame variant (middle step skipped) shows what the step changes
    for name, kw in (("frontend_imu", dict(imu_times=inp["imu_times"], imu_poses=list(inp["imu_poses"]), stamp=STAMP)),
                     ("frontend_constvel", dict(linear_vel=LINEAR_VEL, angular_vel=ANGULAR_VEL))):
        for frame in ("imuframe", "lidarframe"):
            pts, nrm, cov = orc.frontend(p, t, inp["neighbors"], Til, to_imu_frame=(frame == "imuframe"), ref=ref, **kw)
            out[f"{name}.{frame}.points"], out[f"{name}.{frame}.normals"], out[f"{name}.{frame}.covs"] = pts, nrm, cov
    return out


def main():
    from oracle import oracle as orc

    assert orc.ref_lib() is not None, "oracle/_ref could not be built: /root/reference is needed to regenerate this fixture"
    inp = make_inputs(orc)
    out = compute(orc, inp, ref=True)
    np.savez_compressed(os.path.join(HERE, "ref_small.npz"), **inp, **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
