#!/usr/bin/env python3
"""Generates tests/golden/frontend_small.npz from the CPU oracle: the SURVEY.md 8f rows (scan preprocessing, deskewing, submap merge,
GICP factor) on small seeded inputs.  Like vgicp_small.npz this is a REGRESSION pin of the oracle and an oracle-free expectation for
the GPU tests, NOT a reference pin.  Regenerate with:  python tests/golden/make_golden_frontend.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PRE_RANDOM = dict(downsample_target=1000, seed=5)
PRE_VOXEL = dict(use_random_grid_downsampling=0, downsample_resolution=0.5)
T_IMU_LIDAR_XI = [0.05, 0.02, -0.1, 0.2, -0.1, 0.05]
LINEAR_VEL, ANGULAR_VEL = [4.0, -2.0, 0.3], [0.1, -0.2, 1.5]
MERGE_RES, MERGE_TARGET, MERGE_SEED = 0.25, 800, 3
GICP_MAX_D = 1.0


def make_inputs(orc):
    from glim_amd import synth

    scene = synth.Scene.default()
    rng = np.random.default_rng(11)
    # raw scan in the sensor frame (FP32-representable), shuffled firing order, a few invalid points
    dirs = synth.lidar_directions(16, 256)
    pts = synth.scan(scene, synth.arc_trajectory(2)[0], dirs, frame_id=3).astype(np.float64)
    times = np.sort(rng.uniform(0.0, 0.1, len(pts)))
    perm = rng.permutation(len(pts))
    pts, times = pts[perm], times[perm]
    inten = rng.uniform(0, 255, len(pts)).astype(np.float32).astype(np.float64)
    pts[17] = [np.nan, 0.0, 1.0]
    pts[2042] = [1.0, np.inf, 1.0]
    # IMU track for the slerp form of the deskewing
    imu_times = 100.0 + np.array([-0.02, 0.013, 0.031, 0.058, 0.09, 0.13])
    imu_poses = [orc.se3_exp(rng.normal(size=6) * [0.3, 0.3, 0.3, 2, 2, 2])]
    for _ in imu_times[1:]:
        imu_poses.append(imu_poses[-1] @ orc.se3_exp(rng.normal(size=6) * [0.02, 0.02, 0.05, 0.05, 0.05, 0.02]))
    # keyframes of a submap: points + FP32-rounded kNN covariances + poses in the submap origin
    kdirs = synth.lidar_directions(16, 64)
    kposes = synth.arc_trajectory(3, step=0.8, yaw_step_deg=3.0)
    origin = np.linalg.inv(kposes[1])
    kp, kc, kT = [], [], []
    for i, T in enumerate(kposes):
        p = synth.scan(scene, T, kdirs, frame_id=20 + i).astype(np.float64)
        _, c = orc.covariances(p, orc.knn(p, 10))
        kp.append(p)
        kc.append(c.astype(np.float32))
        kT.append(origin @ T)
    n_key = min(len(p) for p in kp)  # equal sizes so that the frames stack into one array
    return dict(raw_points=pts, raw_times=times, raw_intensities=inten, T_imu_lidar=orc.se3_exp(T_IMU_LIDAR_XI), imu_times=imu_times, imu_poses=np.stack(imu_poses),
                key_points=np.stack([p[:n_key] for p in kp]), key_covs=np.stack([c[:n_key] for c in kc]), key_poses=np.stack(kT))


def compute(orc, inp, gicp_pair):
    """Everything the fixture stores, from the stored inputs (used by the generator and by the CPU regression test)."""
    out = {}
    pts, times, inten = inp["raw_points"], inp["raw_times"], inp["raw_intensities"]
    for tag, kw in (("pre_random", PRE_RANDOM), ("pre_voxel", PRE_VOXEL)):
        r = orc.preprocess(pts, times, inten, orc.preprocess_params(**kw))
        out[f"{tag}_points"], out[f"{tag}_times"], out[f"{tag}_intensities"] = r["points"], r["times"], r["intensities"]
        # neighbours on the FP32 image of the cloud (what the device's kNN sees; identical for samplers that select points)
        out[f"{tag}_neighbors"] = orc.knn(r["points"].astype(np.float32).astype(np.float64), 10)
    Til = inp["T_imu_lidar"]
    p, t = out["pre_random_points"], out["pre_random_times"]
    out["deskew_constvel"] = orc.deskew(p, t, Til, linear_vel=LINEAR_VEL, angular_vel=ANGULAR_VEL)
    out["deskew_imu"] = orc.deskew(p, t, Til, imu_times=inp["imu_times"], imu_poses=list(inp["imu_poses"]), stamp=100.0)
    poses, kp, kc = list(inp["key_poses"]), list(inp["key_points"]), [c.astype(np.float64) for c in inp["key_covs"]]
    out["merge_all_points"], out["merge_all_covs"] = orc.merge_frames(poses, kp, kc, MERGE_RES)
    out["merge_target_points"], out["merge_target_covs"] = orc.merge_frames(poses, kp, kc, MERGE_RES, target_num_points=MERGE_TARGET, seed=MERGE_SEED)
    tp, tc, sp, sc, delta = gicp_pair
    L = orc.gicp_linearize(tp, tc, sp, sc, delta, GICP_MAX_D, num_threads=1, want_corr=True)
    for k in ("H_tt", "H_ss", "H_ts", "b_t", "b_s", "corr"):
        out[f"gicp_{k}"] = L[k]
    out["gicp_error"], out["gicp_num_inliers"] = np.float64(L["error"]), np.int64(L["num_inliers"])
    return out


def gicp_pair_from(gold_small):
    return (gold_small["target_points"], gold_small["target_covs"].astype(np.float64), gold_small["source_points"],
            gold_small["source_covs"].astype(np.float64), gold_small["delta"])


def main():
    from oracle import oracle as orc

    inp = make_inputs(orc)
    out = compute(orc, inp, gicp_pair_from(dict(np.load(os.path.join(HERE, "vgicp_small.npz")))))
    np.savez_compressed(os.path.join(HERE, "frontend_small.npz"), **inp, **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
