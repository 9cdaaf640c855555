#!/usr/bin/env python3
"""Generates tests/golden/ref_preprocess.npz with the REFERENCE'S OWN CloudPreprocessor: the outputs come from oracle/_ref/libglim_ref.so, i.e.
/root/reference/src/glim/preprocess/cloud_preprocessor.cpp compiled unmodified (oracle/Makefile target `ref`, oracle/ref_preprocess_shim.cpp).
The three gtsam_points sampling algorithms that file calls are not part of the reference tree and are answered by the oracle's restatements
(oracle/ref_standin/gtsam_points/types/point_cloud_cpu.hpp); everything else -- stage order, distance / cropbox predicates, time sort,
global-shutter rule, frame assembly, neighbour layout (cloud_preprocessor.cpp:92-221) -- is the reference's code.

Reference pin of the in-tree half of SURVEY.md 8f rank 1 and of the row-a1 contract.  Regenerate (only where /root/reference exists):
    python tests/golden/make_golden_ref_preprocess.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

T_IMU_LIDAR_XI = [0.1, -0.2, 0.3, 1.0, 2.0, 0.5]
CASES = {
    "voxelgrid": dict(use_random_grid_downsampling=0, downsample_resolution=0.5),
    "randomgrid": dict(use_random_grid_downsampling=1, downsample_target=600, downsample_resolution=1.0, seed=5),
    "randomgrid_rate": dict(use_random_grid_downsampling=1, downsample_target=0, downsample_rate=0.2, downsample_resolution=0.8, seed=9),
    "range_window": dict(use_random_grid_downsampling=0, downsample_resolution=0.4, distance_near_thresh=5.0, distance_far_thresh=25.0, k_correspondences=8),
    "cropbox_lidar": dict(use_random_grid_downsampling=0, downsample_resolution=0.5, enable_cropbox_filter=1, crop_bbox_min=(-5, -5, -2), crop_bbox_max=(5, 5, 2)),
    "cropbox_imu": dict(use_random_grid_downsampling=0, downsample_resolution=0.5, enable_cropbox_filter=1, crop_bbox_frame_imu=1, crop_bbox_min=(-5, -5, -2),
                        crop_bbox_max=(5, 5, 2)),
    "global_shutter": dict(use_random_grid_downsampling=0, downsample_resolution=0.5, global_shutter=1),
    "outliers": dict(use_random_grid_downsampling=0, downsample_resolution=0.5, enable_outlier_removal=1, outlier_removal_k=10, outlier_std_mul_factor=1.0),
}


def make_inputs():
    rng = np.random.default_rng(77)
    n = 2400
    pts = rng.uniform(-30, 30, size=(n, 3))
    pts[:, 2] *= 0.2
    pts[:40] = rng.uniform(-0.5, 0.5, size=(40, 3))  # inside the near threshold
    pts[40:60] *= 10.0                               # beyond the far threshold
    pts[60] = [np.nan, 1.0, 2.0]                     # non-finite points are dropped
    pts[61] = [3.0, np.inf, 2.0]
    pts = pts.astype(np.float32).astype(np.float64)  # FP32-representable, like a sensor driver's output: the device kNN works on the FP32 image
    times = rng.permutation(np.sort(rng.uniform(0.0, 0.1, n)))  # distinct stamps in arrival order (std::sort leaves equal stamps unordered)
    return dict(points=pts, times=times, intensities=rng.uniform(0, 255, n))


def params_of(orc, name):
    kw = dict(CASES[name])
    if kw.get("crop_bbox_frame_imu"):
        kw["T_imu_lidar"] = orc.se3_exp(np.array(T_IMU_LIDAR_XI))
    return orc.preprocess_params(**kw)


def compute(orc, inp, ref):
    out = {}
    for name in CASES:
        r = orc.preprocess(inp["points"], inp["times"], inp["intensities"], params_of(orc, name), ref=ref)
        for k in ("points", "times", "intensities", "neighbors"):
            out[f"{name}.{k}"] = r[k]
    return out


def main():
    from oracle import oracle as orc

    assert orc.ref_lib() is not None and hasattr(orc.ref_lib(), "ref_preprocess"), "oracle/_ref could not be built: /root/reference is needed"
    inp = make_inputs()
    out = compute(orc, inp, ref=True)
    np.savez_compressed(os.path.join(HERE, "ref_preprocess.npz"), **inp, **out)
    print({k: v.shape for k, v in out.items() if k.endswith(".points")})


if __name__ == "__main__":
    main()
