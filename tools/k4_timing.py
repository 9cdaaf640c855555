#!/usr/bin/env python3
"""Per-block timeline of the fused VGICP kernel on the bench workload (GPU box).  Needs a library built with -DGLIM_AMD_K4_TIMING=1
(tools/ab_variant.sh timing -DGLIM_AMD_K4_TIMING=1; GLIM_AMD_LIB=build/ab/timing/libglim_amd.so): every block leaves its start / end time
(s_memrealtime, 10 ns) and its XCC / SE / CU in the spare slots of its partial row.  Prints, per XCD and overall: when blocks start and end
relative to the first start, block durations, and how many blocks each CU ran -- i.e. whether the launch is one balanced resident set."""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from glim_amd import api, synth  # noqa: E402

F = int(os.environ.get("KEXP_FACTORS", "128"))
dump = os.environ.setdefault("GLIM_AMD_K4_TIMING_DUMP", "/tmp/k4_timing.bin")
ctx = api.Context(0, 1)
yaw0 = math.radians(10.0)
radius = 0.5 / math.radians(2.0)
poses = synth.arc_trajectory(F + 1, start=(-2.0 + radius * math.sin(yaw0), -0.5 - radius * math.cos(yaw0), 1.8), yaw0_deg=10.0)
clouds = bench.make_frames(api, ctx, poses, 128, 1024)
deltas = np.stack([api.pose12(synth.relative_pose(poses[i], poses[i + 1])) for i in range(F)])
vmaps = [api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(x) for x in clouds[:F]]
fset = api.NonlinearFactorSetGPU(ctx)
for i in range(F):
    fset.add(api.IntegratedVGICPFactorGPU(i, i + 1, vmaps[i], clouds[i + 1]))
for combo in json.loads(os.environ.get("KEXP", "null")) or [{}]:
    for k, v in combo.items():
        os.environ[k] = str(v)
    fset2 = api.NonlinearFactorSetGPU(ctx)  # a fresh plan picks the environment up
    for i in range(F):
        fset2.add(api.IntegratedVGICPFactorGPU(i, i + 1, vmaps[i], clouds[i + 1]))
    ms_k, _ = fset2.profile(deltas, iters=20)
    rec = np.fromfile(dump, dtype=np.uint32).reshape(-1, 5)
    rec = rec[rec[:, 0] != 0xffffffff]
    t0 = rec[:, 2].astype(np.int64)
    t1 = rec[:, 3].astype(np.int64)
    base = t0.min()
    start, end = (t0 - base) * 0.01, (t1 - base) * 0.01  # us
    dur = end - start
    xcc = (rec[:, 4] >> 16) & 0xf
    hw = rec[:, 4] & 0xffff
    cu = (xcc.astype(np.int64) << 16) | (hw & 0xff00)  # SE_ID[15:13] SH_ID[12] CU_ID[11:8]
    out = {"combo": combo, "kernel_us_events": round(ms_k * 1e3, 1), "blocks": int(len(rec)), "span_us": round(float(end.max()), 1),
           "start_us": {"p50": round(float(np.median(start)), 1), "p90": round(float(np.percentile(start, 90)), 1), "max": round(float(start.max()), 1)},
           "end_us": {"min": round(float(end.min()), 1), "p10": round(float(np.percentile(end, 10)), 1), "p50": round(float(np.median(end)), 1), "max": round(float(end.max()), 1)},
           "duration_us": {"min": round(float(dur.min()), 1), "p50": round(float(np.median(dur)), 1), "mean": round(float(dur.mean()), 1), "max": round(float(dur.max()), 1)},
           "blocks_per_cu": {"cus": int(len(np.unique(cu))), "min": int(np.bincount(np.unique(cu, return_inverse=True)[1]).min()),
                             "max": int(np.bincount(np.unique(cu, return_inverse=True)[1]).max())},
           "per_xcd": {}}
    for x in sorted(np.unique(xcc)):
        m = xcc == x
        out["per_xcd"][int(x)] = {"blocks": int(m.sum()), "first_start": round(float(start[m].min()), 1), "last_start": round(float(start[m].max()), 1),
                                  "first_end": round(float(end[m].min()), 1), "last_end": round(float(end[m].max()), 1), "mean_dur": round(float(dur[m].mean()), 1),
                                  "factors": int(len(np.unique(rec[m, 0])))}
    # where does the spread of block durations come from?  spread of the per-factor / per-CU means, and the spread left inside a factor
    fac = rec[:, 0].astype(np.int64)
    fmean = np.array([dur[fac == f].mean() for f in np.unique(fac)])
    cmean = np.array([dur[cu == c].mean() for c in np.unique(cu)])
    within = np.array([dur[fac == f].max() - dur[fac == f].min() for f in np.unique(fac)])
    se = (hw >> 13) & 0x7
    out["per_factor_mean_dur"] = {"min": round(float(fmean.min()), 1), "p50": round(float(np.median(fmean)), 1), "max": round(float(fmean.max()), 1)}
    out["per_cu_mean_dur"] = {"min": round(float(cmean.min()), 1), "p50": round(float(np.median(cmean)), 1), "max": round(float(cmean.max()), 1)}
    out["within_factor_range"] = {"p50": round(float(np.median(within)), 1), "max": round(float(within.max()), 1)}
    out["per_se_mean_dur"] = {f"{int(x)}.{int(e)}": round(float(dur[(xcc == x) & (se == e)].mean()), 1) for x in sorted(np.unique(xcc))[:2] for e in sorted(np.unique(se))}
    slow = np.argsort(-dur)[:8]
    out["slowest"] = [{"factor": int(fac[i]), "chunk": int(rec[i, 1]), "xcc": int(xcc[i]), "hw": hex(int(hw[i])), "dur": round(float(dur[i]), 1)} for i in slow]
    # is a factor's block time predictable from what the plan knows?  least-squares fit  dur_f ~ a + b * V_f / N_f  (V: voxels of the target map)
    nv = np.array([v.voxelmap_info()["num_voxels"] for v in vmaps], dtype=np.float64)
    npt = np.array([c.size() for c in clouds[1:]], dtype=np.float64)
    ufac = np.unique(fac)
    x = nv[ufac] / npt[ufac]
    A = np.stack([np.ones_like(x), x], axis=1)
    coef, res, _, _ = np.linalg.lstsq(A, fmean, rcond=None)
    pred = A @ coef
    out["fit_dur_vs_voxels_per_point"] = {"a": round(float(coef[0]), 2), "b": round(float(coef[1]), 2), "r2": round(float(1 - ((fmean - pred) ** 2).sum() / ((fmean - fmean.mean()) ** 2).sum()), 3),
                                          "x_min": round(float(x.min()), 4), "x_max": round(float(x.max()), 4)}
    out["per_factor"] = [[int(f), int(nv[f]), round(float(m), 1)] for f, m in zip(ufac, fmean)][::4]
    out.pop("per_xcd")
    print(json.dumps(out), flush=True)
    for k in combo:
        os.environ.pop(k, None)
    fset2.close()
