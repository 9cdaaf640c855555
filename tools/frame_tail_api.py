#!/usr/bin/env python3
"""Companion of tools/frame_tail_trace.py: WHICH runtime calls sit in the slow frames of the live odometry loop?
  rocprofv3 --kernel-trace --hip-trace --hsa-trace --output-format csv -d <prof> -- <dir>/odometry_frame_loop <dir>/scene.bin 300 3 1
  python tools/frame_tail_api.py <prof> [out.json]
Frames are cut at the frame_build_kernel launches (kernel trace); for every HIP / HSA API function the script compares its calls inside slow frames
(span > 1.4 x median) with those inside normal frames: calls per frame and total microseconds per frame.  A function that only appears -- or whose
time explodes -- in the slow frames names the tail."""
import csv, glob, json, os, sys
from collections import defaultdict

import numpy as np

src = sys.argv[1]


def read(pattern):
    out = []
    for path in glob.glob(os.path.join(src, "**", pattern), recursive=True):
        for r in csv.DictReader(open(path)):
            out.append(r)
    return out


kern = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in read("*kernel_trace.csv"))
cuts = [s for s, e, n in kern if "frame_build_kernel" in n]
spans = np.diff(cuts) / 1e3
first = 40
med = float(np.median(spans[first:]))
slow_idx = [i for i in range(first, len(spans)) if spans[i] > 1.4 * med]
slow_set = set(slow_idx)
api = []
for pat, dom in (("*hip_api_trace.csv", "hip"), ("*hsa_api_trace.csv", "hsa")):
    for r in read(pat):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), dom + ":" + r["Function"]))
api.sort()
cuts_arr = np.array(cuts)
acc = defaultdict(lambda: {"slow_calls": 0, "slow_us": 0.0, "normal_calls": 0, "normal_us": 0.0, "longest_us": 0.0})
for s, e, fn in api:
    f = int(np.searchsorted(cuts_arr, s, side="right")) - 1   # the frame whose build kernel started last before this call ... the call belongs to the NEXT build if it precedes it
    if f < first or f >= len(spans):
        continue
    a = acc[fn]
    k = "slow" if f in slow_set else "normal"
    a[k + "_calls"] += 1
    a[k + "_us"] += (e - s) / 1e3
    a["longest_us"] = max(a["longest_us"], (e - s) / 1e3)
n_slow, n_norm = max(1, len(slow_idx)), max(1, len(spans) - first - len(slow_idx))
rows = []
for fn, a in acc.items():
    rows.append({"function": fn, "us_per_slow_frame": a["slow_us"] / n_slow, "us_per_normal_frame": a["normal_us"] / n_norm, "calls_per_slow_frame": a["slow_calls"] / n_slow,
                 "calls_per_normal_frame": a["normal_calls"] / n_norm, "longest_us": a["longest_us"]})
rows.sort(key=lambda r: -(r["us_per_slow_frame"] - r["us_per_normal_frame"]))
out = {"frames": int(len(spans) - first), "median_span_us": med, "slow_frames": [int(i - first) for i in slow_idx], "slow_span_us": [float(spans[i]) for i in slow_idx],
       "functions_by_excess_time_in_slow_frames": rows[:25], "what": __doc__.split("\n\n")[0].split("\n", 3)[-1] if False else "per HIP / HSA API function: microseconds and calls per slow frame vs per normal frame (frames cut at the frame_build_kernel launches; a call is assigned to the frame whose build kernel started last before it)"}
js = json.dumps(out, indent=1)
print(js[:7000])
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(js)
