#!/usr/bin/env python3
"""Which side carries the slow frames of the live odometry loop -- the device's kernels or the host between them?  ("name the p99".)

  BENCH_KEEP_FRAME_LOOP=<dir> python bench.py --workload odometry_frame          # leaves <dir>/scene.bin and <dir>/odometry_frame_loop
  rocprofv3 --kernel-trace --output-format csv -d <prof> -- <dir>/odometry_frame_loop <dir>/scene.bin 300 3 1
  python tools/frame_tail_trace.py <prof> [out.json]

The trace is cut into frames at the one `frame_build_kernel` launch every glim_amd_frame_create makes.  Per frame: the span from its first kernel's start
to the next frame's first kernel's start, the sum of its kernels' durations, and the longest gap between two consecutive kernels.  A frame is SLOW
when its span exceeds 1.4 x the median span.  If the slow frames' kernel-time sum grows with the span, the kernels themselves ran slower (clock,
contention); if it stays at the median while the span grows, the time went into the host / the launch path between kernels."""
import csv, glob, json, os, sys
from collections import defaultdict

import numpy as np

src = sys.argv[1]
rows = []
for path in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "frame_build_kernel" in r[2]]
frames = []
for a, b in zip(starts[:-1], starts[1:]):
    ks = rows[a:b]
    span = (rows[b][0] - ks[0][0]) / 1e3
    busy = sum(e - s for s, e, _ in ks) / 1e3
    gaps = [(ks[i + 1][0] - ks[i][1]) / 1e3 for i in range(len(ks) - 1)] + [(rows[b][0] - ks[-1][1]) / 1e3]
    frames.append({"span_us": span, "kernel_us": busy, "kernels": len(ks), "longest_gap_us": max(gaps), "t_ms": (ks[0][0] - rows[starts[0]][0]) / 1e6})
frames = frames[40:]  # seeding (17 frames) + warm-up (20)
span = np.array([f["span_us"] for f in frames]); busy = np.array([f["kernel_us"] for f in frames])
med_span, med_busy = float(np.median(span)), float(np.median(busy))
slow = span > 1.4 * med_span
per_kernel = defaultdict(lambda: {"normal": [], "slow": []})
for idx, (a, b) in enumerate(zip(starts[:-1], starts[1:])):
    if idx < 40:
        continue
    for s, e, n in rows[a:b]:
        per_kernel[n.split("(")[0][-60:]]["slow" if slow[idx - 40] else "normal"].append((e - s) / 1e3)
out = {
    "frames": int(len(frames)), "median_span_us": med_span, "median_kernel_time_us": med_busy, "slow_frames": int(slow.sum()),
    "slow_frames_span_us_mean": float(span[slow].mean()) if slow.any() else None, "slow_frames_kernel_time_us_mean": float(busy[slow].mean()) if slow.any() else None,
    "normal_frames_span_us_mean": float(span[~slow].mean()), "normal_frames_kernel_time_us_mean": float(busy[~slow].mean()),
    "slow_frame_list": [{"frame": int(i), **{k: round(v, 1) for k, v in frames[i].items()}} for i in np.nonzero(slow)[0][:40]],
    "kernel_duration_us_normal_vs_slow_frames": {k: {"normal_median": float(np.median(v["normal"])) if v["normal"] else None, "slow_median": float(np.median(v["slow"])) if v["slow"] else None,
                                                     "calls_per_frame": round((len(v["normal"]) + len(v["slow"])) / len(frames), 2)} for k, v in per_kernel.items()},
    "what": __doc__.split("\n\n")[2],
}
js = json.dumps(out, indent=1)
print(js[:6000])
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(js)
