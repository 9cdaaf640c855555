"""find_neighbors on one 131 072-pt synthetic scan with the shipped chunk kernel and with the staged per-lane threshold selection
(GLIM_AMD_KNN_SELECT=1); prints one JSON object.  Run by bench.py in a separate process, outside every timed region."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glim_amd import api, synth  # noqa: E402

ctx = api.Context(0, 1)
xyz = np.asarray(synth.scan(synth.Scene.default(), synth.arc_trajectory(1)[0], synth.lidar_directions(128, 1024), 0))[:, :3].astype(np.float32)
g = api.PointCloudGPU.clone(xyz, ctx=ctx)
res = {}
for tag, val in (("default", None), ("staged_select", "1")):
    os.environ.pop("GLIM_AMD_KNN_SELECT", None)
    if val is not None:
        os.environ["GLIM_AMD_KNN_SELECT"] = val
    g.find_neighbors(10, download=False)
    t0 = time.perf_counter()
    for _ in range(10):
        g.find_neighbors(10, download=False)
    res[tag] = ((time.perf_counter() - t0) / 10 * 1e3, g.find_neighbors(10))
print(json.dumps({"points": int(len(xyz)), "k": 10, "ms_default": res["default"][0], "ms_staged_select": res["staged_select"][0],
                  "lists_equal": bool(np.array_equal(res["default"][1], res["staged_select"][1]))}))
