"""kNN (K2) of one 131 072-pt LiDAR scan and one 307k-pt depth frame, 5 repetitions each -- for rocprofv3 --kernel-trace (GPU box)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glim_amd import api, synth
ctx = api.Context(0, 1)
which = sys.argv[1] if len(sys.argv) > 1 else "lidar"
if which == "lidar":
    pts = synth.scan(synth.Scene.default(), synth.arc_trajectory(1)[0], synth.lidar_directions(128, 1024), 0)
else:
    pts = synth.scan(synth.Scene.small_room(), synth.pose(-2.5, -1.5, 1.4, 0.5), synth.pinhole_directions(640, 480, 70, 55), 0, sigma=0.002, max_range=8.0, min_range=0.3)
g = api.PointCloudGPU.clone(pts, ctx=ctx)
for _ in range(5):
    g.find_neighbors(10, download=False)
