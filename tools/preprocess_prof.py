"""One configuration of the device preprocessing in a loop, for rocprofv3 --kernel-trace --stats (GPU box)."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from glim_amd import api, synth
ctx = api.Context(0, 1)
mode = sys.argv[1] if len(sys.argv) > 1 else "random"
pts = synth.scan(synth.Scene.default(), synth.arc_trajectory(1)[0], synth.lidar_directions(128, 1024), 0).astype(np.float64)
rng = np.random.default_rng(0)
times, inten = np.sort(rng.uniform(0, 0.1, len(pts))), rng.uniform(0, 255, len(pts))
p4 = np.ones((len(pts), 4)); p4[:, :3] = pts
kw = dict() if mode == "random" else dict(use_random_grid_downsampling=0, downsample_resolution=0.25)
prm = api.preprocess_params(**kw)
for _ in range(20):
    api.PointCloudGPU.preprocess(p4, times, inten, prm, ctx=ctx).close()
