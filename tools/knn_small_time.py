"""find_neighbors(k = 10) on preprocessed-frame-sized clouds (the shipped config_preprocess.json keeps 10 000 points): grid path vs Hilbert-chunk path.
usage (GPU box): python tools/knn_small_time.py"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from glim_amd import api, synth
from oracle import oracle as orc

ctx = api.Context(0, 1)
scene = synth.Scene.default()
full = synth.scan(scene, synth.arc_trajectory(1)[0], synth.lidar_directions(128, 1024), 0)
rng = np.random.default_rng(0)
for n in (10000, 12000, 16384, 24576, 32768):
    pts = full[np.sort(rng.choice(len(full), n, replace=False))]
    g = api.PointCloudGPU.clone(pts, ctx=ctx)
    ref = orc.knn(pts, 10)
    row = []
    for path in ("grid", "chunks"):
        ctx.set_diag(f"knn_path={path}")
        g.find_neighbors(10, download=False)
        t = time.perf_counter()
        for _ in range(20):
            g.find_neighbors(10, download=False)
        dt = (time.perf_counter() - t) / 20
        ok = bool((g.find_neighbors(10) == ref).all())
        row.append(f"{path} {dt * 1e3:.3f} ms exact {ok}")
    ctx.set_diag("")
    print(n, " | ".join(row), flush=True)
