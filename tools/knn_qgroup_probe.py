"""A/B of the query-group kNN kernel (knn_qgroup.hip, diag knn_kernel=qgroup) against the shipped chunk kernels: lists against the oracle on the
degenerate clouds of tests/test_gpu_parity.py and on scans, wall time per call at the sizes DESIGN.md quotes."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from glim_amd import api, synth
from oracle import oracle as orc

ctx = api.Context(0, 1)
rng = np.random.default_rng(5)
scene = synth.Scene.default()
scan128 = synth.scan(scene, synth.arc_trajectory(1)[0], synth.lidar_directions(128, 1024), 0)[:, :3]
scan64 = synth.scan(scene, synth.arc_trajectory(1)[0], synth.lidar_directions(64, 1024), 0)[:, :3]
dense = rng.normal(size=(20000, 3)) * [0.05, 0.05, 0.02]
sparse = rng.uniform(-30, 30, size=(9000, 3)) * [1, 1, 0.1]
line = np.c_[np.linspace(0, 40, 2937), np.zeros(2937), np.zeros(2937)]
clouds = {
    "lattice": np.stack(np.meshgrid(np.arange(20), np.arange(20), np.arange(15), indexing="ij"), -1).reshape(-1, 3) * 0.25,
    "identical": np.tile([[1.0, 2.0, 3.0]], (3000, 1)),
    "offset": rng.uniform(-1, 1, (6000, 3)) + [1e5, -2e5, 3e4],
    "two_scales": np.vstack([rng.normal(size=(4000, 3)) * 0.01, rng.uniform(-50, 50, (3000, 3))]),
    "duplicates": np.repeat(rng.uniform(-1, 1, (500, 3)), 9, axis=0),
    "mixed32437": np.vstack([dense, sparse, line, dense[:500]]),
    "scan10000": scan128[np.sort(rng.choice(len(scan128), 10000, replace=False))],
    "scan32768": scan128[np.sort(rng.choice(len(scan128), 32768, replace=False))],
    "scan65536": scan64,
    "scan131072": scan128,
}
ok_all = True
for name, pts in clouds.items():
    pts = np.asarray(pts).astype(np.float32)
    g = api.PointCloudGPU.clone(pts, ctx=ctx)
    small = len(pts) <= 40000
    for k in ((10, 5, 16, 32, 1) if small else (10,)):
        ref = orc.knn(pts.astype(np.float64), k, method="brute") if small else None
        ctx.set_diag("knn_path=chunks,knn_kernel=wave64")
        base = g.find_neighbors(k)
        ctx.set_diag("knn_path=chunks,knn_kernel=qgroup")
        got = g.find_neighbors(k)
        same = bool((got == base).all())
        exact = bool((got == ref).all()) if ref is not None else None
        ok_all &= same and (exact is not False)
        if not same:
            bad = np.nonzero((got != base).any(axis=1))[0]
            print("  MISMATCH", name, k, len(bad), "rows; first", bad[:3], got[bad[0]], base[bad[0]], flush=True)
        print(f"{name:12s} n={len(pts):6d} k={k:2d} equals_shipped={same} equals_oracle={exact}", flush=True)
    ctx.set_diag("")
room = synth.Scene.small_room()
rgbd = synth.scan(room, synth.pose(-2.5, -1.5, 1.4, 0.5), synth.pinhole_directions(640, 480, 70, 55), 0, sigma=0.002, max_range=8.0, min_range=0.3)[:, :3]
for name, pts in (("scan10000", clouds["scan10000"]), ("scan32768", clouds["scan32768"]), ("scan65536", scan64), ("scan131072", scan128), ("rgbd307104", rgbd)):
    pts = np.asarray(pts).astype(np.float32)
    g = api.PointCloudGPU.clone(pts, ctx=ctx)
    row = {}
    variants = ("qgroup", "wave64") + (("pair",) if len(pts) <= 131072 else ())
    for variant in variants:
        ctx.set_diag("")
        ctx.set_diag(f"knn_kernel={variant}")
        g.find_neighbors(10, download=False)
        ts = []
        for _ in range(7):
            t = time.perf_counter(); g.find_neighbors(10, download=False); ts.append(time.perf_counter() - t)
        row[variant] = min(ts) * 1e3
        row[variant + "_lists"] = g.find_neighbors(10)
    same = all(bool((row["qgroup_lists"] == row[v + "_lists"]).all()) for v in variants)
    ok_all &= same
    print(f"time {name:12s} n={len(pts):6d}  " + "   ".join(f"{v} {row[v]:.3f} ms" for v in variants) + f"   identical={same}", flush=True)
    ctx.set_diag("")
    ctx.set_diag("knn_kernel=qgroup,knn_debug=/tmp/knn_dbg.bin")
    g.find_neighbors(10, download=False)
    c = np.fromfile("/tmp/knn_dbg.bin", dtype=np.int32)[:5].astype(np.float64)
    print(f"     query-group kernel: {int(c[0])} wavefronts of {len(pts) / c[0]:.2f} queries; per wavefront: chunk scans {c[1] / c[0]:.1f}, "
          f"exact query-chunk evaluations {c[2] / c[0]:.1f}, insertions {c[3] / c[0]:.1f}, chunk-test rounds {c[4] / c[0]:.1f}", flush=True)
ctx.set_diag("")
print("ALL_OK", ok_all)
