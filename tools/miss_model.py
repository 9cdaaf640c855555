"""CPU model: how many wavefront trips of the fused factor kernel have NO correspondence at all -- 64 consecutive points of the Hilbert-ordered
source stream that all miss the target voxel map -- for pairs of the 256-submap all-pairs cost (bench.py --workload global256: merged submaps of four
40 x 512 scans on a 2 m grid, 1.0 m voxels).  Such a trip adds exact zeros; -DGLIM_AMD_K4_SKIP_ALLMISS=1 lets the general kernel skip its record gather
and algebra.  A design tool, no GPU needed."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from glim_amd import synth
from glim_amd.se3 import se3_exp
from knn_model import hilbert_keys
scene = synth.Scene.default()
dirs = synth.lidar_directions(40, 512)
S = 256; side = 16
origins = synth.grid_trajectory(side, side, spacing=2.0)[:S]
rng = np.random.default_rng(77)
def submap(s_idx):
    T_origin = origins[s_idx]
    pts = []
    r = np.random.default_rng(1000 + s_idx)
    for k in range(4):
        T_frame = T_origin @ se3_exp(np.r_[r.normal(size=3) * 0.02, (0.35 * k, 0.05 * r.normal(), 0.0)])
        scan = np.asarray(synth.scan(scene, T_frame, dirs, frame_id=10 * s_idx + k))[:, :3].astype(np.float64)
        rel = np.linalg.inv(T_origin) @ T_frame
        pts.append(scan @ rel[:3, :3].T + rel[:3, 3])
    p = np.vstack(pts)
    # 0.1 m voxel-grid merge (mean per voxel), like merge_frames
    key = np.floor(p / 0.1).astype(np.int64)
    _, inv = np.unique(key, axis=0, return_inverse=True)
    cnt = np.bincount(inv); out = np.stack([np.bincount(inv, weights=p[:, a]) / cnt for a in range(3)], axis=1)
    return T_origin, out.astype(np.float32)
pairs = [(0, 1), (0, 17), (5, 40), (100, 101), (100, 116), (30, 200), (0, 255), (120, 135), (64, 192), (10, 250), (77, 78), (200, 216)]
if len(sys.argv) > 1 and sys.argv[1] == "--random":  # a uniform sample of the 32 640 pairs instead of the hand-picked near / far ones
    pr = np.random.default_rng(2024)
    pairs = []
    while len(pairs) < int(sys.argv[2]):
        i, j = sorted(pr.integers(0, S, size=2))
        if i != j:
            pairs.append((int(i), int(j)))
cache = {}
tot_trips = tot_allmiss = 0; tot_pts = tot_hits = 0
for (i, j) in pairs:
    for s in (i, j):
        if s not in cache: cache[s] = submap(s)
    Ti, Pi = cache[i]; Tj, Pj = cache[j]
    D = np.linalg.inv(Ti) @ Tj
    vox = set(map(tuple, np.floor(Pi.astype(np.float64) / 1.0).astype(np.int64)))
    order = np.argsort(hilbert_keys(Pj), kind="stable")
    q = Pj[order].astype(np.float64) @ D[:3, :3].T + D[:3, 3]
    c = np.floor(q / 1.0).astype(np.int64)
    hit = np.fromiter((tuple(x) in vox for x in c), bool, len(c))
    n = len(hit) // 64 * 64
    h = hit[:n].reshape(-1, 64).sum(axis=1)
    print(f"pair ({i},{j}): {len(Pj)} pts, inlier fraction {hit.mean():.2f}, wave trips with no hit {np.mean(h == 0):.3f}, with <= 8 hits {np.mean(h <= 8):.3f}, all 64 hit {np.mean(h == 64):.3f}")
    tot_trips += len(h); tot_allmiss += (h == 0).sum(); tot_pts += len(hit); tot_hits += hit.sum()
print("overall: inlier fraction %.3f, all-miss trips %.3f" % (tot_hits / tot_pts, tot_allmiss / tot_trips))
