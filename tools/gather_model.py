"""CPU model of K4's gather traffic: distinct 128-byte table lines a wavefront trip touches, for alternative voxel -> line groupings.

The fused factor kernel reads one table line per distinct voxel GROUP among the 64 points of a wavefront trip (today a group is an x-adjacent
voxel pair: two 8-byte keys + two 48-byte records in one 128-byte bucket).  Points are streamed in the Hilbert order of the source cloud and
dealt to blocks in 256-point hands, so a trip is 64 consecutive points of that order.  This script counts, for the bench workload
(131 072-pt scans, 0.5 m voxels, consecutive scans 0.5 m / 2 deg apart), the distinct groups per trip under other groupings that would fit
a line with denser records -- the number the measured 1.43 M line fetches per 128-factor launch corresponds to.  A design tool, no GPU needed.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    from glim_amd import synth
    from knn_model import hilbert_keys

    scene = synth.Scene.default()
    dirs = synth.lidar_directions(128, 1024)
    poses = synth.arc_trajectory(3)
    res = 0.5
    rows = []
    for f in range(2):
        src = np.asarray(synth.scan(scene, poses[f + 1], dirs, f + 1))[:, :3].astype(np.float32)
        D = synth.relative_pose(poses[f], poses[f + 1])
        order = np.argsort(hilbert_keys(src), kind="stable")
        q = src[order].astype(np.float64) @ D[:3, :3].T + D[:3, 3]
        c = np.floor(q / res).astype(np.int64)
        n = len(c) // 64 * 64
        c = c[:n].reshape(-1, 64, 3)

        def distinct(gx, gy, gz, shift=(0, 0, 0)):
            g = np.stack([(c[..., 0] + shift[0]) // gx, (c[..., 1] + shift[1]) // gy, (c[..., 2] + shift[2]) // gz], axis=-1)
            key = (g[..., 0] * 1_000_003 + g[..., 1]) * 1_000_003 + g[..., 2]
            key.sort(axis=1)
            return 1 + (np.diff(key, axis=1) != 0).sum(axis=1)

        for name, shape in (("single voxel", (1, 1, 1)), ("x pair (today)", (2, 1, 1)), ("x triple", (3, 1, 1)), ("x quad", (4, 1, 1)),
                            ("2 x 2 x 1", (2, 2, 1)), ("2 x 2 x 2", (2, 2, 2)), ("3 x 2 x 1", (3, 2, 1)), ("4 x 2 x 1", (4, 2, 1)), ("4 x 4 x 1", (4, 4, 1))):
            d = distinct(*shape)
            rows.append((f, name, shape[0] * shape[1] * shape[2], d.mean(), np.percentile(d, 90), d.max()))
    print("distinct table lines per wavefront trip (64 consecutive points of the Hilbert-ordered source stream), 131 072-pt scans, 0.5 m voxels")
    print("%-16s %6s %8s %6s %5s   lines per 128-factor launch" % ("group", "voxels", "mean", "p90", "max"))
    for name in dict.fromkeys(r[1] for r in rows):
        rr = [r for r in rows if r[1] == name]
        mean = np.mean([r[3] for r in rr])
        print("%-16s %6d %8.2f %6.1f %5d   %.2f M" % (name, rr[0][2], mean, np.mean([r[4] for r in rr]), max(r[5] for r in rr), mean * 131072 / 64 * 128 / 1e6))


if __name__ == "__main__":
    main()
