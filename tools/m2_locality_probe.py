#!/usr/bin/env python3
"""What bounds the general-form factor kernel on configs[3] -- vector-ALU issue or the memory system?  

The all-pairs cost streams 256 clouds / maps (0.7 GB) through 32 MiB of L2 and the 256 MiB Infinity Cache; the instruction-mix model of DESIGN 4.1
fits its absolute time but has over-predicted every instruction-count saving (FP32 transform, packed FP32, pre-cull).  This probe runs the SAME
kernel, the SAME number of factors and (to within sampling) the same mix of inlier fractions over working sets of different size:

  all_pairs          the 32 640 pairs of 256 submaps, source-major (bench.py's configs[3])
  sample64_x510      64 pairs sampled uniformly, each repeated 510 times back to back: the resident blocks of the device
                     work on <= a few pairs at any time -- every stream and every bucket line is an L2 hit after the first touch
  sample64_shuffled  the same 32 640-factor multiset in random order: 64 sources / targets in flight at once (L2 + Infinity Cache resident)

and reports kernel ms (HIP events, fset.profile), point visits and the kernel's own count of skipped trips for each, so the times can be compared
per point visit.  If sample64_x510 is not faster per visit than all_pairs, the memory system is not what the kernel waits for.
  python tools/m2_locality_probe.py [out.json]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import bench
    from glim_amd import api, synth

    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    ctx = api.Context(0, 1, external_stream=stream.cuda_stream)
    S = int(os.environ.get("PROBE_SUBMAPS", "256"))
    t0 = time.time()
    submaps = bench.make_merged_submaps(api, ctx, S, 4, 40, 560)
    clouds = [g for _, g in submaps]
    poses = [T for T, _ in submaps]
    vmaps = [api.GaussianVoxelMapGPU(1.0, ctx=ctx).insert(c) for c in clouds]
    sizes = np.array([c.size() for c in clouds])
    print(f"[probe] {S} merged submaps, {sizes.mean():.0f} pts on average, {time.time() - t0:.1f}s", file=sys.stderr)
    pairs = [(i, j) for j in range(S) for i in range(j)]
    deltas = np.stack([api.pose12(synth.relative_pose(poses[i], poses[j])) for i, j in pairs])

    def build(idx):
        fs = api.NonlinearFactorSetGPU(ctx)
        for f in idx:
            i, j = pairs[f]
            fs.add(api.IntegratedVGICPFactorGPU(i, j, vmaps[i], clouds[j]))
        return fs

    def measure(name, idx):
        idx = np.asarray(idx)
        fs = build(idx)
        d = deltas[idx]
        out_dev = torch.zeros(len(idx), api._lib.COMPACT_DOUBLES, dtype=torch.float64, device="cuda")
        fs.trip_stats(reset=True)
        fs.linearize_device_async(d, out_dev.data_ptr())
        torch.cuda.synchronize()
        skipped, trips = fs.trip_stats(reset=True)
        inl = out_dev[:, 0].cpu().numpy()
        src_sizes = sizes[[pairs[f][1] for f in idx]].astype(np.float64)
        rounds = [fs.profile(d, iters=3) for _ in range(4)]
        visits = float(src_sizes.sum())
        ms = float(np.mean([r[0] for r in rounds[1:]]))
        out = {"factors": int(len(idx)), "point_visits": visits, "kernel_ms": ms, "kernel_ms_rounds": [round(float(r[0]), 4) for r in rounds],
               "ns_per_1000_visits": ms * 1e6 / visits * 1e3, "skipped_trip_share": skipped / max(1, trips), "trips_per_evaluation": int(trips),
               "inlier_fraction_of_all_visits": float(inl.sum() / visits)}
        print(f"[probe] {name}: {json.dumps(out)}", file=sys.stderr)
        fs.close() if hasattr(fs, "close") else None
        return out

    res = {}
    all_idx = np.arange(len(pairs))
    res["all_pairs"] = measure("all_pairs", all_idx)
    # samples are uniform (the mix of a uniform sample of 64 of 32 640 pairs is close to the population's; the per-visit numbers
    # below are reported with each arm's own skipped-trip share so a difference in mix is visible)
    for seed in (5, 6):
        rng = np.random.default_rng(seed)
        sample = rng.choice(len(pairs), size=64, replace=False)
        reps = len(pairs) // 64
        res[f"sample64_x{reps}_seed{seed}"] = measure(f"sample64_x{reps}_seed{seed}", np.repeat(sample, reps))
        res[f"sample64_shuffled_seed{seed}"] = measure(f"sample64_shuffled_seed{seed}", rng.permutation(np.repeat(sample, reps)))
        res[f"sample64_once_seed{seed}"] = measure(f"sample64_once_seed{seed}", sample)
    # one pair only, 32 640 times: the smallest working set there is (2.4 MB stream + 0.5 MB map)
    one = int(rng.choice(len(pairs)))
    res["one_pair_x32640"] = measure("one_pair_x32640", np.full(len(pairs), one))
    res["what"] = __doc__.split("\n\n")[1]
    js = json.dumps(res, indent=1)
    print(js)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(js)


if __name__ == "__main__":
    main()
