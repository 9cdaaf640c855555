"""Concurrency / lifetime stress on the GPU box: GLIM calls this path from several threads (odometry, sub-mapping and global mapping
run in their own threads: src/glim/odometry/async_odometry_estimation.cpp, src/glim/mapping/async_sub_mapping.cpp,
async_global_mapping.cpp), each with its own stream pool, and builds / drops frames, maps and factor sets continuously."""
import threading

import numpy as np
import pytest


@pytest.mark.gpu
def test_threads_with_own_and_shared_contexts_give_identical_results(orc, small_pair):
    from glim_amd import api

    t, s = small_pair["target"], small_pair["source"]
    T = small_pair["delta"]

    def pipeline(ctx, reps, out):
        try:
            res = []
            for _ in range(reps):
                tg = api.PointCloudGPU.clone(t["points"], ctx=ctx)
                sg = api.PointCloudGPU.clone(s["points"], ctx=ctx)
                for g in (tg, sg):
                    g.find_neighbors(10, download=False)
                    g.estimate_covariances(10)
                vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
                fs = api.NonlinearFactorSetGPU(ctx)
                fs.add(api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, sg))
                L = fs.linearize({1: T})[0]
                gf = api.IntegratedGICPFactor(np.eye(4), 1, tg, sg, max_correspondence_distance=0.5)
                G = gf.linearize({1: T})
                res.append((L["num_inliers"], L["error"], L["H_ss"].copy(), G["num_inliers"], G["error"]))
                gf.close()
            out.append(res)
        except Exception as e:  # surfaced in the main thread
            out.append(e)

    # reference: one thread, one context
    ref_out = []
    pipeline(api.Context(0, 2), 2, ref_out)
    ref = ref_out[0][0]
    assert ref_out[0][1][0] == ref[0] and ref_out[0][1][1] == ref[1]  # bit-reproducible run to run

    shared = api.Context(0, 2)
    outs, threads = [], []
    for i in range(4):
        ctx = shared if i < 2 else api.Context(0, 1)  # two threads share a context (serialised by its mutex), two own theirs
        th = threading.Thread(target=pipeline, args=(ctx, 6, outs))
        threads.append(th)
        th.start()
    for th in threads:
        th.join(timeout=120)
        assert not th.is_alive()
    assert len(outs) == 4
    for o in outs:
        assert not isinstance(o, Exception), o
        for r in o:
            assert r[0] == ref[0] and r[1] == ref[1] and r[3] == ref[3] and r[4] == ref[4]
            np.testing.assert_array_equal(r[2], ref[2])


@pytest.mark.gpu
def test_many_short_lived_objects_do_not_leak_device_memory(small_pair):
    from glim_amd import api

    ctx = api.Context(0, 1)
    s = small_pair["source"]

    def one():
        g = api.PointCloudGPU.clone(s["points"], s["covs"], ctx=ctx)
        vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(g)
        fs = api.NonlinearFactorSetGPU(ctx)
        fs.add(api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, g))
        fs.linearize({1: np.eye(4)})
        pre = api.PointCloudGPU.preprocess(s["points"], np.zeros(len(s["points"])), None, api.preprocess_params(downsample_target=0, downsample_rate=0.5), ctx=ctx)
        pre.deskew(np.eye(4)).estimate_covariances(10)
        for o in (fs, vm, g, pre):
            o.close()

    for _ in range(20):
        one()
    free0 = ctx.device_info()["free_bytes"]
    for _ in range(200):
        one()
    free1 = ctx.device_info()["free_bytes"]
    # the pools cache freed blocks, so steady-state use must not grow the footprint (allow one 2 MiB fragment of slack)
    assert free0 - free1 <= 2 << 20, (free0, free1)
