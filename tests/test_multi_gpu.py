"""Single-process multi-device cost evaluation (glim_amd_multi_*) on the GPU box: a plain C program through the C ABI, and the Python mirror.
With one visible device there is nothing to gather: the library is exercised when the handle is created (a one-rank in-place all-gather that must
come back unchanged: uses_rccl must be true) and, on request, in every evaluation (set_one_rank_collective) -- with identical records."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_multi.c")


def _build(tmp_path):
    from glim_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    exe = str(tmp_path / "test_multi")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", SRC, "-o", exe, "-L" + os.path.join(ROOT, "glim_amd"), "-lglim_amd", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "glim_amd"), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    return exe


def test_multi_c_program_compiles_as_c99(tmp_path):
    _build(tmp_path)


@pytest.mark.gpu
def test_multi_device_c_program(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "test_multi OK" in out.stdout


@pytest.mark.gpu
def test_multi_device_python_mirror_matches_oracle(orc):
    from glim_amd import api, synth

    scene = synth.Scene.default()
    dirs = synth.lidar_directions(32, 256)
    poses = synth.arc_trajectory(4)
    scans = [synth.scan(scene, T, dirs, i) for i, T in enumerate(poses)]
    md = api.MultiDeviceCost(list(range(api.device_count())))
    info = md.info()
    assert info["uses_rccl"] and info["num_devices"] == api.device_count()
    cids, mids = [], []
    for s in scans:
        c = md.add_cloud(s)
        md.estimate_covariances(c, 10)
        cids.append(c)
        mids.append(md.add_voxelmap(c, 1.0))
    pairs = [(i, j) for i in range(4) for j in range(4) if i != j]
    md.set_factors([mids[i] for i, _ in pairs], [cids[j] for _, j in pairs], [api.FACTOR_BINARY] * len(pairs))
    b = md.shard()
    assert b[0] == 0 and b[-1] == len(pairs)
    deltas = np.stack([api.pose12(synth.relative_pose(poses[i], poses[j])) for i, j in pairs])
    out, total = md.linearize(deltas)
    assert total == pytest.approx(sum(o["error"] for o in out), rel=1e-12)
    if info["num_devices"] == 1:  # the no-op library call inside the evaluation changes nothing but the host's account of it
        rec0 = md.records().copy()
        assert md.last_breakdown(0)["library_calls"] == 0.0
        md.set_one_rank_collective(True)
        assert md.evaluate(deltas) == pytest.approx(total, rel=1e-15)
        assert md.last_breakdown(0)["library_calls"] > 0.0
        np.testing.assert_array_equal(md.records(), rec0)
        md.set_one_rank_collective(False)
    bd = md.last_breakdown(0)
    assert bd["total"] > 0 and bd["device_gather"] >= 0 and bd["device_copy_out"] >= 0
    # the records reach the host by a second store of the finalising kernels (default) or by copies behind the pieces: the same bytes
    rec_mirrored = md.records().copy()
    md.set_host_records(0)
    md.set_factors([mids[i] for i, _ in pairs], [cids[j] for _, j in pairs], [api.FACTOR_BINARY] * len(pairs))
    assert md.evaluate(deltas) == pytest.approx(total, rel=1e-15)
    np.testing.assert_array_equal(md.records(), rec_mirrored)
    md.set_host_records(1)
    md.set_factors([mids[i] for i, _ in pairs], [cids[j] for _, j in pairs], [api.FACTOR_BINARY] * len(pairs))
    out2, total2 = md.linearize(deltas)
    assert total2 == total
    np.testing.assert_array_equal(out2[3]["H_ss"], out[3]["H_ss"])
    # oracle on the same inputs (covariances as the device estimated them)
    ctx = api.Context(0, 1)
    covs = []
    for s in scans:
        g = api.PointCloudGPU.clone(s, ctx=ctx)
        g.find_neighbors(10, download=False)
        g.estimate_covariances(10)
        covs.append(g.download(normals=False)[1].astype(np.float64))
    maps = [orc.VoxelMap(1.0).insert(s, c) for s, c in zip(scans, covs)]
    for (i, j), got, d in zip(pairs, out, deltas):
        D = np.eye(4)
        D[:3, :4] = d.reshape(3, 4)
        ref = orc.vgicp_linearize(maps[i], scans[j], covs[j], D)
        assert got["num_inliers"] == ref["num_inliers"]
        step = np.abs(np.linalg.solve(got["H_ss"], -got["b_s"]) - np.linalg.solve(ref["H_ss"], -ref["b_s"])).max()
        assert step < 1e-4
        np.testing.assert_allclose(got["H_tt"], ref["H_tt"], rtol=0, atol=2e-4 * np.abs(ref["H_tt"]).max())
    md.close()


# ---- the N > 1 path on ONE GPU: "virtual devices"  ---------------------------------------------------------------------
# multi.hip with ndev >= 2 -- worker threads, the host barrier, per-device shards and pieces, the exchange behind every piece, the failure paths --
# had never executed anywhere: the GPU box has one device and glim_amd_multi_create refuses one ordinal twice.  A handle made by
# glim_amd_debug_multi_create_virtual (or by glim_amd_multi_create under GLIM_AMD_DIAG=multi_virtual=1) accepts it: every entry gets its own
# context, thread, shard and gathered array, and the in-place ncclAllGather of a piece is replaced by same-device copies of equal slots on the
# collective stream (RCCL refuses duplicate devices; distinct devices keep ncclCommInitAll / ncclAllGather).

PPT = 4  # the same points-per-thread rule on both sides -> the same per-factor block partition -> the same FP32 summation order -> the same bits


@pytest.fixture(scope="module")
def many_small_scans():
    from glim_amd import synth

    scene = synth.Scene.default()
    poses = synth.arc_trajectory(24)
    scans = []
    for i, T in enumerate(poses):
        dirs = synth.lidar_directions(16, 96 + 8 * (i % 5))  # 1 536 .. 2 048 points: shards are cost-balanced, not count-balanced
        scans.append(synth.scan(scene, T, dirs, i))
    return poses, scans


def _unsharded(api, scans, poses, pairs, flags, deltas):
    """The reference of the sharded evaluation: ONE NonlinearFactorSetGPU over the whole list on one context, same ppt."""
    ctx = api.Context(0, 1)
    ctx.set_diag(f"ppt={PPT}")
    clouds, maps = [], []
    for s in scans:
        g = api.PointCloudGPU.clone(s, ctx=ctx)
        g.find_neighbors(10, download=False)
        g.estimate_covariances(10)
        clouds.append(g)
        maps.append(api.GaussianVoxelMapGPU(1.0, ctx=ctx).insert(g))
    fset = api.NonlinearFactorSetGPU(ctx)
    for (i, j), fl in zip(pairs, flags):
        f = api.IntegratedVGICPFactorGPU(i, j, maps[i], clouds[j]) if fl & api.FACTOR_BINARY else api.IntegratedVGICPFactorGPU(np.eye(4), j, maps[i], clouds[j])
        fset.add(f)
    want = fset.linearize_poses(deltas)
    fset.close()
    return want, (ctx, clouds, maps)


def _virtual_cost(api, ndev, scans, pairs, flags, split):
    md = api.MultiDeviceCost([0] * ndev, virtual=True)
    info = md.info()
    assert info["num_devices"] == ndev and (not info["uses_rccl"] or ndev == 1)
    md.set_diag(f"ppt={PPT}")
    md.set_split(split)
    cids, mids = [], []
    for s in scans:
        c = md.add_cloud(s)
        md.estimate_covariances(c, 10)
        cids.append(c)
        mids.append(md.add_voxelmap(c, 1.0))
    md.set_factors([mids[i] for i, _ in pairs], [cids[j] for _, j in pairs], flags)
    return md


def _assert_same_bits(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g["num_inliers"] == w["num_inliers"]
        assert g["error"] == w["error"]
        for k in ("H_tt", "H_ss", "H_ts", "b_t", "b_s"):
            np.testing.assert_array_equal(g[k], w[k])


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("ndev,split", [(2, 1), (2, 2), (2, 4), (3, -1), (3, 2), (8, -1)])
def test_virtual_devices_give_the_unsharded_sets_bits(many_small_scans, ndev, split):
    """552 factors (every ordered pair of 24 scans, binary and unary mixed) over 2 / 3 / 8 virtual devices, 1 / 2 / 4 pieces per shard: every
    record -- through the host array AND in every device's gathered array after the exchange -- is bit-equal to the unsharded
    NonlinearFactorSetGPU with the same per-factor block partition, in both gather modes; total error equal."""
    from glim_amd import api, synth

    poses, scans = many_small_scans
    pairs = [(i, j) for i in range(len(scans)) for j in range(len(scans)) if i != j]
    flags = [api.FACTOR_BINARY if (k % 3) else 0 for k in range(len(pairs))]
    deltas = np.stack([api.pose12(synth.relative_pose(poses[i], poses[j])) for i, j in pairs])
    want, keep = _unsharded(api, scans, poses, pairs, flags, deltas)
    assert min(w["num_inliers"] for w in want) >= 0 and max(w["num_inliers"] for w in want) > 500
    md = _virtual_cost(api, ndev, scans, pairs, flags, split)
    b = md.shard()
    assert b[0] == 0 and b[-1] == len(pairs) and np.all(np.diff(b) >= 0)
    rows, mr, pc, pr = api.shard_layout(b, split)
    assert len(set(rows.tolist())) == len(pairs)  # every factor its own row of the gathered array
    if ndev == 2 and split in (2, 4):
        assert pc == split  # (276 factors per shard: pieces of >= 64 rows)
    want_total = float(np.sum([w["error"] for w in want]))  # (host sum in factor order, the order glim_amd_multi_linearize uses with `out`)
    for mode in (1, 2, 1):
        md.set_gather_mode(mode)
        for rep in range(2):  # (the second evaluation's kernels wait for the first one's exchange on the device)
            got, total = md.linearize(deltas)
            _assert_same_bits(got, want)
            assert total == want_total
        rec_host = md.records().copy()
        for d in range(ndev):  # what a device-side consumer on ANY device reads after the exchange
            np.testing.assert_array_equal(md.gathered_records(d), rec_host)
        dev_total = md.evaluate(deltas)  # summed by the devices, shard by shard
        assert dev_total == pytest.approx(want_total, rel=1e-13)
        np.testing.assert_array_equal(md.records(), rec_host)
    md.set_gather_mode(0)  # no exchange: host records only
    got, total = md.linearize(deltas)
    _assert_same_bits(got, want)
    k_ms, g_ms = md.last_timing()
    assert len(k_ms) == ndev and all(k > 0 for k, lo, hi in zip(k_ms, b[:-1], b[1:]) if hi > lo)
    md.close()
    del keep


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_virtual_devices_ragged_and_empty_shards(many_small_scans):
    """Fewer factors than devices (5 over 8: three shards are empty), one factor, and a list whose costs leave a ragged last shard."""
    from glim_amd import api, synth

    poses, scans = many_small_scans
    scans, poses = scans[:6], poses[:6]
    for pairs in ([(0, 1), (1, 2), (2, 3), (3, 4), (4, 5)], [(2, 0)], [(i, j) for i in range(6) for j in range(6) if i != j][:13]):
        flags = [api.FACTOR_BINARY] * len(pairs)
        deltas = np.stack([api.pose12(synth.relative_pose(poses[i], poses[j])) for i, j in pairs])
        want, keep = _unsharded(api, scans, poses, pairs, flags, deltas)
        for ndev in (3, 8):
            md = _virtual_cost(api, ndev, scans, pairs, flags, -1)
            b = md.shard()
            if len(pairs) < ndev:
                assert np.any(np.diff(b) == 0)  # empty shards
            got, total = md.linearize(deltas)
            _assert_same_bits(got, want)
            for d in range(ndev):
                np.testing.assert_array_equal(md.gathered_records(d), md.records())
            assert md.evaluate(deltas) == pytest.approx(total, rel=1e-13)
            md.close()
        del keep


@pytest.mark.gpu
@pytest.mark.timeout(120)
def test_virtual_devices_failure_injection_returns_on_every_thread(many_small_scans):
    """One "device" fails BEFORE the host barrier: no device starts an exchange, the call returns the error in bounded time (the test's time-out is
    the bound), the handle stays usable and the next evaluation has the right bits.  One fails INSIDE its exchange: the call returns the error, the
    handle is retired (`broken`): later evaluations are refused with GLIM_AMD_ERR_STATE instead of hanging, destroy still works."""
    import time

    from glim_amd import api, synth
    from glim_amd._lib import GlimAmdError

    poses, scans = many_small_scans
    scans, poses = scans[:8], poses[:8]
    pairs = [(i, j) for i in range(8) for j in range(8) if i != j]
    flags = [api.FACTOR_BINARY] * len(pairs)
    deltas = np.stack([api.pose12(synth.relative_pose(poses[i], poses[j])) for i, j in pairs])
    want, keep = _unsharded(api, scans, poses, pairs, flags, deltas)
    for mode in (1, 2):
        md = _virtual_cost(api, 3, scans, pairs, flags, -1)
        md.set_gather_mode(mode)
        _assert_same_bits(md.linearize(deltas)[0], want)
        for victim in (0, 1, 2):  # the caller's own thread (device 0) and the workers
            md.inject_failure(victim, 1)
            t0 = time.perf_counter()
            with pytest.raises(GlimAmdError) as ei:
                md.linearize(deltas)
            assert ei.value.code == -2 and time.perf_counter() - t0 < 5.0
            _assert_same_bits(md.linearize(deltas)[0], want)  # not broken: nothing was exchanged
        md.inject_failure(1, 2)
        t0 = time.perf_counter()
        with pytest.raises(GlimAmdError) as ei:
            md.linearize(deltas)
        assert ei.value.code == -2 and time.perf_counter() - t0 < 5.0
        for _ in range(2):
            with pytest.raises(GlimAmdError) as ei:
                md.linearize(deltas)
            assert ei.value.code == -5  # GLIM_AMD_ERR_STATE: retired
        with pytest.raises(GlimAmdError):
            md.gathered_device(0)
        md.close()
    del keep


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("ndev", [2, 3, 8])
def test_multi_c_program_over_virtual_devices(tmp_path, ndev):
    """The plain-C program of test_multi_device_c_program with device 0 listed N times: glim_amd_multi_create itself accepts that under
    GLIM_AMD_DIAG=multi_virtual=1 (the process-wide switch), and the sharded records agree with the unsharded set."""
    exe = _build(tmp_path)
    env = dict(os.environ, GLIM_AMD_DIAG="multi_virtual=1")
    out = subprocess.run([exe, "virtual", str(ndev)], capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "test_multi OK" in out.stdout and f"devices {ndev}, rccl 0" in out.stdout
    # without the switch one ordinal twice is refused, as before
    bad = subprocess.run([exe, "virtual", "2"], capture_output=True, text=True, timeout=120, env={k: v for k, v in os.environ.items() if k != "GLIM_AMD_DIAG"})
    assert bad.returncode != 0


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_shards_of_thousands_of_short_factors_reach_the_host_array(many_small_scans):
    """More than 2 048 factors of a few rows each in ONE piece -- every piece of configs[3] -- are finalised by the wave-per-factor kernel, which in
    round 5 never made the second store into the host array (the records of glim_amd_multi_records / `out` stayed zero; only the device-summed
    cost was read by the bench, so nobody saw it).  5 520 factors over 1 (one piece of 4 x 1 380... the default split) and 2 virtual devices: host
    records == every device's gathered array == the unsharded set, bit for bit."""
    from glim_amd import api, synth

    poses, scans = many_small_scans
    base = [(i, j) for i in range(len(scans)) for j in range(len(scans)) if i != j]
    pairs = base * 10
    flags = [api.FACTOR_BINARY] * len(pairs)
    deltas = np.stack([api.pose12(synth.relative_pose(poses[i], poses[j]) @ synth.pose(0.001 * (k % 7), 0.0, 0.0)) for k, (i, j) in enumerate(pairs)])
    want, keep = _unsharded(api, scans, poses, pairs, flags, deltas)
    for ndev, split in ((1, 1), (2, 1), (2, -1)):
        md = _virtual_cost(api, ndev, scans, pairs, flags, split)
        got, total = md.linearize(deltas)
        _assert_same_bits(got, want)
        rec = md.records()
        assert np.all(rec[:, 0] == [w["num_inliers"] for w in want]) and np.abs(rec[:, 2:]).max() > 0
        for d in range(ndev):
            np.testing.assert_array_equal(md.gathered_records(d), rec)
        md.close()
    del keep
