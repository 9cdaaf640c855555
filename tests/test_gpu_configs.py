"""GPU parity tests at the sizes and on the input shapes of BASELINE.json configs[2], [3] and [4] (run with -m gpu on an MI355X).

  configs[2]  SubMappingGPU bundle: 20 keyframes x 65 536 pts, all 190 pairs x voxel levels {0.25, 0.5} m = 380 binary factors in one
              NonlinearFactorSetGPU (sub_mapping.cpp:276-315, config_sub_mapping_gpu.json:48-50)
  configs[3]  GlobalMapping on MERGED submaps: every submap is the gtsam_points::merge_frames output of its keyframes
              (sub_mapping.cpp:480-497) -- averaged, rotated covariances, no normals -> the general 36 B/pt factor kernel -- matched
              against 1.0 m voxel maps of the other submaps (global_mapping.cpp:253-266, 430-484)
  configs[4]  dense depth frames: 307 200-ray frames, 0.1 m voxels, kNN + covariances + voxel map + factor on the device

Gates (north_star): correspondences bit-exact, inlier counts equal, Gauss-Newton step within 1e-4 of the CPU oracle on the same
f32-rounded inputs.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-4


@pytest.fixture(scope="module")
def api():
    from glim_amd import api as _api

    assert _api.device_count() >= 1, "these tests need a GPU and must not fall back"
    return _api


@pytest.fixture(scope="module")
def ctx(api):
    return api.Context(0, 2)


def gn_step(L, lam=0.0):
    return np.linalg.solve(L["H_ss"] + lam * np.eye(6), -L["b_s"])


def check_factor(got, ref, binary=True):
    assert got["num_inliers"] == ref["num_inliers"]
    np.testing.assert_allclose(got["error"], ref["error"], rtol=2e-4)
    scale = np.abs(ref["H_ss"]).max()
    np.testing.assert_allclose(got["H_ss"], ref["H_ss"], rtol=0, atol=2e-4 * scale)
    if binary:
        np.testing.assert_allclose(got["H_tt"], ref["H_tt"], rtol=0, atol=2e-4 * np.abs(ref["H_tt"]).max())
        np.testing.assert_allclose(got["H_ts"], ref["H_ts"], rtol=0, atol=2e-4 * np.abs(ref["H_ts"]).max())
    lam = 1e-6 * np.trace(ref["H_ss"]) / 6
    for l in (0.0, lam):
        assert np.abs(gn_step(got, l) - gn_step(ref, l)).max() < POSE_TOL


def check_correspondences(fset, index, delta, ref):
    corr = fset.correspondences(index, delta)
    np.testing.assert_array_equal(corr[:, :3], ref["corr"][:, :3])
    np.testing.assert_array_equal(corr[:, 3] > 0, ref["corr"][:, 3] >= 0)


def test_config2_submap_bundle_380_factors(api, ctx, orc):
    from glim_amd import synth

    K = 20
    scene = synth.Scene.default()
    dirs = synth.lidar_directions(64, 1024)
    poses = synth.arc_trajectory(K, step=0.5, yaw_step_deg=1.5)
    scans = [synth.scan(scene, T, dirs, i) for i, T in enumerate(poses)]
    assert all(len(s) == 65536 for s in scans)
    clouds, covs = [], []
    for s in scans:
        g = api.PointCloudGPU.clone(s, ctx=ctx)
        g.find_neighbors(10, download=False)
        g.estimate_covariances(10)
        clouds.append(g)
        covs.append(g.download(normals=False)[1].astype(np.float64))
    levels = (0.25, 0.5)
    vmaps = [[api.GaussianVoxelMapGPU(r, ctx=ctx).insert(c) for r in levels] for c in clouds]
    rng = np.random.default_rng(42)
    values = {i: T @ orc.se3_exp(rng.normal(size=6) * [2e-3, 2e-3, 2e-3, 2e-2, 2e-2, 2e-2]) for i, T in enumerate(poses)}  # a bundle that is not converged yet
    fset = api.NonlinearFactorSetGPU(ctx)
    factors = []
    for i in range(K):
        for j in range(i + 1, K):
            for lv in range(len(levels)):
                f = api.IntegratedVGICPFactorGPU(i, j, vmaps[i][lv], clouds[j])
                fset.add(f)
                factors.append((i, j, lv, f))
    assert fset.size() == 380
    out = fset.linearize(values)
    # every factor: structural properties; a sample of 44 (every pair distance, both levels): full oracle parity
    for (i, j, lv, f), got in zip(factors, out):
        assert got["num_inliers"] > 0.3 * 65536
        assert np.linalg.eigvalsh(got["H_ss"]).min() > 0
    ref_maps = {}
    sample = [k for k in range(len(factors)) if k % 9 == 0 or factors[k][1] - factors[k][0] == K - 1]
    assert len(sample) >= 40
    for k in sample:
        i, j, lv, f = factors[k]
        if (i, lv) not in ref_maps:
            ref_maps[(i, lv)] = orc.VoxelMap(levels[lv]).insert(scans[i], covs[i])
            assert vmaps[i][lv].voxelmap_info()["num_voxels"] == ref_maps[(i, lv)].num_voxels()
        delta = f.calc_delta(values)
        ref = orc.vgicp_linearize(ref_maps[(i, lv)], scans[j], covs[j], delta, want_corr=True)
        check_factor(out[k], ref)
        check_correspondences(fset, k, delta, ref)
    # the LM accept / reject evaluation over the whole bundle (sub_mapping.cpp:435-443): error() with the CPU-factor semantics equals the
    # error part of a linearisation at the same values, factor by factor
    errs = fset.error(values)
    np.testing.assert_allclose(errs, [o["error"] for o in out], rtol=1e-6)


def make_merged_submaps(api, ctx, orc, n_submaps, frames_per_submap=4, rings=32, azimuths=512, seed=3):
    """Submaps as GLIM builds them: keyframes with device-estimated covariances, merged by merge_frames into the submap origin frame."""
    from glim_amd import synth

    scene = synth.Scene.default()
    dirs = synth.lidar_directions(rings, azimuths)
    side = int(np.ceil(np.sqrt(n_submaps)))
    origins = synth.grid_trajectory(side, side, spacing=3.0)[:n_submaps]
    rng = np.random.default_rng(seed)
    submaps = []
    for s, T_origin in enumerate(origins):
        pts_list, cov_list, rel = [], [], []
        for k in range(frames_per_submap):
            T_frame = T_origin @ orc.se3_exp(np.r_[rng.normal(size=3) * 0.02, (k * 0.4, rng.normal() * 0.1, 0.0)])
            scan = synth.scan(scene, T_frame, dirs, frame_id=100 * s + k)
            g = api.PointCloudGPU.clone(scan, ctx=ctx)
            g.find_neighbors(10, download=False)
            g.estimate_covariances(10)
            _, c, _ = g.download(normals=False)
            pts_list.append(scan.astype(np.float64))
            cov_list.append(c.astype(np.float64))
            rel.append(np.linalg.inv(T_origin) @ T_frame)
        merged = api.merge_frames(rel, pts_list, cov_list, downsample_resolution=0.1, ctx=ctx)
        submaps.append((T_origin, merged))
    return submaps


def test_config3_global_mapping_on_merged_submaps(api, ctx, orc):
    S = 16
    submaps = make_merged_submaps(api, ctx, orc, S)
    host = []
    for T, g in submaps:
        p, c = g.download_merged()
        # the factor path consumes the FP32 image of the merged cloud; the oracle gets exactly those values
        xyz, c32, _ = g.download(normals=False)
        np.testing.assert_array_equal(xyz, p.astype(np.float32))
        host.append((xyz, c32.astype(np.float64)))
        assert len(xyz) > 30000
        # averaged, rotated covariances: NOT the plane form (so this exercises the general kernel)
        ev = np.linalg.eigvalsh(c32[:: max(1, len(c32) // 200)].astype(np.float64))
        assert np.median(ev[:, 0]) > 1e-3 * 0.99
    vmaps = [api.GaussianVoxelMapGPU(1.0, ctx=ctx).insert(g) for _, g in submaps]  # global_mapping.cpp:59
    ref_maps = [orc.VoxelMap(1.0).insert(xyz, c) for xyz, c in host]
    rng = np.random.default_rng(9)
    values = {i: T @ orc.se3_exp(rng.normal(size=6) * [1e-3, 1e-3, 1e-3, 1e-2, 1e-2, 1e-2]) for i, (T, _) in enumerate(submaps)}
    fset = api.NonlinearFactorSetGPU(ctx)
    pairs = [(i, j) for i in range(S) for j in range(i + 1, S)]
    factors = []
    for i, j in pairs:
        f = api.IntegratedVGICPFactorGPU(i, j, vmaps[i], submaps[j][1])
        fset.add(f)
        factors.append(f)
    out = fset.linearize(values)
    n_checked = 0
    for k, ((i, j), f) in enumerate(zip(pairs, factors)):
        delta = f.calc_delta(values)
        ov = api.overlap_gpu(vmaps[i], submaps[j][1], delta)
        assert ov == orc.overlap(ref_maps[i], host[j][0], delta)  # global_mapping.cpp:448 min_implicit_loop_overlap test
        ref = orc.vgicp_linearize(ref_maps[i], host[j][0], host[j][1], delta, want_corr=(k % 7 == 0))
        assert out[k]["num_inliers"] == ref["num_inliers"]
        if ref["num_inliers"] > 2000:
            check_factor(out[k], ref)
            n_checked += 1
        if k % 7 == 0:
            check_correspondences(fset, k, delta, ref)
    assert n_checked >= 60
    # both error semantics on general-form clouds
    errs = fset.error(values)
    np.testing.assert_allclose(errs, [o["error"] for o in out], rtol=1e-6)
    moved = {i: T @ orc.se3_exp(rng.normal(size=6) * 2e-3) for i, T in values.items()}
    frozen = fset.error(moved, values_lin=values)
    for k in (0, 17, 63, len(pairs) - 1):
        i, j = pairs[k]
        f = factors[k]
        e_ref, n_ref = orc.vgicp_error(ref_maps[i], host[j][0], host[j][1], f.calc_delta(moved), delta_lin=f.calc_delta(values))
        assert fset.last_error_inliers[k] == n_ref
        assert frozen[k] == pytest.approx(e_ref, rel=3e-4)


def test_mixed_plane_and_general_sources_in_one_set(api, ctx, orc):
    """One factor set holding plane-form sources (device-estimated covariances) and general ones (a merged submap, an uploaded cloud):
    every factor is dispatched to its own kernel variant and matches the oracle; the order of the records follows the order of add()."""
    from glim_amd import synth

    scene = synth.Scene.default()
    dirs = synth.lidar_directions(32, 512)
    poses = synth.arc_trajectory(3)
    scans = [synth.scan(scene, T, dirs, i) for i, T in enumerate(poses)]
    dev = []
    for s in scans:
        g = api.PointCloudGPU.clone(s, ctx=ctx)
        g.find_neighbors(10, download=False)
        g.estimate_covariances(10)
        dev.append(g)
    covs = [g.download(normals=False)[1].astype(np.float64) for g in dev]
    (T_sub, merged), = make_merged_submaps(api, ctx, orc, 1, frames_per_submap=3, rings=32, azimuths=256)
    m_xyz, m_cov, _ = merged.download(normals=False)
    # a cloud uploaded with averaged (non-plane) covariances
    blur = 0.5 * (covs[2] + np.roll(covs[2], 1, axis=0))
    up = api.PointCloudGPU.clone(scans[2].astype(np.float64), blur, ctx=ctx)
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(dev[0])
    ref_vm = orc.VoxelMap(0.5).insert(scans[0], covs[0])
    sources = [(dev[1], scans[1], covs[1], synth.relative_pose(poses[0], poses[1])),
               (merged, m_xyz, m_cov.astype(np.float64), synth.relative_pose(poses[0], T_sub)),
               (dev[2], scans[2], covs[2], synth.relative_pose(poses[0], poses[2])),
               (up, scans[2], blur.astype(np.float32).astype(np.float64), synth.relative_pose(poses[0], poses[2]))]
    fset = api.NonlinearFactorSetGPU(ctx)
    values = {0: np.eye(4)}
    order = [0, 1, 2, 3, 1, 0, 3, 2, 1, 0, 0, 1, 2, 3, 3, 2, 1, 0, 2, 3]  # >= 16 factors: also the XCD-grouped block map of both segments
    for n, k in enumerate(order):
        fset.add(api.IntegratedVGICPFactorGPU(0, n + 1, vm, sources[k][0]))
        values[n + 1] = sources[k][3] @ orc.se3_exp(np.full(6, 1e-3 * (n % 3)))
    out = fset.linearize(values)
    for n, k in enumerate(order):
        _, xyz, cov, _ = sources[k]
        ref = orc.vgicp_linearize(ref_vm, xyz, cov, values[n + 1])
        check_factor(out[n], ref)


def test_uploaded_plane_covariances_take_the_plane_kernel(api, ctx, orc, monkeypatch):
    """GLIM's stock flow keeps CPU covariance estimation and uploads the frame with PointCloudGPU::clone(frame): covariances of the form
    I - 0.999 n n^T with their normals are recognised at upload (24 B/pt kernel); results equal the general kernel's and the oracle's."""
    from glim_amd import synth

    scene = synth.Scene.default()
    dirs = synth.lidar_directions(64, 512)
    poses = synth.arc_trajectory(2)
    tgt, src = synth.scan(scene, poses[0], dirs, 0), synth.scan(scene, poses[1], dirs, 1)
    nt, ct = orc.covariances(tgt, orc.knn(tgt, 10))
    ns, cs = orc.covariances(src, orc.knn(src, 10))
    delta = synth.relative_pose(poses[0], poses[1])
    tg = api.PointCloudGPU.clone(tgt.astype(np.float64), ct, nt, ctx=ctx)   # the reference's Vector4d / Matrix4d upload
    sg = api.PointCloudGPU.clone(src.astype(np.float64), cs, ns, ctx=ctx)
    sg32 = api.PointCloudGPU.clone(src, cs.astype(np.float32), ns.astype(np.float32), ctx=ctx)
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
    c32 = lambda c: c.astype(np.float32).astype(np.float64)  # noqa: E731
    ref = orc.vgicp_linearize(orc.VoxelMap(0.5).insert(tgt, c32(ct)), src, c32(cs), delta)
    base = sg.memory_usage_gpu()
    res = {}
    for name, cloud in (("f64", sg), ("f32", sg32)):
        fset = api.NonlinearFactorSetGPU(ctx)
        fset.add(api.IntegratedVGICPFactorGPU(0, 1, vm, cloud))
        res[name] = fset.linearize({0: np.eye(4), 1: delta})[0]
        check_factor(res[name], ref)
    # the plane-form stream (24 B per point) serves it, not the general one (36 B per point): written by the upload kernel itself for clouds of
    # up to 32 768 points (nothing is added on first use), by the first factor otherwise
    assert sg.memory_usage_gpu() - base in (0, len(src) * 24)
    assert sg.memory_usage_gpu() == len(src) * (56 + 24)  # points + covariances + normals: 56 B per point
    with ctx.diag("plane=0"):
        fset = api.NonlinearFactorSetGPU(ctx)
        fset.add(api.IntegratedVGICPFactorGPU(0, 1, vm, sg))
        gen = fset.linearize({0: np.eye(4), 1: delta})[0]
    assert gen["num_inliers"] == res["f64"]["num_inliers"]
    assert np.abs(gn_step(gen) - gn_step(res["f64"])).max() < 1e-5
    # a cloud whose covariances are NOT of that form stays general even though it has normals
    bad = cs.copy()
    bad[7] = np.eye(3) * 0.5
    g = api.PointCloudGPU.clone(src.astype(np.float64), bad, ns, ctx=ctx)
    b0 = g.memory_usage_gpu()
    fset = api.NonlinearFactorSetGPU(ctx)
    fset.add(api.IntegratedVGICPFactorGPU(0, 1, vm, g))
    r = fset.linearize({0: np.eye(4), 1: delta})[0]
    assert g.memory_usage_gpu() - b0 in (0, len(src) * (36 + 16))
    assert g.memory_usage_gpu() == len(src) * (56 + 36 + 16)  # general stream + stream-ordered normals
    check_factor(r, orc.vgicp_linearize(orc.VoxelMap(0.5).insert(tgt, c32(ct)), src, c32(bad), delta))


def test_config4_dense_depth_frame_300k(api, ctx, orc):
    from glim_amd import synth

    room = synth.Scene.small_room()
    dirs = synth.pinhole_directions(640, 480, 70, 55)
    Ts = [synth.pose(-2.5, -1.5, 1.4, 0.5), synth.pose(-2.45, -1.48, 1.4, 0.51)]
    frames = [synth.scan(room, T, dirs, i, sigma=0.002, max_range=8.0, min_range=0.3) for i, T in enumerate(Ts)]
    assert all(len(f) > 300000 for f in frames)
    clouds, covs = [], []
    for f in frames:
        g = api.PointCloudGPU.clone(f, ctx=ctx)
        nb = g.find_neighbors(10)
        ref_nb = orc.knn(f, 10)
        np.testing.assert_array_equal(nb, ref_nb)  # exact lists incl. order at 307k points
        g.estimate_covariances(10)
        clouds.append(g)
        covs.append(g.download(normals=False)[1].astype(np.float64))
    vm = api.GaussianVoxelMapGPU(0.1, ctx=ctx).insert(clouds[0])
    ref_vm = orc.VoxelMap(0.1).insert(frames[0], covs[0])
    gc, gn, gm, gC = vm.voxels()
    rc, rn, rm, rC = ref_vm.voxels()
    og, orr = np.lexsort(gc.T[::-1]), np.lexsort(rc.T[::-1])
    np.testing.assert_array_equal(gc[og], rc[orr])  # voxel coordinate set
    np.testing.assert_array_equal(gn[og], rn[orr])  # member counts
    np.testing.assert_allclose(gm[og], rm[orr], rtol=2e-7, atol=1e-7)
    delta = synth.relative_pose(Ts[0], Ts[1]) @ orc.se3_exp([1e-3, -2e-3, 1e-3, 5e-3, -4e-3, 2e-3])
    f = api.IntegratedVGICPFactorGPU(Ts[0], 1, vm, clouds[1])  # unary factor against the previous frame
    fset = api.NonlinearFactorSetGPU(ctx)
    fset.add(f)
    values = {1: Ts[0] @ delta}
    got = fset.linearize(values)[0]
    ref = orc.vgicp_linearize(ref_vm, frames[1], covs[1], f.calc_delta(values), want_corr=True)
    check_factor(got, ref, binary=False)
    check_correspondences(fset, 0, f.calc_delta(values), ref)


def test_back_to_back_async_calls_keep_their_own_poses(api, ctx, orc, small_pair):
    """glim_amd_factor_set_linearize_device_async called repeatedly with different poses and no synchronisation in between: every call
    must be evaluated at ITS poses (the pinned staging is a ring guarded by events)."""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")  # caller-owned device memory, allocated the way a C caller of the ABI would

    def dmalloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
        assert hip.hipMemset(p, 0, C.c_size_t(nbytes)) == 0
        return p

    def dread(p, shape):
        a = np.zeros(shape)
        assert hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), p, C.c_size_t(a.nbytes), 2) == 0  # hipMemcpyDeviceToHost
        return a

    t, s = small_pair["target"], small_pair["source"]
    tg = api.PointCloudGPU.clone(t["points"].astype(np.float64), t["covs"], ctx=ctx)
    sg = api.PointCloudGPU.clone(s["points"].astype(np.float64), s["covs"], ctx=ctx)
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
    F, calls = 6, 12
    fset = api.NonlinearFactorSetGPU(ctx)
    for _ in range(F):
        fset.add(api.IntegratedVGICPFactorGPU(0, 1, vm, sg))
    rng = np.random.default_rng(0)
    pose_sets = [np.stack([api.pose12(small_pair["delta"] @ orc.se3_exp(rng.normal(size=6) * 0.02)) for _ in range(F)]) for _ in range(calls)]
    outs = [dmalloc(F * 29 * 8) for _ in range(calls)]
    for P, o in zip(pose_sets, outs):
        fset.linearize_device_async(P, o.value, 0)
    ctx.synchronize()
    for P, o in zip(pose_sets, outs):
        want = fset.linearize_poses(P)
        got = dread(o, (F, 29))
        for f in range(F):
            assert int(round(got[f, 0])) == want[f]["num_inliers"]
            assert got[f, 1] == want[f]["error"]
    # destroying the clouds right after an asynchronous launch is safe (the context quiesces first)
    fset.linearize_device_async(pose_sets[0], outs[0].value, 0)
    sg.close()
    tg.close()
    ctx.synchronize()
    for o in outs:
        hip.hipFree(o)


def test_config3_at_its_own_size_sampled_pairs_match_the_oracle(api, orc):
    """BASELINE configs[3] at the size bench.py times it: 256 merged submaps (merge_frames of 4 keyframes, 0.1 m), 1.0 m maps, ALL 32 640
    pairs in one factor set.  64 pairs spread over the inlier-count distribution (the 16 least overlapping, the 16 most, 32 quantiles)
    are linearised by the FP64 oracle on the downloaded clouds: inlier counts equal, damped Gauss-Newton step within 1e-4, correspondence
    lists bit-exact on 8 of them -- the same check bench.py attaches to its `m2_global256` block (bench.sampled_pair_parity)."""
    import bench
    from glim_amd import multi, synth

    ctx = api.Context(0, 1)
    S = 256
    submaps = bench.make_merged_submaps(api, ctx, S, 4, 40, 512)
    clouds = [g for _, g in submaps]
    poses = [T for T, _ in submaps]
    vmaps = [api.GaussianVoxelMapGPU(1.0, ctx=ctx).insert(c) for c in clouds]
    pairs = [(i, j) for i in range(S) for j in range(i + 1, S)]
    deltas = np.stack([api.pose12(synth.relative_pose(poses[i], poses[j])) for i, j in pairs])
    fset = api.NonlinearFactorSetGPU(ctx)
    for i, j in pairs:
        fset.add(api.IntegratedVGICPFactorGPU(i, j, vmaps[i], clouds[j]))
    out = fset.linearize_poses(deltas)
    records = np.stack([multi.compact_from_linearized(L) for L in out])
    assert len(records) == 32640
    par = bench.sampled_pair_parity(api, fset, range(len(pairs)), pairs, deltas, clouds, records)
    print("configs[3] parity at full size:", par)
    assert par["pairs_checked"] >= 64 and par["inlier_counts_equal"] and par["correspondences_bit_exact"]
    assert par["correspondence_lists_compared"] >= 8 and par["gn_steps_compared"] >= 32
    assert par["max_pose_delta_err"] < 1e-4
    # the batch equals each factor evaluated on its own (a different plan: one chip-wide factor instead of a share of 32 640)
    for k in (0, 12345, len(pairs) - 1):
        i, j = pairs[k]
        alone = api.NonlinearFactorSetGPU(ctx)
        alone.add(api.IntegratedVGICPFactorGPU(i, j, vmaps[i], clouds[j]))
        a = alone.linearize_poses(deltas[k:k + 1])[0]
        assert a["num_inliers"] == out[k]["num_inliers"]
        np.testing.assert_allclose(a["H_ss"], out[k]["H_ss"], rtol=2e-5, atol=1e-6 * np.abs(out[k]["H_ss"]).max())
