"""Edge cases of the HIP path against the oracle: ragged factor sets, bucket spills, voxel-boundary coordinates, duplicates,
degenerate sizes.  (-m gpu)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from glim_amd import api as _api

    assert _api.device_count() >= 1
    return _api


@pytest.fixture(scope="module")
def ctx(api):
    return api.Context(0, 1)


def gn(L):
    return np.linalg.solve(L["H_ss"], -L["b_s"])


def test_ragged_factor_set_sizes(api, ctx, orc, small_pair):
    """One set holding sources of 1, 63, 64, 65, 257, 1000 and 4096 points against two maps: every factor equals its own oracle."""
    t, s = small_pair["target"], small_pair["source"]
    tg = api.PointCloudGPU.clone(t["points"].astype(np.float64), t["covs"], ctx=ctx)
    vms = {r: api.GaussianVoxelMapGPU(r, ctx=ctx).insert(tg) for r in (0.5, 1.0)}
    refs = {r: orc.VoxelMap(r).insert(t["points"], t["covs"]) for r in (0.5, 1.0)}
    sizes = [1, 63, 64, 65, 257, 1000, 4096]
    fset = api.NonlinearFactorSetGPU(ctx)
    keep = []
    values = {0: np.eye(4)}
    for k, n in enumerate(sizes):
        sel = slice(k * 3, k * 3 + n)
        g = api.PointCloudGPU.clone(s["points"][sel].astype(np.float64), s["covs"][sel], ctx=ctx)
        keep.append((g, sel, 0.5 if k % 2 == 0 else 1.0))
        values[k + 1] = small_pair["delta"]
        fset.add(api.IntegratedVGICPFactorGPU(0, k + 1, vms[keep[-1][2]], g))
    out = fset.linearize(values)
    for (g, sel, r), got in zip(keep, out):
        ref = orc.vgicp_linearize(refs[r], s["points"][sel], s["covs"][sel], small_pair["delta"])
        assert got["num_inliers"] == ref["num_inliers"]
        np.testing.assert_allclose(got["H_ss"], ref["H_ss"], rtol=0, atol=3e-4 * max(1e-12, np.abs(ref["H_ss"]).max()))
        np.testing.assert_allclose(got["b_s"], ref["b_s"], rtol=0, atol=3e-4 * max(1e-12, np.abs(ref["b_s"]).max()) + 1e-9)
        np.testing.assert_allclose(got["error"], ref["error"], rtol=3e-4, atol=1e-12)
    errs = fset.error(values)
    for (g, sel, r), e in zip(keep, errs):
        e_ref, _ = orc.vgicp_error(refs[r], s["points"][sel], s["covs"][sel], small_pair["delta"])
        assert e == pytest.approx(e_ref, rel=3e-4, abs=1e-12)


@pytest.mark.parametrize("factor", ["1", "2"])
def test_crowded_bucket_table_spills_stay_exact(api, ctx, orc, small_pair, monkeypatch, factor):
    """diag bucket_factor=1 packs one key per two-way bucket on average: many keys spill to following buckets.  Lookups,
    map contents and the factor must stay exact (the probe compares full keys)."""
    ctx.set_diag(f"bucket_factor={factor}")
    t, s = small_pair["target"], small_pair["source"]
    tg = api.PointCloudGPU.clone(t["points"].astype(np.float64), t["covs"], ctx=ctx)
    sg = api.PointCloudGPU.clone(s["points"].astype(np.float64), s["covs"], ctx=ctx)
    vm = api.GaussianVoxelMapGPU(0.25, ctx=ctx).insert(tg)
    ref = orc.VoxelMap(0.25).insert(t["points"], t["covs"])
    info = vm.voxelmap_info()
    assert info["num_voxels"] == ref.num_voxels()
    assert info["num_buckets"] <= int(factor) * info["num_voxels"] + 16
    gc, gn_, gm, gC = vm.voxels()
    rc, rn, rm, rC = ref.voxels()
    og, orr = np.lexsort(gc.T[::-1]), np.lexsort(rc.T[::-1])
    np.testing.assert_array_equal(gc[og], rc[orr])
    np.testing.assert_array_equal(gn_[og], rn[orr])
    np.testing.assert_allclose(gm[og], rm[orr], rtol=2e-7, atol=1e-7)
    fset = api.NonlinearFactorSetGPU(ctx)
    fset.add(api.IntegratedVGICPFactorGPU(0, 1, vm, sg))
    got = fset.linearize({0: np.eye(4), 1: small_pair["delta"]})[0]
    r = orc.vgicp_linearize(ref, s["points"], s["covs"], small_pair["delta"], want_corr=True)
    assert got["num_inliers"] == r["num_inliers"]
    corr = fset.correspondences(0, small_pair["delta"])
    np.testing.assert_array_equal(corr[:, :3], r["corr"][:, :3])
    np.testing.assert_array_equal(corr[:, 3] > 0, r["corr"][:, 3] >= 0)
    assert np.abs(gn(got) - gn(r)).max() < 1e-4
    assert api.overlap_gpu(vm, sg, small_pair["delta"]) == orc.overlap(ref, s["points"], small_pair["delta"])


def test_points_on_voxel_faces_and_signed_zero(api, ctx, orc):
    """Coordinates exactly on cell faces, +-0, tiny negatives and half-ulp neighbours: voxel coordinates are bit-exact."""
    res = 0.5
    base = np.array([0.0, -0.0, 0.5, -0.5, 1.0, -1.0, 1.5, 2.0, -2.0, 3.0, 1e-30, -1e-30, 0.49999997, 0.50000006, -0.49999997, -0.50000006],
                    dtype=np.float32)
    ulps = np.concatenate([base, np.nextafter(base, np.float32(np.inf)), np.nextafter(base, np.float32(-np.inf))])
    rng = np.random.default_rng(11)
    pts = np.stack([rng.permutation(ulps), rng.permutation(ulps), rng.permutation(ulps)], axis=1).astype(np.float32)
    covs = np.tile(np.eye(3), (len(pts), 1, 1))
    g = api.PointCloudGPU.clone(pts.astype(np.float64), covs, ctx=ctx)
    vm = api.GaussianVoxelMapGPU(res, ctx=ctx).insert(g)
    ref = orc.VoxelMap(res).insert(pts, covs)
    gc, gcnt, _, _ = vm.voxels()
    rc, rcnt, _, _ = ref.voxels()
    og, orr = np.lexsort(gc.T[::-1]), np.lexsort(rc.T[::-1])
    np.testing.assert_array_equal(gc[og], rc[orr])
    np.testing.assert_array_equal(gcnt[og], rcnt[orr])
    # correspondences of the same points under poses that put many of them exactly on faces
    fset = api.NonlinearFactorSetGPU(ctx)
    fset.add(api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, g))
    for shift in ([0, 0, 0], [0.5, -0.5, 1.0], [0.25, 0.25, 0.25], [-1e-7, 1e-7, 0.0]):
        T = np.eye(4)
        T[:3, 3] = shift
        corr = fset.correspondences(0, T)
        r = orc.vgicp_linearize(ref, pts, covs, T, want_corr=True)
        np.testing.assert_array_equal(corr[:, :3], r["corr"][:, :3])
        np.testing.assert_array_equal(corr[:, 3] > 0, r["corr"][:, 3] >= 0)
        assert fset.linearize({1: T})[0]["num_inliers"] == r["num_inliers"]


def test_duplicate_points_and_single_voxel(api, ctx, orc):
    """All points identical: one voxel, its mean is the point, covariance the common covariance; residual exactly zero at identity."""
    p = np.tile(np.array([[1.3, -2.2, 0.7]], dtype=np.float32), (500, 1))
    C = np.diag([0.3, 0.2, 0.1])
    covs = np.tile(C, (500, 1, 1))
    g = api.PointCloudGPU.clone(p.astype(np.float64), covs, ctx=ctx)
    vm = api.GaussianVoxelMapGPU(1.0, ctx=ctx).insert(g)
    coords, counts, means, vcovs = vm.voxels()
    assert len(coords) == 1 and counts[0] == 500
    np.testing.assert_allclose(means[0], p[0], atol=1e-6)
    np.testing.assert_allclose(vcovs[0], C, atol=1e-7)
    fset = api.NonlinearFactorSetGPU(ctx)
    fset.add(api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, g))
    L = fset.linearize({1: np.eye(4)})[0]
    assert L["num_inliers"] == 500 and L["error"] < 1e-9
    ref = orc.vgicp_linearize(orc.VoxelMap(1.0).insert(p, covs), p, covs, np.eye(4))
    np.testing.assert_allclose(L["H_ss"], ref["H_ss"], rtol=1e-5, atol=1e-3)
    # kNN with all-equal distances: ties resolved by index, like the oracle
    np.testing.assert_array_equal(g.find_neighbors(10)[:20], orc.knn(p, 10, method="brute")[:20])


def test_far_from_origin_keeps_precision(api, ctx, orc, small_pair):
    """A map 100 km from the origin: coordinates still exact, GN step still within tolerance (means are stored centre-relative)."""
    t, s = small_pair["target"], small_pair["source"]
    off = np.array([1.0e5, -7.5e4, 2.0e3])
    tp = (t["points"].astype(np.float64) + off).astype(np.float32)
    T = small_pair["delta"].copy()
    T[:3, 3] += off
    tg = api.PointCloudGPU.clone(tp.astype(np.float64), t["covs"], ctx=ctx)
    sg = api.PointCloudGPU.clone(s["points"].astype(np.float64), s["covs"], ctx=ctx)
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
    ref = orc.VoxelMap(0.5).insert(tp, t["covs"])
    assert vm.voxelmap_info()["num_voxels"] == ref.num_voxels()
    fset = api.NonlinearFactorSetGPU(ctx)
    fset.add(api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, sg))
    got = fset.linearize({1: T})[0]
    r = orc.vgicp_linearize(ref, s["points"], s["covs"], T, want_corr=True)
    assert got["num_inliers"] == r["num_inliers"] > 100
    corr = fset.correspondences(0, T)
    np.testing.assert_array_equal(corr[:, :3], r["corr"][:, :3])
    assert np.abs(gn(got) - gn(r)).max() < 1e-4


def test_device_memory_pool_can_be_disabled(api, monkeypatch):
    """The memory caches are a process-wide switch (GLIM_AMD_DIAG="pool=0"); here we only check that repeated create/destroy cycles do not leak or crash."""
    c2 = api.Context(0, 1)
    for n in (10, 1000, 100000):
        for _ in range(3):
            g = api.PointCloudGPU.clone(np.random.default_rng(n).normal(size=(n, 3)).astype(np.float32), ctx=c2)
            g.find_neighbors(5, download=False)
            g.estimate_covariances(5)
            vm = api.GaussianVoxelMapGPU(0.5, ctx=c2).insert(g)
            assert vm.voxelmap_info()["num_voxels"] > 0
            vm.close()
            g.close()
    c2.close()


@pytest.mark.gpu
def test_context_refuses_to_die_before_its_children(small_pair):
    """SURVEY 8b ownership: destroying a context with live handles is an error code, not undefined behaviour."""
    from glim_amd import api

    ctx = api.Context(0, 1)
    g = api.PointCloudGPU.clone(small_pair["source"]["points"], small_pair["source"]["covs"], ctx=ctx)
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(g)
    fs = api.NonlinearFactorSetGPU(ctx)
    with pytest.raises(api.GlimAmdError):
        ctx.close()
    assert g.size() == len(small_pair["source"]["points"])  # still usable
    fs.close()
    vm.close()
    with pytest.raises(api.GlimAmdError):
        ctx.close()
    g.close()
    ctx.close()  # now fine


def test_invalid_caller_input_is_an_error_code_not_a_fault(api, ctx, small_pair):
    """Host-supplied neighbour indices outside [0, n) are refused (GLIM_AMD_ERR_INVALID); handles of ANOTHER CONTEXT of the same device are fine
    (GLIM's modules own a stream pool each and hand frames and maps to one another)."""
    pts = small_pair["source"]["points"][:500]
    g = api.PointCloudGPU.clone(pts, ctx=ctx)
    nb = np.tile(np.arange(5, dtype=np.int32), (500, 1))
    g.set_neighbors(nb)  # valid
    for bad in (500, -1, 2**31 - 1):
        nb2 = nb.copy()
        nb2[123, 2] = bad
        with pytest.raises(api.GlimAmdError):
            g.set_neighbors(nb2)
    g.estimate_covariances(5)  # the valid lists are still in place
    other = api.Context(0, 1)
    g2 = api.PointCloudGPU.clone(pts, small_pair["source"]["covs"][:500], ctx=other)
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(g)
    g_same = api.PointCloudGPU.clone(pts, small_pair["source"]["covs"][:500], ctx=ctx)
    assert api.overlap_gpu(vm, g2, np.eye(4), ctx=ctx) == api.overlap_gpu(vm, g_same, np.eye(4), ctx=ctx)
    assert api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(g2).voxelmap_info()["num_voxels"] == api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(g_same).voxelmap_info()["num_voxels"]
    g2.close()
    other.close()


# ---- plan cache, host-finalised single-factor call, batched overlap -----------------------------------------------------------------


def _full_size_pair(api, ctx, rings=64, azimuths=512):
    from glim_amd import synth

    scene = synth.Scene.default()
    dirs = synth.lidar_directions(rings, azimuths)
    poses = synth.arc_trajectory(2)
    tgt, src = synth.scan(scene, poses[0], dirs, 0), synth.scan(scene, poses[1], dirs, 1)
    tg, sg = api.PointCloudGPU.clone(tgt, ctx=ctx), api.PointCloudGPU.clone(src, ctx=ctx)
    for g in (tg, sg):
        g.find_neighbors(10, download=False)
        g.estimate_covariances(10)
    return tgt, src, tg, sg, synth.relative_pose(poses[0], poses[1])


def test_synchronous_call_variants_give_identical_bits(api, ctx, orc):
    """The fast paths of the synchronous call -- pose + descriptor in the kernel arguments (single factor), poses read from host-mapped memory
    (small sets), completion word instead of a stream synchronise -- against the plain path with each of them switched off: identical bits,
    linearise and error, unary and binary, a chip-wide single factor, a small single factor and a 34-factor set shaped like the odometry's,
    repeated (a re-used plan, a re-used pose slot)."""
    for rings, azimuths, copies in ((64, 512, 1), (16, 128, 1), (16, 256, 34)):
        tgt, src, tg, sg, delta = _full_size_pair(api, ctx, rings, azimuths)
        vms = [api.GaussianVoxelMapGPU(r, ctx=ctx).insert(tg) for r in (0.5, 1.0)]
        for target in (0, np.eye(4)):
            factors = [api.IntegratedVGICPFactorGPU(target, 1 + k, vms[k % 2], sg) for k in range(copies)]
            res = {}
            for mode in ("", "resident=1", "fuse=0", "poll=0", "host_poses=0", "inline_pose=0", "fuse=0,inline_pose=0", "inline_pose=0,host_poses=0,poll=0"):
                with ctx.diag(mode):  # (each mode on top of the DEFAULTS: set_diag alone would pile the switches up)
                    fset = api.NonlinearFactorSetGPU(ctx)
                    for f in factors:
                        fset.add(f)
                    for rep in range(5):
                        values = {0: np.eye(4)}
                        for k in range(copies):
                            values[1 + k] = delta @ orc.se3_exp(np.array([0.002, -0.001, 0.003, 0.02, 0.01, -0.02]) * (1 + 0.1 * k + 0.05 * rep))
                        res[(mode, rep)] = (fset.linearize(values), fset.error(values))
            assert res[("", 0)][0][0]["num_inliers"] > 100
            for (mode, rep), (Ls, es) in res.items():
                base = res[("inline_pose=0,host_poses=0,poll=0", rep)]
                for k, (L, B) in enumerate(zip(Ls, base[0])):
                    assert L["num_inliers"] == B["num_inliers"], mode
                    for key in ("H_ss", "b_s", "H_tt", "H_ts", "b_t"):
                        np.testing.assert_array_equal(L[key], B[key], err_msg=f"{mode} rep {rep} factor {k} {key}")
                    assert L["error"] == B["error"], mode
                assert list(es) == list(base[1]), mode


def test_single_dispatch_form_under_repetition_and_mixed_segments(api, ctx, orc, small_pair):
    """The single-dispatch synchronous form (tagged partial rows handed to finalising blocks inside the launch, vgicp.hip FUSED) re-uses its rows call
    after call: 400 back-to-back calls alternating between two poses must return, every time, exactly the record of THAT pose as the two-dispatch
    form (fuse=0) computes it -- a stale row or a stale tag would surface as the other pose's bits.  Chip-wide single factor, the odometry's
    34-factor shape, and a set that mixes plane-form and general (merged-submap-like) sources, i.e. two launch segments with their own finalisers."""
    t, s = small_pair["target"], small_pair["source"]
    tgt, src, tg, sg, delta = _full_size_pair(api, ctx, 128, 1024)
    vms = [api.GaussianVoxelMapGPU(r, ctx=ctx).insert(tg) for r in (0.5, 1.0)]
    general = api.PointCloudGPU.clone(s["points"].astype(np.float64), s["covs"] * 1.5, ctx=ctx)  # not plane-form: 36 B/pt kernel
    small_t = api.PointCloudGPU.clone(t["points"].astype(np.float64), t["covs"], ctx=ctx)
    small_vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(small_t)
    cases = {
        "single": [api.IntegratedVGICPFactorGPU(0, 1, vms[0], sg)],
        "set34": [api.IntegratedVGICPFactorGPU(0 if k < 4 else np.eye(4), 1 + k, vms[k % 2], sg) for k in range(34)],
        "mixed": [api.IntegratedVGICPFactorGPU(0, 1 + k, small_vm if k % 3 == 0 else vms[k % 2], general if k % 3 == 0 else sg) for k in range(7)],
    }
    for name, factors in cases.items():
        poses = []
        for which in (0, 1):
            values = {0: np.eye(4)}
            for k in range(len(factors)):
                d = small_pair["delta"] if (name == "mixed" and k % 3 == 0) else delta
                values[1 + k] = d @ orc.se3_exp(np.array([0.002, -0.001, 0.003, 0.02, 0.01, -0.02]) * (1 + 0.1 * k + 3.0 * which))
            poses.append(values)
        with ctx.diag("fuse=0"):
            ref_set = api.NonlinearFactorSetGPU(ctx)
            for f in factors:
                ref_set.add(f)
            want = [(ref_set.linearize(v), ref_set.error(v)) for v in poses]
        fset = api.NonlinearFactorSetGPU(ctx)
        for f in factors:
            fset.add(f)
        assert want[0][0][0]["num_inliers"] > 100 and want[0][0][0]["H_ss"][0, 0] != want[1][0][0]["H_ss"][0, 0]
        for rep in range(400):
            which = (rep * 7 // 3) % 2
            Ls = fset.linearize(poses[which])
            es = fset.error(poses[which]) if rep % 5 == 0 else None
            for k, (L, B) in enumerate(zip(Ls, want[which][0])):
                assert L["num_inliers"] == B["num_inliers"], (name, rep, k)
                for key in ("H_ss", "b_s", "H_tt", "H_ts", "b_t"):
                    np.testing.assert_array_equal(L[key], B[key], err_msg=f"{name} rep {rep} factor {k} {key}")
                assert L["error"] == B["error"], (name, rep, k)
            if es is not None:
                assert list(es) == list(want[which][1]), (name, rep)


def test_wave_per_factor_finalise_gives_the_block_finalise_bits(api, ctx, orc, small_pair):
    """Sets of more than 2 048 short factors (configs[3]'s 32 640 pairs) are finalised by one wavefront per factor instead of one block per
    factor: the records must carry the same bits as the same factors evaluated in sets small enough for the block kernel (points per thread
    pinned, so that both plans cut every factor into the same rows)."""
    t, s = small_pair["target"], small_pair["source"]
    tg = api.PointCloudGPU.clone(t["points"].astype(np.float64), t["covs"], ctx=ctx)
    clouds = [api.PointCloudGPU.clone(s["points"][k::3].astype(np.float64), s["covs"][k::3], ctx=ctx) for k in range(3)]
    vms = [api.GaussianVoxelMapGPU(r, ctx=ctx).insert(tg) for r in (0.5, 1.0)]
    nf = 2100
    factors = [api.IntegratedVGICPFactorGPU(0, 1 + k, vms[k % 2], clouds[k % 3]) for k in range(nf)]
    values = {0: np.eye(4)}
    for k in range(nf):
        values[1 + k] = small_pair["delta"] @ orc.se3_exp(np.array([0.002, -0.001, 0.003, 0.02, 0.01, -0.02]) * (1 + 0.01 * k))

    def run(lo, hi):
        fs = api.NonlinearFactorSetGPU(ctx)
        for f in factors[lo:hi]:
            fs.add(f)
        out = fs.linearize(values), fs.error(values)
        fs.close()
        return out

    with ctx.diag("ppt=2"):
        big = run(0, nf)
        parts = [run(lo, min(nf, lo + 700)) for lo in range(0, nf, 700)]
    want_L = [L for part in parts for L in part[0]]
    want_e = [e for part in parts for e in part[1]]
    assert len(want_L) == nf and sum(L["num_inliers"] for L in want_L) > 100 * nf
    for k, (a, b) in enumerate(zip(big[0], want_L)):
        assert a["num_inliers"] == b["num_inliers"] and a["error"] == b["error"], k
        for key in ("H_ss", "b_s", "H_tt", "H_ts", "b_t"):
            np.testing.assert_array_equal(a[key], b[key], err_msg=f"factor {k} {key}")
    assert list(big[1]) == want_e


def test_plan_cache_serves_fresh_sets_and_follows_object_identity(api, ctx, orc, small_pair):
    """GLIM builds a fresh NonlinearFactorSetGPU per linearisation: sets with the same (map, cloud, flags) list share one cached plan and
    give the same records as a set built with the cache off; destroying / rebuilding an object never resurrects a stale plan."""
    t, s = small_pair["target"], small_pair["source"]
    tg = api.PointCloudGPU.clone(t["points"].astype(np.float64), t["covs"], ctx=ctx)
    clouds = [api.PointCloudGPU.clone(s["points"][k::3].astype(np.float64), s["covs"][k::3], ctx=ctx) for k in range(3)]
    vms = [api.GaussianVoxelMapGPU(r, ctx=ctx).insert(tg) for r in (0.5, 1.0)]
    factors = [api.IntegratedVGICPFactorGPU(0, 1 + k, vms[k % 2], clouds[k]) for k in range(3)]
    values = {0: np.eye(4), 1: small_pair["delta"], 2: small_pair["delta"], 3: small_pair["delta"]}

    def fresh():
        fs = api.NonlinearFactorSetGPU(ctx)
        for f in factors:
            fs.add(f)
        out = fs.linearize(values)
        fs.close()
        return out

    with ctx.diag("plan_cache=0"):
        want = fresh()
    for _ in range(4):  # the second and later iterations adopt the plan the first one parked
        got = fresh()
        for a, b in zip(got, want):
            assert a["num_inliers"] == b["num_inliers"]
            np.testing.assert_array_equal(a["H_ss"], b["H_ss"])
            np.testing.assert_array_equal(a["b_s"], b["b_s"])
    # clear() + add() on one set object, a different list in between
    fs = api.NonlinearFactorSetGPU(ctx)
    for rounds in range(3):
        fs.clear()
        for f in (factors if rounds != 1 else factors[:2]):
            fs.add(f)
        got = fs.linearize(values)
        for a, b in zip(got, want):
            np.testing.assert_array_equal(a["H_ss"], b["H_ss"])
    # a new cloud / map at (possibly) the same address must not match the old plan: replace the second source by different points
    clouds[1].close()
    clouds[1] = api.PointCloudGPU.clone(s["points"][2::5].astype(np.float64), s["covs"][2::5], ctx=ctx)
    factors[1] = api.IntegratedVGICPFactorGPU(0, 2, vms[1], clouds[1])
    ref = orc.vgicp_linearize(orc.VoxelMap(1.0).insert(t["points"], t["covs"]), s["points"][2::5], s["covs"][2::5], small_pair["delta"])
    got = fresh()[1]
    assert got["num_inliers"] == ref["num_inliers"]
    assert np.abs(gn(got) - gn(ref)).max() < 1e-4


def test_a_new_list_of_the_evicted_plans_shape_takes_over_its_buffers(api, orc, small_pair):
    """GLIM's odometry brings a NEW factor list with every frame (the new cloud against the same number of targets).  Once the context's
    plan cache is full, the plan it would evict next hands its buffers to the new list when the shape is the same: same bits as a context
    that never recycles, lists of another shape in between included, three linearisations per list (tags keep counting across lives)."""
    t, s = small_pair["target"], small_pair["source"]
    a_ctx, b_ctx = api.Context(0, 2), api.Context(0, 2)
    b_ctx.set_diag("plan_recycle=0")
    n = 1500
    frames = [(s["points"][k:k + n].astype(np.float64), s["covs"][k:k + n]) for k in range(0, 40 * 7, 7)]  # 40 different clouds of one size
    results = {}
    for name, c in (("recycle", a_ctx), ("plain", b_ctx)):
        tg = api.PointCloudGPU.clone(t["points"].astype(np.float64), t["covs"], ctx=c)
        vms = [api.GaussianVoxelMapGPU(r, ctx=c).insert(tg) for r in (0.5, 1.0)]
        out = []
        for k, (p, cv) in enumerate(frames):
            g = api.PointCloudGPU.clone(p if k % 9 != 4 else p[:700], cv if k % 9 != 4 else cv[:700], ctx=c)  # (every ninth list has another shape)
            values = {0: np.eye(4), 1: small_pair["delta"]}
            for it in range(3):  # a fresh set per optimiser iteration, as odometry_estimation_gpu.cpp:383-385 drives it
                fs = api.NonlinearFactorSetGPU(c)
                for vm in vms:
                    fs.add(api.IntegratedVGICPFactorGPU(0, 1, vm, g))
                if k % 5 == 3:
                    fs.add(api.IntegratedVGICPFactorGPU(0, 1, vms[0], g))  # (and every fifth one more factor)
                values[1] = small_pair["delta"] @ np.block([[np.eye(3), np.full((3, 1), 1e-3 * it)], [np.zeros((1, 3)), np.ones((1, 1))]])
                out.append(fs.linearize(values))
                fs.close()
            g.close()
        results[name] = out
        stats = api.plan_stats(c)
        assert stats["built"] == len(frames)
        if name == "recycle":
            assert stats["recycled"] >= 10, stats
        else:
            assert stats["recycled"] == 0
    for a, b in zip(results["recycle"], results["plain"]):
        for x, y in zip(a, b):
            assert x["num_inliers"] == y["num_inliers"]
            np.testing.assert_array_equal(x["H_ss"], y["H_ss"])
            np.testing.assert_array_equal(x["b_s"], y["b_s"])
            assert x["error"] == y["error"]
    # one of them against the oracle
    ref = orc.vgicp_linearize(orc.VoxelMap(0.5).insert(t["points"], t["covs"]), frames[-1][0], frames[-1][1], small_pair["delta"])
    got = results["recycle"][-3][0]
    assert got["num_inliers"] == ref["num_inliers"]
    assert np.abs(gn(got) - gn(ref)).max() < 1e-4


def test_reestimating_covariances_rebuilds_plans_that_streamed_the_old_ones(api, ctx, orc):
    """A cloud uploaded with general covariances sits in a LIVE factor set (its plan holds the addresses of the 36 B/pt streams); then
    glim_amd_cloud_estimate_covariances replaces the covariances (plane form, streams freed).  The next linearise of the same set must run
    on the new data (it used to read freed memory with plane = 0)."""
    tgt, src, tg, sg, delta = _full_size_pair(api, ctx, 32, 256)
    _, cov_dev, _ = sg.download()
    blur = 0.5 * (cov_dev + np.roll(cov_dev, 1, axis=0)).astype(np.float64)  # not of the plane form: the general kernel
    up = api.PointCloudGPU.clone(src.astype(np.float64), blur, ctx=ctx)
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
    fset = api.NonlinearFactorSetGPU(ctx)
    fset.add(api.IntegratedVGICPFactorGPU(0, 1, vm, up))
    values = {0: np.eye(4), 1: delta}
    first = fset.linearize(values)[0]
    _, ct, _ = tg.download()
    ref_map = orc.VoxelMap(0.5).insert(tgt, ct.astype(np.float64))
    ref_blur = orc.vgicp_linearize(ref_map, src, blur.astype(np.float32).astype(np.float64), delta)
    assert first["num_inliers"] == ref_blur["num_inliers"] and np.abs(gn(first) - gn(ref_blur)).max() < 1e-4
    up.find_neighbors(10, download=False)
    up.estimate_covariances(10)
    second = fset.linearize(values)[0]  # same set object, nothing re-added
    ref_plane = orc.vgicp_linearize(ref_map, src, cov_dev.astype(np.float64), delta)
    assert second["num_inliers"] == ref_plane["num_inliers"]
    assert np.abs(gn(second) - gn(ref_plane)).max() < 1e-4
    assert abs(second["error"] - ref_plane["error"]) <= 3e-4 * abs(ref_plane["error"])


def test_overlap_batch_equals_single_calls_and_oracle(api, ctx, orc, small_pair):
    t, s = small_pair["target"], small_pair["source"]
    tg = api.PointCloudGPU.clone(t["points"].astype(np.float64), t["covs"], ctx=ctx)
    sg = api.PointCloudGPU.clone(s["points"].astype(np.float64), s["covs"], ctx=ctx)
    empty = api.PointCloudGPU.clone(np.zeros((0, 3), dtype=np.float32), ctx=ctx)
    vms = [api.GaussianVoxelMapGPU(r, ctx=ctx).insert(tg) for r in (0.5, 1.0, 2.0)]
    refs = [orc.VoxelMap(r).insert(t["points"], t["covs"]) for r in (0.5, 1.0, 2.0)]
    delta = small_pair["delta"]
    far = np.eye(4)
    far[:3, 3] = 1e4
    shifted = delta @ orc.se3_exp([0, 0, 0.3, 2.0, 1.0, 0])
    many = [vms[k % 3] for k in range(20)]  # more targets than fit the kernel arguments: the staged path
    many_refs = [refs[k % 3] for k in range(20)]
    many_T = [far] * 19 + [shifted]
    queries = [([vms[0]], sg, [delta]), ([vms[0], vms[1]], sg, [far, shifted]), ([vms[2]], tg, [np.eye(4)]), ([vms[1]], empty, [delta]),
               (many, sg, many_T), ([vms[0]], sg, [far])]
    want = [orc.overlap(refs[0], s["points"], delta), orc.overlap([refs[0], refs[1]], s["points"], [far, shifted]), orc.overlap(refs[2], t["points"], np.eye(4)), 0.0,
            orc.overlap(many_refs, s["points"], many_T), 0.0]
    assert api.overlap_gpu_batch(queries, ctx=ctx) == want
    assert api.overlap_gpu_batch(queries, ctx=ctx) == want  # counters were reset by the kernel
    assert [api.overlap_gpu(q[0], q[1], q[2]) for q in queries] == want
    assert api.overlap_gpu_batch(queries[:1], ctx=ctx) == want[:1]
    assert api.overlap_gpu_batch([queries[4]], ctx=ctx) == [want[4]]


def test_frame_create_equals_the_separate_calls(api, orc, small_pair):
    """glim_amd_frame_create (create_frame of the GPU odometry as one submission, one synchronise): the cloud, its plane-form verdict, both voxel maps
    and a factor over them are bit for bit what PointCloudGPU::clone + two GaussianVoxelMapGPU::insert give; frames that do not fit the small-cloud
    path (no covariances, > 32 768 points) take the separate calls inside the same entry point."""
    ctx = api.Context(0, 1)
    t, s = small_pair["target"], small_pair["source"]

    def packed(d, covs=True, normals=True):
        n = len(d["points"])
        p4 = np.ones((n, 4))
        p4[:, :3] = d["points"][:, :3]
        c16 = n4 = None
        if covs:
            m = np.zeros((n, 4, 4))
            m[:, :3, :3] = d["covs"][:, :3, :3]
            c16 = np.ascontiguousarray(np.transpose(m, (0, 2, 1))).reshape(n, 16)
        if normals:
            n4 = np.zeros((n, 4))
            n4[:, :3] = d["normals"][:, :3]
        return p4, c16, n4

    levels = [0.5, 1.0]
    sg = api.PointCloudGPU.clone_packed(*packed(s), ctx=ctx)
    # (the second and later rounds take pre-cleared tables from the maps the first one dropped; the switches: one launch for the pull and every
    # level's build + one for every level's records against 1 + 2 per level; the pull kernel launched before / after the host conversion)
    for reps, switches in enumerate(("", "", "", "frame_fused=0", "pull_gated=0", "frame_fused=0,pull_gated=0", "")):
        ctx.set_diag("")  # (set_diag adds to the switches in force: back to the defaults first)
        if switches:
            ctx.set_diag(switches)
        p4, c16, n4 = packed(t)
        cloud, maps = api.frame_create(p4, c16, n4, levels, ctx=ctx)
        ref_cloud = api.PointCloudGPU.clone_packed(p4, c16, n4, ctx=ctx)
        ref_maps = [api.GaussianVoxelMapGPU(r, ctx=ctx).insert(ref_cloud) for r in levels]
        for a, b in zip(cloud.download(), ref_cloud.download()):
            np.testing.assert_array_equal(a, b)
        for m, r in zip(maps, ref_maps):
            assert m.voxelmap_info() == r.voxelmap_info()
            (ca, na, ma, va), (cb, nb, mb, vb) = m.voxels(), r.voxels()
            oa, ob = np.lexsort(ca.T[::-1]), np.lexsort(cb.T[::-1])
            np.testing.assert_array_equal(ca[oa], cb[ob])
            np.testing.assert_array_equal(na[oa], nb[ob])
            np.testing.assert_array_equal(ma[oa], mb[ob])
            np.testing.assert_array_equal(va[oa], vb[ob])
            out = []
            for vm in (m, r):
                fs = api.NonlinearFactorSetGPU(ctx)
                fs.add(api.IntegratedVGICPFactorGPU(0, 1, vm, sg))
                out.append(fs.linearize({0: np.eye(4), 1: small_pair["delta"]})[0])
                fs.close()
            assert out[0]["num_inliers"] == out[1]["num_inliers"] > 100 and out[0]["error"] == out[1]["error"]
            np.testing.assert_array_equal(out[0]["H_ss"], out[1]["H_ss"])
        for x in maps + ref_maps:
            x.close()
        cloud.close()
        ref_cloud.close()
    ctx.set_diag("")
    # the other routes of the same entry point
    p4, c16, n4 = packed(t, covs=True, normals=False)
    cloud, maps = api.frame_create(p4, c16, None, levels, ctx=ctx)
    assert [m.voxelmap_info()["num_voxels"] for m in maps] == [orc.VoxelMap(r).insert(t["points"], t["covs"]).num_voxels() for r in levels]
    reps = -(-40000 // len(p4))
    big, bigc = np.tile(p4, (reps, 1)), np.tile(c16, (reps, 1))  # > 32 768 points: the separate calls inside the same entry point
    cloud2, maps2 = api.frame_create(big, bigc, None, levels[:1], ctx=ctx)
    assert cloud2.size() == len(big) > 32768 and maps2[0].voxelmap_info()["num_voxels"] == maps[0].voxelmap_info()["num_voxels"]
    with pytest.raises(api.GlimAmdError):
        api.frame_create(p4, c16, None, [0.5, -1.0], ctx=ctx)
    far = p4.copy()
    far[0, 0] = 1e9  # a voxel coordinate outside the key range: nothing is created
    with pytest.raises(api.GlimAmdError):
        api.frame_create(far, c16, None, levels, ctx=ctx)
    for x in maps + maps2:
        x.close()
    cloud.close()
    cloud2.close()
    sg.close()
    ctx.close()  # (refused if the failed calls had left anything behind)


def test_stale_density_hint_does_not_leave_an_overfull_table(api, orc):
    """the direct build of a > 32 768-point cloud sizes its table from the voxels-per-point ratio of the PREVIOUS map at that resolution.
    A dense scan (ratio ~0.1) followed by a cloud that was already downsampled at about the voxel size (ratio ~1) would leave the floor table of
    N / 2 two-way buckets 100 % full for the whole life of the map; the build must notice and fall back to the counting path (6 buckets per voxel
    present).  Contents stay exact either way."""
    ctx = api.Context(0, 1)
    rng = np.random.default_rng(9)
    res = 0.5
    # dense: 65 536 points inside a 10 m cube -> ~8000 voxels (ratio 0.12)
    dense = rng.uniform(-5, 5, size=(65536, 3)).astype(np.float32)
    cov = np.tile((np.eye(3) * 1e-2).astype(np.float32), (65536, 1, 1))
    dg = api.PointCloudGPU.clone(dense, cov, ctx=ctx)
    api.GaussianVoxelMapGPU(res, ctx=ctx).insert(dg)                      # first map at this resolution: counting path, leaves the hint
    hinted = api.GaussianVoxelMapGPU(res, ctx=ctx).insert(dg).voxelmap_info()  # direct build with a good hint
    assert hinted["num_voxels"] <= 0.35 * 2 * hinted["num_buckets"]
    # sparse: one point per voxel on a 40^3 lattice (64 000 points, ratio 1.0) at the same resolution
    g = np.stack(np.meshgrid(*[np.arange(40)] * 3, indexing="ij"), -1).reshape(-1, 3)
    sparse = ((g + 0.5) * res + rng.uniform(-0.2, 0.2, size=g.shape)).astype(np.float32)
    cov2 = np.tile((np.eye(3) * 1e-2).astype(np.float32), (len(sparse), 1, 1))
    sg = api.PointCloudGPU.clone(sparse, cov2, ctx=ctx)
    vm = api.GaussianVoxelMapGPU(res, ctx=ctx).insert(sg)
    info = vm.voxelmap_info()
    assert info["num_voxels"] == 64000
    assert info["num_voxels"] <= 0.35 * 2 * info["num_buckets"], info   # without the check: 64 000 keys in 32 768 two-way buckets
    ref = orc.VoxelMap(res).insert(sparse, cov2.astype(np.float64))
    assert ref.num_voxels() == 64000
    coords, counts, means, covs = vm.voxels()
    assert len(np.unique(coords, axis=0)) == 64000 and np.all(counts == 1)
    # and the next sparse map builds directly with the corrected hint
    again = api.GaussianVoxelMapGPU(res, ctx=ctx).insert(sg).voxelmap_info()
    assert again["num_voxels"] == 64000 and again["num_voxels"] <= 0.35 * 2 * again["num_buckets"]


def test_resident_session_lifecycle(api, ctx, orc, small_pair):
    """The resident session (vgicp.hip resident_kernel): bits equal to the launch-per-call forms from the first request it serves; it idles out and
    is restarted transparently; a second plan takes it over once the first has gone quiet; destroying the served set's inputs / the context while it
    is alive is safe; error() and asynchronous calls keep working next to it."""
    import time

    tgt, src, tg, sg, delta = _full_size_pair(api, ctx, 64, 512)
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
    vm2 = api.GaussianVoxelMapGPU(1.0, ctx=ctx).insert(tg)
    poses = [{0: np.eye(4), 1: delta @ orc.se3_exp(np.array([0.002, -0.001, 0.003, 0.02, 0.01, -0.02]) * (1 + k))} for k in range(3)]
    with ctx.diag("resident=0,fuse=0"):
        ref = api.NonlinearFactorSetGPU(ctx)
        ref.add(api.IntegratedVGICPFactorGPU(0, 1, vm, sg))
        want = [ref.linearize(v)[0] for v in poses]
        ref2 = api.NonlinearFactorSetGPU(ctx)
        ref2.add(api.IntegratedVGICPFactorGPU(0, 1, vm2, sg))
        want2 = [ref2.linearize(v)[0] for v in poses]
        want_err = ref.error(poses[1])
    # the session is opt-in (a context created with priority 1, or resident=1): without either, no session is ever started
    api.resident_stop(ctx)
    before = api.resident_stats(ctx)
    plain = api.NonlinearFactorSetGPU(ctx)
    plain.add(api.IntegratedVGICPFactorGPU(0, 1, vm, sg))
    for rep in range(8):
        plain.linearize(poses[rep % 3])
    assert api.resident_stats(ctx)["launches"] == before["launches"] and not api.resident_stats(ctx)["alive"]
    plain.close()

    def same(L, B, what):
        assert L["num_inliers"] == B["num_inliers"], what
        for key in ("H_ss", "b_s", "H_tt", "H_ts", "b_t"):
            np.testing.assert_array_equal(L[key], B[key], err_msg=f"{what} {key}")
        assert L["error"] == B["error"], what

    with ctx.diag("resident=1,resident_idle_us=500"):
        fset = api.NonlinearFactorSetGPU(ctx)
        fset.add(api.IntegratedVGICPFactorGPU(0, 1, vm, sg))
        for rep in range(40):  # calls 4.. are served by the session
            same(fset.linearize(poses[rep % 3])[0], want[rep % 3], f"rep {rep}")
        assert fset.error(poses[1]) == want_err  # a launch-per-call evaluation next to the live session
        same(fset.linearize(poses[2])[0], want[2], "after error()")
        time.sleep(0.05)  # far beyond the idle time-out: the kernel has left; the next call restarts it
        for rep in range(10):
            same(fset.linearize(poses[rep % 3])[0], want[rep % 3], f"after idle rep {rep}")
        # a second plan: served launch-per-call while the first is hot, through the session once that has gone quiet
        other = api.NonlinearFactorSetGPU(ctx)
        other.add(api.IntegratedVGICPFactorGPU(0, 1, vm2, sg))
        for rep in range(12):
            same(fset.linearize(poses[rep % 3])[0], want[rep % 3], f"interleaved a{rep}")
            same(other.linearize(poses[rep % 3])[0], want2[rep % 3], f"interleaved b{rep}")
        time.sleep(0.01)
        for rep in range(12):
            same(other.linearize(poses[rep % 3])[0], want2[rep % 3], f"takeover {rep}")
        # fresh sets with the same factor list adopt the plan -- and its session (GLIM's per-iteration sets)
        for rep in range(8):
            fresh = api.NonlinearFactorSetGPU(ctx)
            fresh.add(api.IntegratedVGICPFactorGPU(0, 1, vm2, sg))
            same(fresh.linearize(poses[rep % 3])[0], want2[rep % 3], f"fresh {rep}")
            fresh.close()
        # tearing things down under a live session
        other.close()
        fset.close()
        c2 = api.Context(0, 1, priority=1)  # (the odometry's kind of context: the session is on by itself)
        t2 = api.PointCloudGPU.clone(small_pair["target"]["points"].astype(np.float64), small_pair["target"]["covs"], ctx=c2)
        s2 = api.PointCloudGPU.clone(small_pair["source"]["points"].astype(np.float64), small_pair["source"]["covs"], ctx=c2)
        m2 = api.GaussianVoxelMapGPU(0.5, ctx=c2).insert(t2)
        f2 = api.NonlinearFactorSetGPU(c2)
        f2.add(api.IntegratedVGICPFactorGPU(0, 1, m2, s2))
        time.sleep(0.01)
        outs = [f2.linearize({0: np.eye(4), 1: small_pair["delta"]})[0] for _ in range(10)]
        for o in outs[1:]:
            same(o, outs[0], "second context")
        served = api.resident_stats(ctx)
        assert served["launches"] > before["launches"] and served["requests"] >= before["requests"] + 60, (before, served)
        f2.close(); m2.close(); s2.close(); t2.close()
        c2.close()


def test_resident_requests_that_race_the_idle_out(api, ctx, orc, small_pair):
    """A request may arrive in the very moment the session's leader decides to idle out.  Calls spaced around a SHORT idle time-out (100 us; pauses
    of 0..400 us on top of the binding's own time per call) hit that moment many times in a few thousand calls: every call must return the
    launch-per-call bits and none may hang (the call whose request the leaving kernel did not take is answered by a restarted session, which finds
    the request in the lines), for one chip-wide factor and for a 34-factor set.  (Written for round 6's direct look of every block into the host's
    request lines -- refuted by measurement and removed, profiles/r06/probe/resident_direct_host_poll_refuted.json; the race it guards is older.)"""
    import time

    rng = np.random.default_rng(11)
    for rings, azimuths, copies, calls in ((64, 512, 1, 1500), (16, 256, 34, 600)):  # (the 34-factor set at the odometry's size: a plan of <= 2 x CUs rows)
        tgt, src, tg, sg, delta = _full_size_pair(api, ctx, rings, azimuths)
        vms = [api.GaussianVoxelMapGPU(r, ctx=ctx).insert(tg) for r in (0.5, 1.0)]
        factors = [api.IntegratedVGICPFactorGPU(0 if k < 4 else np.eye(4), 1 + k, vms[k % 2], sg) for k in range(copies)]
        poses = []
        for which in range(3):
            values = {0: np.eye(4)}
            for k in range(copies):
                values[1 + k] = delta @ orc.se3_exp(np.array([0.002, -0.001, 0.003, 0.02, 0.01, -0.02]) * (1 + 0.1 * k + which))
            poses.append(values)
        with ctx.diag("resident=0,fuse=0"):
            ref = api.NonlinearFactorSetGPU(ctx)
            for f in factors:
                ref.add(f)
            want = [ref.linearize(v) for v in poses]
            ref.close()
        for mode in ("resident=1,resident_idle_us=100",):
            with ctx.diag(mode):
                before = api.resident_stats(ctx)
                fset = api.NonlinearFactorSetGPU(ctx)
                for f in factors:
                    fset.add(f)
                t0 = time.perf_counter()
                for rep in range(calls):
                    got = fset.linearize(poses[rep % 3])
                    for k, (L, B) in enumerate(zip(got, want[rep % 3])):
                        assert L["num_inliers"] == B["num_inliers"], (mode, rep, k)
                        assert L["error"] == B["error"], (mode, rep, k)
                        np.testing.assert_array_equal(L["H_ss"], B["H_ss"], err_msg=f"{mode} rep {rep} factor {k}")
                        np.testing.assert_array_equal(L["b_s"], B["b_s"], err_msg=f"{mode} rep {rep} factor {k}")
                    pause = rng.uniform(0.0, 400e-6)
                    t1 = time.perf_counter()
                    while time.perf_counter() - t1 < pause:
                        pass
                elapsed = time.perf_counter() - t0
                after = api.resident_stats(ctx)
                fset.close()
                api.resident_stop(ctx)
                # the session was restarted many times (the race was exercised) and no call waited out a lost row (~1 s each)
                assert after["launches"] >= before["launches"] + 10, (mode, before, after)
                assert elapsed < 60.0, (mode, elapsed)


def test_a_stale_scratch_word_equal_to_the_next_build_number_does_not_end_the_wait(api, orc, small_pair):
    """the polled completion word of voxelmap_insert (word 2) and of glim_amd_frame_create (word 4 * (levels - 1) + 2) lives in the context's
    shared pinned scratch, where read_back_sync() leaves small integers (kNN counters, kept points).  A leftover equal to the NEXT build's sequence
    number used to end the host's wait before the kernel had written anything -- the voxel count then was the previous map's.  The word is cleared
    before every launch now: with the scratch poisoned the counts are the oracle's."""
    ctx = api.Context(0, 1)
    t, s = small_pair["target"], small_pair["source"]
    tg = api.PointCloudGPU.clone(t["points"].astype(np.float64), t["covs"], ctx=ctx)
    few = slice(0, 700)
    sg = api.PointCloudGPU.clone(s["points"][few].astype(np.float64), s["covs"][few], ctx=ctx)
    want_t = {r: orc.VoxelMap(r).insert(t["points"], t["covs"]).num_voxels() for r in (0.5, 1.0)}
    want_s = orc.VoxelMap(0.5).insert(s["points"][few], s["covs"][few]).num_voxels()
    assert want_s != want_t[0.5]
    first = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)  # leaves its own count in word 0
    assert first.voxelmap_info()["num_voxels"] == want_t[0.5]
    for _ in range(3):
        api.scratch_poke(ctx, 2)  # "the next build is complete"
        m = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(sg)
        assert m.voxelmap_info()["num_voxels"] == want_s
        api.scratch_poke(ctx, 2)
        m2 = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
        assert m2.voxelmap_info()["num_voxels"] == want_t[0.5]
    n = len(t["points"])
    p4 = np.ones((n, 4))
    p4[:, :3] = t["points"][:, :3]
    c = np.zeros((n, 4, 4))
    c[:, :3, :3] = t["covs"][:, :3, :3]
    c16 = np.ascontiguousarray(np.transpose(c, (0, 2, 1))).reshape(n, 16)
    n4 = np.zeros((n, 4))
    n4[:, :3] = t["normals"][:, :3]
    for switches in ("", "frame_fused=0"):
        ctx.set_diag(switches)
        api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(sg)  # (word 0 / 4 now hold other counts)
        api.scratch_poke(ctx, 6)
        cloud, maps = api.frame_create(p4, c16, n4, [0.5, 1.0], ctx=ctx)
        assert [m.voxelmap_info()["num_voxels"] for m in maps] == [want_t[0.5], want_t[1.0]]
    ctx.set_diag("")


def test_pre_cull_of_the_general_kernel_changes_no_bit(api, orc):
    """The pre-pass of large general-form sets (vgicp.hip cull_kernel) marks the wavefront trips whose 64-point chunk box, moved by the pose, touches no
    occupied cell of the target's occupancy mask; the factor kernel then walks the live trips only.  A marked trip would have added exact zeros,
    so linearise and error give the same bits with the pre-cull forced on (cull=2: sets of any size) and off -- for partly overlapping scans
    (something IS culled), for a pose that moves the source clean off the map (everything is), for every ppt, and against the oracle."""
    from glim_amd import synth

    ctx = api.Context(0, 1)
    scene = synth.Scene.default()
    dirs = synth.lidar_directions(32, 256)
    poses = synth.arc_trajectory(5)
    rng = np.random.default_rng(5)
    clouds, maps, refs = [], [], []
    for i, T in enumerate(poses):
        pts = synth.scan(scene, T, dirs, i)
        a = rng.normal(size=(len(pts), 3, 3)) * 0.05
        covs = np.zeros((len(pts), 4, 4))
        covs[:, :3, :3] = a @ np.transpose(a, (0, 2, 1)) + 1e-3 * np.eye(3)  # general (non-plane) covariances: the 36 B/pt kernel
        covs = covs.astype(np.float32).astype(np.float64)
        g = api.PointCloudGPU.clone(pts.astype(np.float64), covs, ctx=ctx)
        clouds.append((g, pts, covs))
        maps.append(api.GaussianVoxelMapGPU(1.0, ctx=ctx).insert(g))
        covs = covs[:, :3, :3]
        clouds[-1] = (g, pts, covs)
        refs.append(orc.VoxelMap(1.0).insert(pts, covs))
    pairs = [(i, j) for i in range(5) for j in range(5) if i != j]
    far = np.eye(4)
    far[:3, 3] = (500.0, -300.0, 40.0)
    for case, pose_of in enumerate((lambda i, j: synth.relative_pose(poses[i], poses[j]), lambda i, j: far if (i + j) % 2 else synth.relative_pose(poses[i], poses[j]) @ far)):
        deltas = np.stack([api.pose12(pose_of(i, j)) for i, j in pairs])
        got = {}
        for ppt in (1, 4, 64):
            for cull in (0, 2):
                ctx.set_diag("")
                ctx.set_diag(f"cull={cull},ppt={ppt},fuse=0")  # (fuse=0: a small synchronous set takes the two-dispatch form, whose factor launch is the culled one)
                fset = api.NonlinearFactorSetGPU(ctx)
                for i, j in pairs:
                    fset.add(api.IntegratedVGICPFactorGPU(i, j, maps[i], clouds[j][0]))
                fset.cull_stats(reset=True)  # (arm the counters)
                out = fset.linearize_poses(deltas)
                stats = fset.cull_stats()
                assert (stats is None) == (cull == 0)
                if cull:
                    assert 0 < stats[1] and 0 <= stats[0] <= stats[1]
                    got[("culled", ppt)] = stats[0] / stats[1]
                got[(cull, ppt)] = out
                fset.close()
            for a, b in zip(got[(0, ppt)], got[(2, ppt)]):
                assert a["num_inliers"] == b["num_inliers"] and a["error"] == b["error"]
                for k in ("H_ss", "b_s", "H_tt", "H_ts", "b_t"):
                    np.testing.assert_array_equal(a[k], b[k])
        if case == 1:
            assert got[("culled", 1)] > 0.9, got[("culled", 1)]  # a source moved 500 m off the map: the pre-pass proves (nearly) every chunk empty
    ctx.set_diag("")
    # ... and the culled evaluation is the oracle's (first pair of the overlapping case)
    deltas = np.stack([api.pose12(synth.relative_pose(poses[i], poses[j])) for i, j in pairs])
    ctx.set_diag("cull=2,fuse=0")
    fset = api.NonlinearFactorSetGPU(ctx)
    for i, j in pairs:
        fset.add(api.IntegratedVGICPFactorGPU(i, j, maps[i], clouds[j][0]))
    out = fset.linearize_poses(deltas)
    for k in (0, 7, 19):
        i, j = pairs[k]
        D = np.eye(4)
        D[:3, :4] = deltas[k].reshape(3, 4)
        ref = orc.vgicp_linearize(refs[i], clouds[j][1], clouds[j][2], D)
        assert out[k]["num_inliers"] == ref["num_inliers"]
        assert np.abs(gn(out[k]) - gn(ref)).max() < 1e-4
    ctx.set_diag("")
