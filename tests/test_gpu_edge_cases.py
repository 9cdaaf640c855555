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
    """GLIM_AMD_BUCKET_FACTOR=1 packs one key per two-way bucket on average: many keys spill to following buckets.  Lookups,
    map contents and the factor must stay exact (the probe compares full keys)."""
    monkeypatch.setenv("GLIM_AMD_BUCKET_FACTOR", factor)
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
    """GLIM_AMD_NO_POOL is read once per process; here we only check that repeated create/destroy cycles do not leak or crash."""
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
    """Host-supplied neighbour indices outside [0, n) and handles of another context are refused (GLIM_AMD_ERR_INVALID)."""
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
    with pytest.raises(api.GlimAmdError):
        api.overlap_gpu(vm, g2, np.eye(4), ctx=ctx)
    with pytest.raises(api.GlimAmdError):
        api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(g2)
    g2.close()
    other.close()
