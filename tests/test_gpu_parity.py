"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the CPU oracle on the
same seeded, f32-rounded inputs.

Gates (BASELINE.json north_star / SURVEY.md 8d):
  * correspondences bit-exact as (source i -> voxel integer coordinate | none); num_inliers equal
  * Gauss-Newton step delta = -(H + lambda I)^-1 b within 1e-4 m / 1e-4 rad of the oracle's, per iteration
  * kNN index sets exact; covariances within 1e-5 (FP32 storage)
"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-4  # metres / radians per Gauss-Newton iteration (north_star)


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


def upload_pair(api, ctx, pair):
    t, s = pair["target"], pair["source"]
    tg = api.PointCloudGPU.clone(t["points"].astype(np.float64), t["covs"], t["normals"], ctx=ctx)
    sg = api.PointCloudGPU.clone(s["points"].astype(np.float64), s["covs"], s["normals"], ctx=ctx)
    return tg, sg


def assert_linearization_close(got, ref, binary):
    assert got["num_inliers"] == ref["num_inliers"]
    np.testing.assert_allclose(got["error"], ref["error"], rtol=2e-4)
    scale = np.abs(ref["H_ss"]).max()
    np.testing.assert_allclose(got["H_ss"], ref["H_ss"], rtol=0, atol=2e-4 * scale)
    np.testing.assert_allclose(got["b_s"], ref["b_s"], rtol=0, atol=2e-4 * np.abs(ref["b_s"]).max() + 1e-6 * scale)
    if binary:
        np.testing.assert_allclose(got["H_tt"], ref["H_tt"], rtol=0, atol=2e-4 * np.abs(ref["H_tt"]).max())
        np.testing.assert_allclose(got["H_ts"], ref["H_ts"], rtol=0, atol=2e-4 * np.abs(ref["H_ts"]).max())
        np.testing.assert_allclose(got["b_t"], ref["b_t"], rtol=0, atol=2e-4 * np.abs(ref["b_t"]).max() + 1e-6 * scale)
    else:
        assert not np.any(got["H_tt"]) and not np.any(got["H_ts"]) and not np.any(got["b_t"])
    d_got, d_ref = gn_step(got), gn_step(ref)
    assert np.abs(d_got - d_ref).max() < POSE_TOL, (d_got, d_ref)
    lam = 1e-6 * np.trace(ref["H_ss"]) / 6
    assert np.abs(gn_step(got, lam) - gn_step(ref, lam)).max() < POSE_TOL


# ---------------------------------------------------------------------------------------------------------------


def test_device_is_gfx950(api, ctx):
    info = ctx.device_info()
    assert "gfx950" in info["name"], info
    assert info["num_cus"] >= 200


def test_cloud_roundtrip_f64_and_f32_layouts(api, ctx, small_pair):
    s = small_pair["source"]
    g64 = api.PointCloudGPU.clone(s["points"].astype(np.float64), s["covs"], s["normals"], ctx=ctx)
    g32 = api.PointCloudGPU.clone(s["points"], s["covs"].astype(np.float32), s["normals"].astype(np.float32), ctx=ctx)
    for g in (g64, g32):
        assert g.size() == len(s["points"])
        xyz, covs, nrm = g.download()
        np.testing.assert_array_equal(xyz, s["points"])
        np.testing.assert_array_equal(covs, s["covs"].astype(np.float32))
        np.testing.assert_array_equal(nrm, s["normals"].astype(np.float32))
    # points + covariances + normals, and -- a cloud of this size gets them from the upload kernel -- its factor stream (24 B per point in the
    # plane form, 36 + 16 otherwise)
    assert g64.memory_usage_gpu() in (len(s["points"]) * (16 + 24 + 16 + 24), len(s["points"]) * (16 + 24 + 16 + 36 + 16))


@pytest.mark.parametrize("n", [1, 255, 256, 257, 4095, 4096, 4097, 10000, 32768])
def test_gated_pull_equals_the_conversion_first_form(api, ctx, n):
    """The pull kernel of a small cloud is launched BEFORE the host converts the arrays and its blocks wait for their piece (pull_gated=1, the
    default); pull_gated=0 converts everything first.  Same arrays, same plane-form decision, whatever the piece boundaries."""
    rng = np.random.default_rng(n)
    pts = rng.uniform(-40, 40, (n, 3))
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm = nrm.astype(np.float32).astype(np.float64)
    covs = np.eye(3)[None] - 0.999 * nrm[:, :, None] * nrm[:, None, :]
    if n % 2 == 0:
        covs[n // 2] += 0.05 * np.eye(3)  # (one point that is not in plane form: the verdict must flip in both forms)
    got = {}
    for mode in ("pull_gated=1", "pull_gated=0"):
        with ctx.diag(mode):
            for rep in range(3):  # (staging blocks are recycled: an earlier upload's gate words must never open this one's)
                g = api.PointCloudGPU.clone(pts, covs, nrm, ctx=ctx)
                got[mode, rep] = (g.download(), g.memory_usage_gpu())
                g.close()
    for rep in range(3):
        (a, ma), (b, mb) = got["pull_gated=1", rep], got["pull_gated=0", rep]
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
        assert ma == mb
        np.testing.assert_array_equal(a[0], pts.astype(np.float32))


def test_small_cloud_host_pack_equals_the_device_pack(api, ctx, orc, small_pair):
    """Clouds of up to 32 768 points are converted to the device layout on the host and pulled over by one kernel (host_pack=1, the default);
    the general path uploads the FP64 arrays and packs on the device.  Same arrays, same plane-form decision (a factor over either cloud runs
    the same kernel and gives the same bits) -- for plane-form covariances and for general ones."""
    t, s = small_pair["target"], small_pair["source"]
    tg = api.PointCloudGPU.clone(t["points"].astype(np.float64), t["covs"], ctx=ctx)
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
    general = s["covs"][:, :3, :3] + 0.01 * np.eye(3)[None]
    for covs in (s["covs"], general):
        out = {}
        for mode in ("host_pack=1", "host_pack=0"):
            with ctx.diag(mode):
                g = api.PointCloudGPU.clone(s["points"].astype(np.float64), covs, s["normals"], ctx=ctx)
            fset = api.NonlinearFactorSetGPU(ctx)
            fset.add(api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, g))
            out[mode] = (g.download(), fset.linearize({1: small_pair["delta"]})[0])
        (a, La), (b, Lb) = out["host_pack=1"], out["host_pack=0"]
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
        assert La["num_inliers"] == Lb["num_inliers"] > 100 and La["error"] == Lb["error"]
        np.testing.assert_array_equal(La["H_ss"], Lb["H_ss"])
        np.testing.assert_array_equal(La["b_s"], Lb["b_s"])
    # without covariances / normals
    with ctx.diag("host_pack=1"):
        bare = api.PointCloudGPU.clone(s["points"].astype(np.float64), ctx=ctx)
    np.testing.assert_array_equal(bare.download(covs=False, normals=False)[0], s["points"])


def test_empty_cloud_and_empty_set(api, ctx, small_pair):
    e = api.PointCloudGPU.clone(np.zeros((0, 3)), np.zeros((0, 3, 3)), ctx=ctx)
    assert e.size() == 0
    fset = api.NonlinearFactorSetGPU(ctx)
    assert fset.linearize({}) == []
    t = small_pair["target"]
    tg = api.PointCloudGPU.clone(t["points"].astype(np.float64), t["covs"], ctx=ctx)
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
    f = api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, e)
    fset.add(f)
    L = fset.linearize({1: np.eye(4)})[0]
    assert L["num_inliers"] == 0 and L["error"] == 0.0 and not np.any(L["H_ss"]) and not np.any(L["b_s"])


@pytest.mark.parametrize("res", [0.25, 0.5, 1.0])
def test_voxelmap_matches_oracle(api, ctx, orc, small_pair, res):
    t = small_pair["target"]
    tg = api.PointCloudGPU.clone(t["points"].astype(np.float64), t["covs"], ctx=ctx)
    vm = api.GaussianVoxelMapGPU(res, ctx=ctx).insert(tg)
    ref = orc.VoxelMap(res).insert(t["points"], t["covs"])
    rc, rn, rm, rC = ref.voxels()
    info = vm.voxelmap_info()
    assert info["num_voxels"] == ref.num_voxels()
    assert info["num_buckets"] >= 2 * info["num_voxels"]  # lossless: load factor <= 1/2
    gc, gn, gm, gC = vm.voxels()
    # voxel identity = integer coordinate: identical SETS, identical member counts
    order_g = np.lexsort(gc.T[::-1])
    order_r = np.lexsort(rc.T[::-1])
    np.testing.assert_array_equal(gc[order_g], rc[order_r])
    np.testing.assert_array_equal(gn[order_g], rn[order_r])
    assert gn.sum() == len(t["points"])
    # statistics: mean of means / mean of covs, FP32 storage (<= 1 ulp_f32 of the FP64 oracle value)
    np.testing.assert_allclose(gm[order_g], rm[order_r], rtol=2e-7, atol=1e-7)
    np.testing.assert_allclose(gC[order_g], rC[order_r], rtol=2e-7, atol=1e-7)
    # bit-reproducible build (fixed-point accumulation): a second build is identical
    vm2 = api.GaussianVoxelMapGPU(res, ctx=ctx).insert(tg)
    c2, n2, m2, C2 = vm2.voxels()
    o2 = np.lexsort(c2.T[::-1])
    np.testing.assert_array_equal(m2[o2], gm[order_g])
    np.testing.assert_array_equal(C2[o2], gC[order_g])


@pytest.mark.parametrize("res", [0.25, 1.0])
def test_voxelmap_incremental_insert_matches_oracle(api, ctx, orc, small_pair, res):
    """A second and third insert() into the same map add to the voxels already there (GaussianVoxelMapCPU semantics: a finalised voxel is re-opened
    with mean *= n, cov *= n; odometry_estimation_cpu.cpp:66-67,189): coordinate sets and member counts equal the oracle's after every insert, the
    statistics agree to FP32 storage accuracy, a factor against the grown map matches the oracle's, and a live factor set follows the new table."""
    t, s = small_pair["target"], small_pair["source"]
    shifted = s["points"] + np.array([0.3, -0.2, 0.1], dtype=np.float32)  # overlapping clouds: old voxels grow, new ones appear
    parts = [(t["points"], t["covs"]), (shifted, s["covs"]), (t["points"][::3] + np.float32(7.0), t["covs"][::3])]
    vm = api.GaussianVoxelMapGPU(res, ctx=ctx)
    ref = orc.VoxelMap(res)
    sg = api.PointCloudGPU.clone(s["points"].astype(np.float64), s["covs"], ctx=ctx)
    fset = None
    for k, (p, c) in enumerate(parts):
        vm.insert(api.PointCloudGPU.clone(p.astype(np.float64), c, ctx=ctx))
        ref.insert(p, c)
        assert vm.voxelmap_info()["num_voxels"] == ref.num_voxels()
        gc, gn, gm, gC = vm.voxels()
        rc, rn, rm, rC = ref.voxels()
        og, orr = np.lexsort(gc.T[::-1]), np.lexsort(rc.T[::-1])
        np.testing.assert_array_equal(gc[og], rc[orr])
        np.testing.assert_array_equal(gn[og], rn[orr])
        assert gn.sum() == sum(len(q[0]) for q in parts[: k + 1])
        # the first insert is exact to FP32 storage; a re-opened voxel starts from its FP32 mean / covariance (what the map stores)
        np.testing.assert_allclose(gm[og], rm[orr], rtol=1e-6, atol=2e-7 * (k + 1))
        np.testing.assert_allclose(gC[og], rC[orr], rtol=1e-6, atol=2e-7 * (k + 1))
        if fset is None:
            fset = api.NonlinearFactorSetGPU(ctx)
            fset.add(api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, sg))
        got = fset.linearize({1: small_pair["delta"]})[0]  # the SAME set object: its plan must follow the rebuilt table
        want = orc.vgicp_linearize(ref, s["points"], s["covs"], small_pair["delta"])
        assert got["num_inliers"] == want["num_inliers"]
        assert np.abs(gn_step(got) - gn_step(want)).max() < POSE_TOL
    # an empty cloud changes nothing
    before = vm.voxels()
    vm.insert(api.PointCloudGPU.clone(np.zeros((0, 3)), np.zeros((0, 3, 3)), ctx=ctx))
    after = vm.voxels()
    o0, o1 = np.lexsort(before[0].T[::-1]), np.lexsort(after[0].T[::-1])
    np.testing.assert_array_equal(before[0][o0], after[0][o1])
    np.testing.assert_array_equal(before[1][o0], after[1][o1])


def test_voxelmap_negative_coordinates_and_range_error(api, ctx):
    pts = np.array([[-0.1, -0.1, -0.1], [-0.9, -0.2, -0.3], [0.1, 0.1, 0.1], [-1.0, 0.0, 0.0]], dtype=np.float64)
    covs = np.tile(np.eye(3), (4, 1, 1))
    g = api.PointCloudGPU.clone(pts, covs, ctx=ctx)
    vm = api.GaussianVoxelMapGPU(1.0, ctx=ctx).insert(g)
    coords, counts, means, _ = vm.voxels()
    got = {tuple(c): n for c, n in zip(coords, counts)}
    assert got == {(-1, -1, -1): 2, (0, 0, 0): 1, (-1, 0, 0): 1}
    far = api.PointCloudGPU.clone(np.array([[3e6, 0, 0]], dtype=np.float64), covs[:1], ctx=ctx)
    with pytest.raises(api.GlimAmdError) as ei:
        api.GaussianVoxelMapGPU(1.0, ctx=ctx).insert(far)
    assert ei.value.code == -4


@pytest.mark.parametrize("binary", [False, True])
@pytest.mark.parametrize("res", [0.5, 1.0])
def test_linearize_matches_oracle(api, ctx, orc, small_pair, binary, res):
    t, s = small_pair["target"], small_pair["source"]
    tg, sg = upload_pair(api, ctx, small_pair)
    vm = api.GaussianVoxelMapGPU(res, ctx=ctx).insert(tg)
    ref_vm = orc.VoxelMap(res).insert(t["points"], t["covs"])
    rng = np.random.default_rng(5)
    for trial in range(4):
        xi = rng.normal(size=6) * [0.01, 0.01, 0.01, 0.05, 0.05, 0.05] * (trial > 0)
        if binary:
            Tt = orc.se3_exp(rng.normal(size=6) * 0.3)
            Ts = Tt @ small_pair["delta"] @ orc.se3_exp(xi)
            factor = api.IntegratedVGICPFactorGPU(0, 1, vm, sg)
            values = {0: Tt, 1: Ts}
        else:
            Tt = orc.se3_exp(rng.normal(size=6) * 0.3)
            Ts = Tt @ small_pair["delta"] @ orc.se3_exp(xi)
            factor = api.IntegratedVGICPFactorGPU(Tt, 1, vm, sg)
            values = {1: Ts}
        delta = factor.calc_delta(values)
        fset = api.NonlinearFactorSetGPU(ctx)
        fset.add(factor)
        got = fset.linearize(values)[0]
        ref = orc.vgicp_linearize(ref_vm, s["points"], s["covs"], delta, want_corr=True)
        assert_linearization_close(got, ref, binary)
        # correspondences: bit-exact (i -> voxel coordinate | none)
        corr = fset.correspondences(0, delta)
        np.testing.assert_array_equal(corr[:, :3], ref["corr"][:, :3])
        np.testing.assert_array_equal(corr[:, 3] > 0, ref["corr"][:, 3] >= 0)
        assert factor.linearize(values) is got  # the factor serves the batch result (store_linearized)
        assert factor.inlier_fraction() == pytest.approx(ref["num_inliers"] / len(s["points"]))


def test_linearize_is_bit_reproducible(api, ctx, small_pair):
    tg, sg = upload_pair(api, ctx, small_pair)
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
    fset = api.NonlinearFactorSetGPU(ctx)
    fset.add(api.IntegratedVGICPFactorGPU(0, 1, vm, sg))
    a = fset.linearize_poses(api.pose12(small_pair["delta"])[None])[0]
    for _ in range(3):
        b = fset.linearize_poses(api.pose12(small_pair["delta"])[None])[0]
        np.testing.assert_array_equal(a["H_ss"], b["H_ss"])
        np.testing.assert_array_equal(a["b_s"], b["b_s"])
        assert a["error"] == b["error"]


def test_error_matches_oracle_both_semantics(api, ctx, orc, small_pair):
    t, s = small_pair["target"], small_pair["source"]
    tg, sg = upload_pair(api, ctx, small_pair)
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
    ref_vm = orc.VoxelMap(0.5).insert(t["points"], t["covs"])
    f = api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, sg)
    fset = api.NonlinearFactorSetGPU(ctx)
    fset.add(f)
    T_lin = small_pair["delta"]
    T_eval = T_lin @ orc.se3_exp([0.002, -0.001, 0.003, 0.02, -0.03, 0.01])
    # CPU-factor semantics: correspondences recomputed at the evaluation point
    e = fset.error({1: T_eval})[0]
    e_ref, n_ref = orc.vgicp_error(ref_vm, s["points"], s["covs"], T_eval)
    assert fset.last_error_inliers[0] == n_ref
    assert e == pytest.approx(e_ref, rel=2e-4)
    assert f.error({1: T_eval}) == pytest.approx(e_ref, rel=2e-4)
    # GPU-factor semantics: correspondences frozen at the linearisation point
    e2 = fset.error({1: T_eval}, values_lin={1: T_lin})[0]
    e2_ref, n2_ref = orc.vgicp_error(ref_vm, s["points"], s["covs"], T_eval, delta_lin=T_lin)
    assert fset.last_error_inliers[0] == n2_ref
    assert e2 == pytest.approx(e2_ref, rel=2e-4)


def test_batched_set_equals_individual_factors(api, ctx, orc, small_pair):
    """NonlinearFactorSetGPU with many factors (>= 16 switches on the XCD-aware block map) == per-factor results."""
    t, s = small_pair["target"], small_pair["source"]
    tg, sg = upload_pair(api, ctx, small_pair)
    vms = [api.GaussianVoxelMapGPU(r, ctx=ctx).insert(tg) for r in (0.5, 1.0)]
    refs = [orc.VoxelMap(r).insert(t["points"], t["covs"]) for r in (0.5, 1.0)]
    rng = np.random.default_rng(7)
    fset = api.NonlinearFactorSetGPU(ctx)
    values = {0: np.eye(4)}
    factors = []
    for k in range(20):
        values[k + 1] = small_pair["delta"] @ orc.se3_exp(rng.normal(size=6) * 0.01)
        f = api.IntegratedVGICPFactorGPU(0, k + 1, vms[k % 2], sg) if k % 3 else api.IntegratedVGICPFactorGPU(np.eye(4), k + 1, vms[k % 2], sg)
        if k % 5 == 0:
            f.set_enable_surface_validation(False)
        factors.append(f)
        fset.add(f)
    assert fset.size() == 20
    out = fset.linearize(values)
    for k, (f, got) in enumerate(zip(factors, out)):
        ref = orc.vgicp_linearize(refs[k % 2], s["points"], s["covs"], f.calc_delta(values))
        assert_linearization_close(got, ref, f.is_binary)
        single = api.NonlinearFactorSetGPU(ctx)
        single.add(f.clone())
        alone = single.linearize(values)[0]
        assert alone["num_inliers"] == got["num_inliers"]
        np.testing.assert_allclose(alone["H_ss"], got["H_ss"], rtol=1e-5)


def test_points_per_thread_variants_agree(api, ctx, small_pair):
    tg, sg = upload_pair(api, ctx, small_pair)
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
    T = api.pose12(small_pair["delta"])[None]
    results = []
    for ppt in ("1", "3", "8"):
        with ctx.diag(f"ppt={ppt}"):
            fset = api.NonlinearFactorSetGPU(ctx)
            fset.add(api.IntegratedVGICPFactorGPU(0, 1, vm, sg))
            results.append(fset.linearize_poses(T)[0])
    for r in results[1:]:
        assert r["num_inliers"] == results[0]["num_inliers"]
        np.testing.assert_allclose(r["H_ss"], results[0]["H_ss"], rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(r["b_s"], results[0]["b_s"], rtol=1e-4, atol=1e-3)


def test_surface_validation_rejects_back_facing_points(api, ctx, small_pair):
    t, s = small_pair["target"], small_pair["source"]
    tg, sg = upload_pair(api, ctx, small_pair)
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
    f = api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, sg)
    f.set_enable_surface_validation(True)
    fset = api.NonlinearFactorSetGPU(ctx)
    fset.add(f)
    delta = small_pair["delta"]
    got = fset.linearize({1: delta})[0]
    corr = fset.correspondences(0, delta)
    # predicate: reject when (R n) . q > 0, evaluated with the same FP32 expression on the host
    R = delta[:3, :3].astype(np.float32)
    q = (s["points"].astype(np.float64) @ delta[:3, :3].T + delta[:3, 3]).astype(np.float32)
    rn = s["normals"].astype(np.float32) @ R.T
    facing = np.einsum("ij,ij->i", rn, q) <= 0
    plain = api.NonlinearFactorSetGPU(ctx)
    plain.add(api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, sg))
    hits = plain.correspondences(0, delta)[:, 3] > 0
    margin = np.abs(np.einsum("ij,ij->i", rn.astype(np.float64), q.astype(np.float64))) > 1e-3
    np.testing.assert_array_equal((corr[:, 3] > 0)[margin], (hits & facing)[margin])
    assert got["num_inliers"] == int((corr[:, 3] > 0).sum())
    assert got["num_inliers"] <= int(hits.sum())


def test_overlap_matches_oracle(api, ctx, orc, small_pair):
    t, s = small_pair["target"], small_pair["source"]
    tg, sg = upload_pair(api, ctx, small_pair)
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
    vm1 = api.GaussianVoxelMapGPU(1.0, ctx=ctx).insert(tg)
    ref = orc.VoxelMap(0.5).insert(t["points"], t["covs"])
    ref1 = orc.VoxelMap(1.0).insert(t["points"], t["covs"])
    delta = small_pair["delta"]
    assert api.overlap_gpu(vm, sg, delta) == orc.overlap(ref, s["points"], delta)
    far = np.eye(4)
    far[:3, 3] = 1e4
    assert api.overlap_gpu(vm, sg, far) == 0.0
    shifted = delta @ orc.se3_exp([0, 0, 0.3, 2.0, 1.0, 0])
    assert api.overlap_gpu([vm, vm1], sg, [far, shifted]) == orc.overlap([ref, ref1], s["points"], [far, shifted])
    assert api.overlap_gpu([vm, vm1], sg, [delta, shifted]) == orc.overlap([ref, ref1], s["points"], [delta, shifted])


def test_knn_and_covariances_on_device_match_oracle(api, ctx, orc, small_pair):
    s = small_pair["source"]
    g = api.PointCloudGPU.clone(s["points"], ctx=ctx)
    nb = g.find_neighbors(10)
    ref_nb = s["neighbors"]
    assert np.all(nb[:, 0] == np.arange(len(nb)))
    np.testing.assert_array_equal(nb, ref_nb)  # exact, including the (distance, index) order
    g.estimate_covariances(10)
    _, covs, normals = g.download()
    rn, rc = orc.covariances(s["points"], ref_nb)
    # every point, no conditioning mask (tests/test_ref.py covariance_report: invariants everywhere, forward error explained by the gap)
    from test_ref import covariance_report

    frac, worst_gap = covariance_report(s["points"], ref_nb, 10, covs, normals, rc, rn)
    print(f"covariance parity, 4096-pt scan, k=10: fraction beyond 1e-5 = {frac:.2e}, largest relative gap among them = {worst_gap:.2e}")
    assert frac <= 2e-3 and worst_gap < 1e-6
    # fewer points than k: the tail is 0, like the reference's zero-initialised result vector (cloud_preprocessor.cpp:193, :200)
    tiny = api.PointCloudGPU.clone(np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], dtype=np.float32), ctx=ctx)
    np.testing.assert_array_equal(tiny.find_neighbors(5), orc.knn(np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], dtype=np.float32), 5))


@pytest.mark.parametrize("rings,azimuths,truth_tol", [(128, 128, None), (64, 256, 0.02)])
def test_gauss_newton_iterations_track_oracle(api, ctx, orc, rings, azimuths, truth_tol):
    """config 1 (plumbing): 16k-pt pair, 1.0 m voxels, unary factor, <= 8 iterations; per-iteration pose delta within 1e-4.  128 rings x 128
    azimuths is the geometry SURVEY 8d / BASELINE.md name for configs[0] (2.8 deg between azimuths: the kNN neighbourhoods of far walls straddle
    rings, the covariances are what the reference's estimator would give on such a scan -- both sides consume the same ones); 64 x 256 is the
    denser-azimuth variant rounds 1-5 ran.  What is asserted at both: the same inlier count and the same step every iteration, and that the pose
    iterated with the HIP path's own steps ends where the oracle's does.  Distance to the simulated truth is only asserted on the dense geometry:
    at 128 x 128 the undamped Gauss-Newton iteration of the ORACLE itself settles 0.114 m from the truth (cost 68.6e3 after the first step, 79.8e3
    at the fixed point -- the vertical-line neighbourhoods give the cost a biased minimum; measured on the CPU, round 6), so a truth bound there
    would test the scan pattern, not the kernels."""
    from glim_amd import synth

    scene = synth.Scene.default()
    dirs = synth.lidar_directions(rings, azimuths)  # 16 384 rays
    Tw = synth.pose(-8.0, -5.0, 1.8, 0.2)
    xi = np.array([0.01, -0.02, 0.015, 0.10, -0.05, 0.02])
    tgt = synth.scan(scene, Tw, dirs, 0)
    src = synth.scan(scene, Tw @ orc.se3_exp(xi), dirs, 1)
    tg = api.PointCloudGPU.clone(tgt, ctx=ctx)
    sg = api.PointCloudGPU.clone(src, ctx=ctx)
    for g in (tg, sg):
        g.find_neighbors(10, download=False)
        g.estimate_covariances(10)
    _, ct, _ = tg.download()
    _, cs, _ = sg.download()
    vm = api.GaussianVoxelMapGPU(1.0, ctx=ctx).insert(tg)
    ref_vm = orc.VoxelMap(1.0).insert(tgt, ct.astype(np.float64))
    f = api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, sg)
    fset = api.NonlinearFactorSetGPU(ctx)
    fset.add(f)
    T = np.eye(4)
    for it in range(8):
        got = fset.linearize({1: T})[0]
        ref = orc.vgicp_linearize(ref_vm, src, cs.astype(np.float64), T)
        assert got["num_inliers"] == ref["num_inliers"]
        lam = 1e-6 * np.trace(ref["H_ss"]) / 6
        d_got, d_ref = gn_step(got, lam), gn_step(ref, lam)
        assert np.abs(d_got - d_ref).max() < POSE_TOL, (it, d_got, d_ref)
        T_hip = T @ orc.se3_exp(d_got)
        T = T @ orc.se3_exp(d_ref)
        if np.linalg.norm(d_ref[3:]) < 1e-3 and np.linalg.norm(d_ref[:3]) < 1e-3 * np.pi / 180:
            break
    assert np.abs(np.linalg.inv(T) @ T_hip - np.eye(4)).max() < 2 * POSE_TOL
    if truth_tol is not None:
        err = np.linalg.inv(orc.se3_exp(xi)) @ T
        assert np.linalg.norm(err[:3, 3]) < truth_tol


def test_full_size_scan_properties(api, ctx, orc):
    """BASELINE config 2 size (131 072 points, 0.5 m voxels): inlier count + GN step vs the oracle, and size-independent
    properties: H symmetric PSD, identity-on-voxel-means gives zero gradient, linearity of H/b in a duplicated cloud."""
    from glim_amd import synth

    scene = synth.Scene.default()
    dirs = synth.lidar_directions(128, 1024)
    poses = synth.arc_trajectory(2)
    tgt = synth.scan(scene, poses[0], dirs, 0)
    src = synth.scan(scene, poses[1], dirs, 1)
    assert len(src) == 131072
    delta = synth.relative_pose(poses[0], poses[1])
    tg = api.PointCloudGPU.clone(tgt, ctx=ctx)
    sg = api.PointCloudGPU.clone(src, ctx=ctx)
    nbs = []
    for g in (tg, sg):
        nbs.append(g.find_neighbors(10))
        g.estimate_covariances(10)
    # exact kNN sets at full size vs the oracle's grid search
    ref_nb = orc.knn(src, 10)
    np.testing.assert_array_equal(np.sort(nbs[1], 1), np.sort(ref_nb, 1))
    _, ct, _ = tg.download()
    _, cs, _ = sg.download()
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
    ref_vm = orc.VoxelMap(0.5).insert(tgt, ct.astype(np.float64))
    assert vm.voxelmap_info()["num_voxels"] == ref_vm.num_voxels()
    fset = api.NonlinearFactorSetGPU(ctx)
    fset.add(api.IntegratedVGICPFactorGPU(0, 1, vm, sg))
    got = fset.linearize({0: np.eye(4), 1: delta})[0]
    ref = orc.vgicp_linearize(ref_vm, src, cs.astype(np.float64), delta, want_corr=True)
    assert_linearization_close(got, ref, True)
    corr = fset.correspondences(0, delta)
    np.testing.assert_array_equal(corr[:, :3], ref["corr"][:, :3])
    np.testing.assert_array_equal(corr[:, 3] > 0, ref["corr"][:, 3] >= 0)
    # properties
    np.testing.assert_allclose(got["H_ss"], got["H_ss"].T, rtol=0, atol=0)
    assert np.linalg.eigvalsh(got["H_ss"]).min() > 0
    full = np.block([[got["H_tt"], got["H_ts"]], [got["H_ts"].T, got["H_ss"]]])
    assert np.linalg.eigvalsh(full).min() > -1e-6 * np.abs(full).max()
    # duplicated source cloud -> exactly twice the inliers, H and b double
    dup = api.PointCloudGPU.clone(np.concatenate([src, src]), np.concatenate([cs, cs]), ctx=ctx)
    f2 = api.NonlinearFactorSetGPU(ctx)
    f2.add(api.IntegratedVGICPFactorGPU(0, 1, vm, dup))
    got2 = f2.linearize({0: np.eye(4), 1: delta})[0]
    assert got2["num_inliers"] == 2 * got["num_inliers"]
    # (the duplicate was uploaded with explicit FP32 covariances -> general kernel; the original is plane-form -> 24 B/pt kernel:
    #  the two evaluate C_A from differently rounded images of the same matrix, hence the 1e-5-level tolerance)
    # (FP32 block partials over a different partition of twice the points: ~sqrt(N) eps = 2-3e-5 of the largest entry)
    np.testing.assert_allclose(got2["H_ss"], 2 * got["H_ss"], rtol=0, atol=5e-5 * np.abs(got["H_ss"]).max())
    np.testing.assert_allclose(got2["b_s"], 2 * got["b_s"], rtol=0, atol=2e-4 * np.abs(got["b_s"]).max() + 1e-3)
    assert np.abs(gn_step(got2) - gn_step(got)).max() < 1e-6
    # voxel means against their own map at identity: zero residual
    coords, counts, means, covs = vm.voxels()
    mg = api.PointCloudGPU.clone(means, covs, ctx=ctx)
    f3 = api.NonlinearFactorSetGPU(ctx)
    f3.add(api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, mg))
    z = f3.linearize({1: np.eye(4)})[0]
    assert z["num_inliers"] >= len(means) - 5
    # (the downloaded means are rounded to FP32 at ~30 m, i.e. to ~2e-6 m, so the residual is tiny but not exactly zero)
    assert z["error"] < 1e-3
    assert np.abs(gn_step(z)).max() < 1e-5


def test_plane_form_kernel_matches_general_kernel_and_oracle(api, ctx, orc):
    """Clouds whose covariances were estimated on the device are plane-form (C = I - 0.999 n n^T) and take the 24 B/pt kernel;
    the same cloud through the general 36 B/pt kernel (diag plane=0) and the oracle must agree."""
    from glim_amd import synth

    scene = synth.Scene.default()
    dirs = synth.lidar_directions(64, 512)
    poses = synth.arc_trajectory(2)
    tgt, src = synth.scan(scene, poses[0], dirs, 0), synth.scan(scene, poses[1], dirs, 1)
    delta = synth.relative_pose(poses[0], poses[1]) @ orc.se3_exp([0.004, -0.003, 0.002, 0.03, 0.02, -0.01])
    tg, sg = api.PointCloudGPU.clone(tgt, ctx=ctx), api.PointCloudGPU.clone(src, ctx=ctx)
    for g in (tg, sg):
        g.find_neighbors(10, download=False)
        g.estimate_covariances(10)
    vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
    _, ct, _ = tg.download()
    _, cs, ns = sg.download()
    # the streamed normals reproduce the stored covariance: C = I - 0.999 n n^T within FP32 rounding
    np.testing.assert_allclose(cs, np.eye(3)[None] - 0.999 * ns[:, :, None] * ns[:, None, :], atol=3e-7)
    ref = orc.vgicp_linearize(orc.VoxelMap(0.5).insert(tgt, ct.astype(np.float64)), src, cs.astype(np.float64), delta, want_corr=True)
    results = {}
    for mode in ("plane", "general"):
        with ctx.diag("plane=1" if mode == "plane" else "plane=0"):
            for sv in (False, True):
                f = api.IntegratedVGICPFactorGPU(0, 1, vm, sg)
                f.set_enable_surface_validation(sv)
                fset = api.NonlinearFactorSetGPU(ctx)
                fset.add(f)
                results[(mode, sv)] = fset.linearize({0: np.eye(4), 1: delta})[0]
    for mode in ("plane", "general"):
        assert_linearization_close(results[(mode, False)], ref, True)
    # surface validation reads the normals from the plane-form stream in one kernel and from the normal array in the other
    assert results[("plane", True)]["num_inliers"] == results[("general", True)]["num_inliers"] <= ref["num_inliers"]
    assert np.abs(gn_step(results[("plane", True)]) - gn_step(results[("general", True)])).max() < 1e-5


def test_voxelmap_lru_horizon_matches_oracle_over_an_insert_sequence(api, ctx, orc):
    """GaussianVoxelMapCPU::set_lru_horizon as the CPU odometry uses it (odometry_estimation_cpu.cpp:63-68, update_target :177-191: one insert per frame
    into ONE incremental map): 14 inserts of a sensor moving down a corridor, horizon 3, clear cycle 2 -- after EVERY insert the device map holds
    exactly the oracle's voxels (coordinates, counts; means / covariances within FP32), a factor over it has the oracle's correspondences, and the
    map without a horizon keeps everything."""
    rng = np.random.default_rng(2)
    res = 0.5
    vm = api.GaussianVoxelMapGPU(res, ctx=ctx).set_lru_horizon(3, 2)
    keep = api.GaussianVoxelMapGPU(res, ctx=ctx)
    ref = orc.VoxelMap(res).set_lru_horizon(3, 2)
    evicted_any = False
    for step in range(14):
        pts = (rng.uniform([-1.0, -2.0, 0.0], [1.0, 2.0, 2.0], size=(400, 3)) + [1.5 * step, 0.0, 0.0]).astype(np.float32)
        covs = np.tile((np.eye(3) * 0.01).astype(np.float32), (len(pts), 1, 1))
        g = api.PointCloudGPU.clone(pts, covs, ctx=ctx)
        vm.insert(g)
        keep.insert(g)
        ref.insert(pts.astype(np.float64), covs.astype(np.float64))
        gc, gn, gm, gC = vm.voxels()
        rc, rn, rm, rC = ref.voxels()
        og, orr = np.lexsort(gc.T[::-1]), np.lexsort(rc.T[::-1])
        np.testing.assert_array_equal(gc[og], rc[orr], err_msg=f"insert {step}")
        np.testing.assert_array_equal(gn[og], rn[orr], err_msg=f"insert {step}")
        np.testing.assert_allclose(gm[og], rm[orr][:, :3], atol=2e-6)
        np.testing.assert_allclose(gC[og], rC[orr][:, :3, :3], atol=1e-7)
        evicted_any = evicted_any or vm.voxelmap_info()["num_voxels"] < keep.voxelmap_info()["num_voxels"]
    assert evicted_any and keep.voxelmap_info()["num_voxels"] > vm.voxelmap_info()["num_voxels"] == ref.num_voxels()
    # a factor over the evicting map: the dropped voxels give no correspondences any more
    src = (rng.uniform([-1.0, -2.0, 0.0], [1.0, 2.0, 2.0], size=(600, 3)) + [1.5 * 2, 0.0, 0.0]).astype(np.float32)   # where the sensor was 11 inserts ago
    sc = np.tile((np.eye(3) * 0.01).astype(np.float32), (len(src), 1, 1))
    sg = api.PointCloudGPU.clone(src, sc, ctx=ctx)
    fs = api.NonlinearFactorSetGPU(ctx)
    fs.add(api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, sg))
    fs.add(api.IntegratedVGICPFactorGPU(np.eye(4), 1, keep, sg))
    got = fs.linearize({1: np.eye(4)})
    want = orc.vgicp_linearize(ref, src.astype(np.float64), sc.astype(np.float64), np.eye(4))
    assert got[0]["num_inliers"] == want["num_inliers"] < got[1]["num_inliers"]


def test_plane_view_written_with_the_map_equals_the_one_built_on_first_use(api, ctx, orc, small_pair):
    """Plane-form factors read the PLANE VIEW of the target table (records hold (C_B + I)^-1, Sherman-Morrison in the kernel).  A map built from a
    plane-form cloud gets the view from its own finalise kernel (view_fused=1, default); any other map -- or view_fused=0 -- on the first factor
    that needs it.  Same table, same bits, by both routes and after an incremental insert; and the general kernel (plane=0: plain table,
    cofactor inverse) agrees with both within FP32 rounding."""
    t, s = small_pair["target"], small_pair["source"]
    tg = api.PointCloudGPU.clone(t["points"].astype(np.float32), ctx=ctx)
    sg = api.PointCloudGPU.clone(s["points"].astype(np.float32), ctx=ctx)
    for g in (tg, sg):
        g.find_neighbors(10, download=False)
        g.estimate_covariances(10)
    values = {0: np.eye(4), 1: small_pair["delta"]}

    def run(vm):
        fset = api.NonlinearFactorSetGPU(ctx)
        fset.add(api.IntegratedVGICPFactorGPU(0, 1, vm, sg))
        out = fset.linearize(values)[0], fset.error(values)[0]
        fset.close()
        return out

    fused = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
    with ctx.diag("view_fused=0"):
        lazy = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
    (La, ea), (Lb, eb) = run(fused), run(lazy)
    assert La["num_inliers"] == Lb["num_inliers"] > 100 and La["error"] == Lb["error"] and ea == eb
    for key in ("H_ss", "b_s", "H_tt", "H_ts", "b_t"):
        np.testing.assert_array_equal(La[key], Lb[key], err_msg=key)
    # incremental insert: the view follows the rebuilt table
    for vm in (fused, lazy):
        vm.insert(sg)
    with ctx.diag("view_fused=0"):
        both = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(tg)
        both.insert(sg)
    (Lc, _), (Ld, _), (Le, _) = run(fused), run(lazy), run(both)
    assert Lc["num_inliers"] == Ld["num_inliers"] == Le["num_inliers"] > La["num_inliers"]
    np.testing.assert_array_equal(Lc["H_ss"], Ld["H_ss"])
    np.testing.assert_array_equal(Lc["H_ss"], Le["H_ss"])
    with ctx.diag("plane=0"):
        Lg, eg = run(fused)
    assert Lg["num_inliers"] == Lc["num_inliers"]
    np.testing.assert_allclose(Lg["H_ss"], Lc["H_ss"], rtol=0, atol=2e-5 * np.abs(Lc["H_ss"]).max())
    assert np.abs(gn_step(Lg) - gn_step(Lc)).max() < 1e-6


def test_knn_chunk_and_grid_paths_agree(api, ctx, orc):
    """The two device kNN implementations (Hilbert-ordered chunks, hashed grid) and the oracle give identical lists, also on a cloud
    with a strongly non-uniform density, exact duplicates and a size that is not a multiple of the chunk length."""
    rng = np.random.default_rng(11)
    dense = rng.normal(size=(20000, 3)) * [0.05, 0.05, 0.02]
    sparse = rng.uniform(-30, 30, size=(9000, 3)) * [1, 1, 0.1]
    line = np.c_[np.linspace(0, 40, 2937), np.zeros(2937), np.zeros(2937)]
    pts = np.vstack([dense, sparse, line, dense[:500]]).astype(np.float32)  # the last 500 duplicate earlier points exactly
    ref = orc.knn(pts.astype(np.float64), 10)
    g = api.PointCloudGPU.clone(pts, ctx=ctx)
    got_chunk = g.find_neighbors(10)  # the default: the query-group kernel (knn_qgroup.hip), 2 queries per wavefront at this size
    for variant in ("wave64", "pair"):  # the lane-per-query kernels
        with ctx.diag(f"knn_kernel={variant}"):
            np.testing.assert_array_equal(g.find_neighbors(10), ref)
    with ctx.diag("knn_path=grid"):
        got_grid = g.find_neighbors(10)
    np.testing.assert_array_equal(got_chunk, ref)
    np.testing.assert_array_equal(got_grid, ref)
    for k in (1, 5, 16, 32):
        np.testing.assert_array_equal(g.find_neighbors(k), orc.knn(pts.astype(np.float64), k))
    # small clouds default to the grid path; the chunk path must agree there too (last chunk partially filled)
    small = api.PointCloudGPU.clone(pts[::7], ctx=ctx)
    ref_small = orc.knn(pts[::7].astype(np.float64), 10)
    np.testing.assert_array_equal(small.find_neighbors(10), ref_small)
    with ctx.diag("knn_path=chunks"):
        np.testing.assert_array_equal(small.find_neighbors(10), ref_small)


@pytest.mark.parametrize("name", ["lattice", "identical", "offset", "two_scales", "astronomic"])
def test_knn_exact_on_degenerate_distributions(api, ctx, orc, name):
    """Exact ties everywhere (lattice), all points identical, a cloud far from the origin, 1000x density contrast: both device paths.
    "astronomic": an extent beyond 1e18 m, where FP32 squared distances overflow -- the chunk path must hand over to the exhaustive FP64 kernel."""
    rng = np.random.default_rng(5)
    if name == "lattice":
        pts = np.stack(np.meshgrid(np.arange(20), np.arange(20), np.arange(15), indexing="ij"), -1).reshape(-1, 3) * 0.25
    elif name == "identical":
        pts = np.tile([[1.0, 2.0, 3.0]], (3000, 1))
    elif name == "offset":
        pts = rng.uniform(-1, 1, (6000, 3)) + [1e5, -2e5, 3e4]
    elif name == "astronomic":
        pts = rng.uniform(-1, 1, (3000, 3)) * 2e18
    else:
        pts = np.vstack([rng.normal(size=(4000, 3)) * 0.01, rng.uniform(-50, 50, (3000, 3))])
    pts = pts.astype(np.float32)
    ref = orc.knn(pts.astype(np.float64), 10, method="brute")
    g = api.PointCloudGPU.clone(pts, ctx=ctx)
    for variant in ("wave64", "pair", "qgroup"):  # 64 queries per wavefront / 32 queries with two lanes each / lanes = candidates
        with ctx.diag(f"knn_path=chunks,knn_kernel={variant}"):
            np.testing.assert_array_equal(g.find_neighbors(10), ref)
            for k in (3, 16, 32):
                np.testing.assert_array_equal(g.find_neighbors(k), orc.knn(pts.astype(np.float64), k, method="brute"))
    if name == "astronomic":
        return
    with ctx.diag("knn_path=grid"):
        np.testing.assert_array_equal(g.find_neighbors(10), ref)


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
def test_knn_refuses_non_finite_points_and_the_context_survives(api, ctx, orc, bad):
    """A non-finite coordinate is an error (GLIM_AMD_ERR_RANGE), found on the device: the bounding box never travels to the host, the chunk
    kernels stand down behind their guard word.  The next call on the same context answers as usual."""
    from glim_amd._lib import GlimAmdError

    rng = np.random.default_rng(11)
    pts = rng.uniform(-20, 20, (6000, 3)).astype(np.float32)
    ref = orc.knn(pts.astype(np.float64), 10, method="brute")
    broken = pts.copy()
    broken[1234, 1] = bad
    for variant in ("wave64", "pair", "qgroup"):
        with ctx.diag(f"knn_path=chunks,knn_kernel={variant}"):
            with pytest.raises(GlimAmdError) as err:
                api.PointCloudGPU.clone(broken, ctx=ctx).find_neighbors(10)
            assert err.value.code == -4
            np.testing.assert_array_equal(api.PointCloudGPU.clone(pts, ctx=ctx).find_neighbors(10), ref)


def test_knn_threshold_selection_is_exact(api, ctx, orc):
    """The per-lane threshold selection of the chunk kernels (knn_chunks.hip; the default for k <= 10) must leave every neighbour list
    bit-identical: both kernels (64 queries per wavefront / pair lanes), k = 10 and k = 5 (the two list sizes it is instantiated for), with the
    selection on (default) and off (diag knn_select=0), on ties, duplicates, a far offset, two scales and a real scan."""
    from glim_amd import synth

    rng = np.random.default_rng(5)
    clouds = {
        "lattice": np.stack(np.meshgrid(np.arange(20), np.arange(20), np.arange(15), indexing="ij"), -1).reshape(-1, 3) * 0.25,
        "identical": np.tile([[1.0, 2.0, 3.0]], (3000, 1)),
        "offset": rng.uniform(-1, 1, (6000, 3)) + [1e5, -2e5, 3e4],
        "two_scales": np.vstack([rng.normal(size=(4000, 3)) * 0.01, rng.uniform(-50, 50, (3000, 3))]),
        "duplicates": np.repeat(rng.uniform(-1, 1, (500, 3)), 9, axis=0),
        "scan": synth.scan(synth.Scene.default(), synth.arc_trajectory(1)[0], synth.lidar_directions(64, 512), 0)[:, :3],
    }
    for name, pts in clouds.items():
        pts = np.asarray(pts).astype(np.float32)
        g = api.PointCloudGPU.clone(pts, ctx=ctx)
        for k in (10, 5):
            ref = orc.knn(pts.astype(np.float64), k, method="brute")
            for select in (1, 0):
                for variant in ("wave64", "pair", "qgroup"):
                    with ctx.diag(f"knn_path=chunks,knn_kernel={variant},knn_select={select}"):
                        np.testing.assert_array_equal(g.find_neighbors(k), ref, err_msg=f"{name} k={k} {variant} select={select}")


def test_diag_switches_parse_and_reject(api, ctx):
    """glim_amd_ctx_set_diag: known keys change the context's switches, unknown keys / bad values change nothing; "" restores the defaults."""
    base = ctx.get_diag()
    assert base["knn_path"] == "auto" and base["plane"] == "1" and base["knn_select"] == "1" and base["resident"] == "auto"
    with ctx.diag("knn_path=grid,ppt=3"):
        assert ctx.get_diag()["knn_path"] == "grid" and ctx.get_diag()["ppt"] == "3"
        ctx.set_diag("plane=0")  # set_diag ADDS to the switches in force
        assert ctx.get_diag()["knn_path"] == "grid" and ctx.get_diag()["plane"] == "0"
        # pool / multi_rccl / multi_host_gather are process-wide (GLIM_AMD_DIAG only): a context refuses them instead of accepting a no-op
        for bad in ("no_such_key=1", "knn_path=fast", "ppt=-1", "plane", "pool=0", "plane=0,multi_rccl=0", "multi_host_gather=1", "resident=2"):
            with pytest.raises(api.GlimAmdError):
                ctx.set_diag(bad)
            assert ctx.get_diag()["knn_path"] == "grid"
    assert ctx.get_diag() == base  # the scope restores the process defaults
    with pytest.raises(RuntimeError):
        with ctx.diag("plane=0"):
            raise RuntimeError("a failure inside the scope")
    assert ctx.get_diag() == base
    assert ctx.get_diag() == base
