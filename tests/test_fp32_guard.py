"""CPU check of the exactness guard of the factor kernels' OPT-IN FP32 point transform (glim_amd/csrc/vgicp.hip, locate_point; compiled only
with -DGLIM_AMD_K4_F32_TRANSFORM=1 -- the default build transforms in FP64).

The kernels compute q = R p + t, t32 = q / res and floor(t32) in FP32 and accept that voxel coordinate only when t32 lies at least
E = 2^-21 * (|px| + |py| + |pz| + max|t|) / res inside its unit interval on all three axes; every other lane falls back to FP64.  This test
restates the FP32 path in numpy (fmaf emulated through FP64, which is exact up to a double rounding of the sum) and checks the claim the
guard rests on: wherever the guard says "exact", the FP32 coordinate IS the FP64 coordinate of the oracle expression -- on uniformly random
points and on points placed 0.3 ... 5 E from a voxel face."""
import numpy as np
import pytest

f32 = np.float32


def fmaf(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def fp32_path(R, t, res, p):
    Rf, tf = R.astype(f32), t.astype(f32)
    ir = f32(1.0 / res)  # the kernel rounds the FP64 reciprocal the map stores
    q = [fmaf(np.full(len(p), Rf[r, 0]), p[:, 0], fmaf(np.full(len(p), Rf[r, 1]), p[:, 1], fmaf(np.full(len(p), Rf[r, 2]), p[:, 2], np.full(len(p), tf[r]))))
         for r in range(3)]
    u = [(q[r] * ir).astype(f32) for r in range(3)]
    fl = [np.floor(u[r]) for r in range(3)]
    d = [((u[r] - fl[r]).astype(f32) - f32(0.5)).astype(f32) for r in range(3)]
    ke = f32(f32(4.76837158203125e-7) * ir)
    ce = f32(f32(ke * np.abs(tf).max()) * f32(1.0000002))
    s = (np.abs(p[:, 0]) + np.abs(p[:, 1])).astype(f32)
    s = (s + np.abs(p[:, 2])).astype(f32)
    E = fmaf(s, np.full(len(p), ke), np.full(len(p), ce))
    m = np.maximum(np.abs(d[0]), np.maximum(np.abs(d[1]), np.abs(d[2])))
    exact = (m <= (f32(0.5) - E).astype(f32)) & (E <= f32(f32(6.103515625e-5) * ir))
    return np.stack(fl, axis=1).astype(np.int64), exact


def fp64_path(R, t, res, p):
    pd = p.astype(np.float64)
    ir = 1.0 / res
    # orc_transform_point: fma(R0, px, fma(R1, py, fma(R2, pz, t))) -- plain FP64 arithmetic differs from the fused form by < 2^-50 relative,
    # 25 binary orders below the guard's margin, so numpy's separate multiply-adds decide the same cell except within 1e-13 of a face
    q = pd @ R.T + t
    tq = q * ir
    return np.floor(tq).astype(np.int64), np.abs(tq - np.round(tq)).min(axis=1)


def se3(rng, rot, trans):
    w = rng.normal(size=3) * rot
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + (np.sin(th) / th) * K + ((1 - np.cos(th)) / th**2) * K @ K if th > 1e-12 else np.eye(3)
    return R, rng.normal(size=3) * trans


@pytest.mark.parametrize("res", [0.1, 0.25, 0.5, 1.0])
def test_guard_never_accepts_a_wrong_fp32_voxel_coordinate(res):
    rng = np.random.default_rng(int(res * 100))
    checked = accepted = 0
    for trial in range(6):
        R, t = se3(rng, 1.0, [5.0, 5.0, 1.0])
        n = 120000
        q = rng.uniform(-1.0, 1.0, size=(n, 3)) * [90.0, 90.0, 20.0]
        if trial % 2 == 1:  # adversarial: 0.3 ... 5 E from a face
            p0 = (q - t) @ R
            bound = 2.0 ** -21 * (np.abs(p0).sum(axis=1) + np.abs(t).max())
            ax = rng.integers(0, 3, size=n)
            q[np.arange(n), ax] = np.round(q[np.arange(n), ax] / res) * res + rng.choice([-1.0, 1.0], size=n) * rng.uniform(0.3, 5.0, size=n) * bound
        p = ((q - t) @ R).astype(f32)
        c32, exact = fp32_path(R, t, res, p)
        c64, face_dist = fp64_path(R, t, res, p)
        sure = face_dist > 1e-12  # numpy's unfused FP64 sum decides these cells like the oracle's fused one
        bad = exact & sure & (c32 != c64).any(axis=1)
        assert not bad.any(), f"{int(bad.sum())} accepted FP32 coordinates differ from FP64 (res {res}, trial {trial})"
        checked += int(sure.sum())
        accepted += int((exact & sure).sum())
    assert accepted > 0.5 * checked  # the guard is not vacuous: most points take the FP32 path


def test_guard_sends_far_maps_and_non_finite_points_to_fp64():
    rng = np.random.default_rng(3)
    R, _ = se3(rng, 0.5, [1.0, 1.0, 1.0])
    p = rng.uniform(-30, 30, size=(1000, 3)).astype(f32)
    _, exact = fp32_path(R, np.array([1.0e5, -7.5e4, 2.0e3]), 0.5, p)  # a map 100 km from the origin
    assert not exact.any()
    p[::7, 1] = np.nan
    p[3::7, 2] = np.inf
    with np.errstate(invalid="ignore"):
        _, exact = fp32_path(R, np.array([0.5, 0.25, -1.0]), 0.5, p)
    assert not exact[::7].any() and not exact[3::7].any()
    assert exact[1::7].mean() > 0.9
