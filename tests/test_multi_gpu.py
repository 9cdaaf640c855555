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
