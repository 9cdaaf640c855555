"""On-disk map formats: graph.txt / data.txt text (CPU, C++ header) and the *_compact.bin cloud files (GPU)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_text_formats_roundtrip_and_known_bytes(tmp_path):
    from glim_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    exe = str(tmp_path / "test_io")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "test_io.cpp"), "-o", exe, "-L" + os.path.join(ROOT, "glim_amd"),
                           "-lglim_amd", "-Wl,-rpath," + os.path.join(ROOT, "glim_amd"), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    out_dir = tmp_path / "dump"
    out_dir.mkdir()
    out = subprocess.run([exe, str(out_dir)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "test_io OK" in out.stdout


@pytest.mark.gpu
def test_compact_cloud_files_roundtrip(orc, small_pair, tmp_path):
    """frame->save_compact(path) layout (sub_map.cpp:62): FP32 xyz + the six unique covariance entries; load gives the same device cloud."""
    from glim_amd import api

    ctx = api.Context(0, 1)
    s = small_pair["source"]
    g = api.PointCloudGPU.clone(s["points"], s["covs"], s["normals"], ctx=ctx)
    d = tmp_path / "000000"
    d.mkdir()
    g.save_compact(d)
    n = len(s["points"])
    pts = np.fromfile(d / "points_compact.bin", dtype=np.float32).reshape(n, 3)
    cov = np.fromfile(d / "covs_compact.bin", dtype=np.float32).reshape(n, 6)
    nrm = np.fromfile(d / "normals_compact.bin", dtype=np.float32).reshape(n, 3)
    np.testing.assert_array_equal(pts, s["points"].astype(np.float32))
    c32 = s["covs"].astype(np.float32)
    np.testing.assert_array_equal(cov, np.stack([c32[:, 0, 0], c32[:, 0, 1], c32[:, 0, 2], c32[:, 1, 1], c32[:, 1, 2], c32[:, 2, 2]], 1))
    np.testing.assert_array_equal(nrm, s["normals"].astype(np.float32))
    assert not (d / "times_compact.bin").exists()
    back = api.PointCloudGPU.load_compact(d, ctx=ctx)
    for a, b in zip(back.download(), g.download()):
        np.testing.assert_array_equal(a, b)
    # a preprocessed cloud also carries times / intensities; the full-precision pair points.bin / covs.bin loads as well
    rng = np.random.default_rng(0)
    pre = api.PointCloudGPU.preprocess(s["points"], np.sort(rng.uniform(0, 0.1, n)), rng.uniform(0, 255, n),
                                       api.preprocess_params(downsample_target=0, downsample_rate=1.0), ctx=ctx)
    d2 = tmp_path / "000001"
    d2.mkdir()
    pre.save_compact(d2)
    fr = pre.download_frame()
    np.testing.assert_array_equal(np.fromfile(d2 / "times_compact.bin", dtype=np.float32), fr["times"].astype(np.float32))
    np.testing.assert_array_equal(np.fromfile(d2 / "intensities_compact.bin", dtype=np.float32), fr["intensities"].astype(np.float32))
    d3 = tmp_path / "000002"
    d3.mkdir()
    p4 = np.ones((n, 4))
    p4[:, :3] = s["points"]
    c16 = np.zeros((n, 4, 4))
    c16[:, :3, :3] = s["covs"]
    p4.tofile(d3 / "points.bin")
    c16.tofile(d3 / "covs.bin")
    full = api.PointCloudGPU.load_compact(d3, ctx=ctx)
    xyz, c, _ = full.download(normals=False)
    np.testing.assert_array_equal(xyz, s["points"].astype(np.float32))
    np.testing.assert_array_equal(c, c32)
    with pytest.raises(api.GlimAmdError):
        api.PointCloudGPU.load_compact(tmp_path / "nothing_here", ctx=ctx)
