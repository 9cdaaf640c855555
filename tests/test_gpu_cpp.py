"""Builds and runs the C++ test of the gtsam_points-compatible host layer (tests/cpp/test_compat.cpp) on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_compat.cpp")


def _build(tmp_path):
    from glim_amd import _lib
    from oracle import oracle

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    oracle.build()
    exe = str(tmp_path / "test_compat")
    cmd = ["g++", "-std=c++17", "-O1", SRC, "-o", exe, "-L" + os.path.join(ROOT, "glim_amd"), "-lglim_amd", "-L" + os.path.join(ROOT, "oracle"),
           "-lvgicp_oracle", "-Wl,-rpath," + os.path.join(ROOT, "glim_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    subprocess.check_call(cmd)
    return exe


def test_cpp_mirror_compiles_without_a_gpu(tmp_path):
    """The header-only mirror + C ABI compile and link with plain g++ (no HIP headers needed by the host layer)."""
    _build(tmp_path)


@pytest.mark.gpu
def test_cpp_mirror_matches_oracle_on_gpu(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "test_compat OK" in out.stdout
