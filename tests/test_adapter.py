"""Builds the GTSAM-facing adapter (adapters/gtsam/glim_amd_gtsam.hpp) against the stand-in GTSAM / Eigen / gtsam_points headers of
tests/cpp/mock/ (those libraries are not installed here) and runs tests/cpp/test_adapter.cpp on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_adapter.cpp")


def _build(tmp_path):
    from glim_amd import _lib
    from oracle import oracle

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    oracle.build()
    exe = str(tmp_path / "test_adapter")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "tests", "cpp", "mock"), "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "adapters", "gtsam"), SRC, "-o", exe, "-L" + os.path.join(ROOT, "glim_amd"), "-lglim_amd",
           "-L" + os.path.join(ROOT, "oracle"), "-lvgicp_oracle", "-Wl,-rpath," + os.path.join(ROOT, "glim_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    subprocess.check_call(cmd)
    return exe


def test_gtsam_adapter_compiles_against_the_stand_in_headers(tmp_path):
    _build(tmp_path)


@pytest.mark.gpu
def test_gtsam_adapter_matches_oracle_on_gpu(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "test_adapter OK" in out.stdout
