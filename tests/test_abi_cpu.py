"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol declared in
include/glim_amd.h; host-only entry points behave; no compute calls are made (there is no GPU here)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from glim_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported(L):
    from glim_amd import _lib

    declared = set()
    for h in ("glim_amd.h", "glim_amd_diag.h"):
        header = open(os.path.join(ROOT, "include", h)).read()
        found = set(re.findall(r"\b(glim_amd_[a-z0-9_]+)\s*\(", header))
        assert found, f"no declarations parsed in {h}"
        assert not (found & declared), f"declared in both headers: {found & declared}"
        declared |= found
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name


def _diag_symbols():
    header = open(os.path.join(ROOT, "include", "glim_amd_diag.h")).read()
    return set(re.findall(r"\b(glim_amd_[a-z0-9_]+)\s*\(", header))


def test_the_stable_header_holds_no_probe_and_the_drop_in_layers_never_touch_the_diag_header():
    """include/glim_amd.h is the drop-in boundary; include/glim_amd_diag.h holds every timing loop, counter, switch and debug view.  A GLIM
    maintainer binding the ABI must be able to tell them apart: no *_profile* / glim_amd_debug_* / *_trip_stats / set_diag name is declared in the
    stable header, and nothing under adapters/, examples/ or include/glim_amd/ includes the diag header or calls one of its entry points."""
    stable = open(os.path.join(ROOT, "include", "glim_amd.h")).read()
    names = set(re.findall(r"\b(glim_amd_[a-z0-9_]+)\s*\(", stable))
    for n in names:
        assert not re.search(r"profile|_debug_|trip_stats|_diag$|linearize_repeat|last_timing|last_breakdown|inject", n), n
    assert "#include \"glim_amd_diag.h\"" not in stable and "#include <glim_amd_diag.h>" not in stable  # (named in a comment only)
    diag = _diag_symbols()
    assert len(diag) >= 20
    for top in ("adapters", "examples", os.path.join("include", "glim_amd")):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if not f.endswith((".h", ".hpp", ".c", ".cpp", ".cc")):
                    continue
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "glim_amd_diag.h" not in text, os.path.join(dirpath, f)
                used = {n for n in re.findall(r"\b(glim_amd_[a-z0-9_]+)\s*\(", text)} & diag
                assert not used, (os.path.join(dirpath, f), used)


def test_version_and_error_strings(L):
    assert L.glim_amd_version() >= 100
    assert L.glim_amd_error_string(0) == b"ok"
    for code in range(-7, 0):
        assert L.glim_amd_error_string(code) not in (b"", b"unknown error")
    assert L.glim_amd_error_string(-99) == b"unknown error"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from glim_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _lib.lib()


def test_no_device_is_an_error_code_not_a_fallback(L):
    from glim_amd import api

    if api.device_count() > 0:
        pytest.skip("a GPU is visible")
    h = C.c_void_p()
    assert L.glim_amd_ctx_create(0, 1, None, C.byref(h)) == -3  # GLIM_AMD_ERR_NO_DEVICE
    with pytest.raises(api.GlimAmdError):
        api.Context(0, 1)


def test_invalid_arguments_return_codes(L):
    assert L.glim_amd_ctx_create(0, 1, None, None) == -1
    assert L.glim_amd_ctx_synchronize(None) == -1
    n = C.c_int64()
    assert L.glim_amd_cloud_size(None, C.byref(n)) == -1
    assert L.glim_amd_cloud_destroy(None) == 0
    assert L.glim_amd_voxelmap_destroy(None) == 0
    assert L.glim_amd_factor_set_destroy(None) == 0


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py may touch oracle/."""
    pkg = os.path.join(ROOT, "glim_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                code = "\n".join(line for line in text.splitlines() if not line.strip().startswith(("//", "#", "*", "/*")))
                assert "orc_" not in code, f
                assert "import oracle" not in code and "from oracle" not in code and "libvgicp_oracle" not in code, f


def test_expand_compact_matches_oracle_binary_blocks(orc, small_pair):
    """Host-side FP64 expansion of the compact 6x6 source block into the binary factor's H_tt / H_ts / b_t (adjoint identity)."""
    from glim_amd import api

    t, s = small_pair["target"], small_pair["source"]
    vm = orc.VoxelMap(0.5).insert(t["points"], t["covs"])
    T = small_pair["delta"] @ orc.se3_exp([0.01, -0.02, 0.005, 0.05, 0.02, -0.01])
    ref = orc.vgicp_linearize(vm, s["points"], s["covs"], T)
    iu = np.triu_indices(6)
    compact = np.concatenate([[ref["num_inliers"], ref["error"]], ref["H_ss"][iu], ref["b_s"]])
    got = api.expand_compact(compact, T, api.FACTOR_BINARY)
    assert got["num_inliers"] == ref["num_inliers"]
    for k in ("H_tt", "H_ts", "H_ss"):
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-6 * np.abs(ref[k]).max())
    np.testing.assert_allclose(got["b_t"], ref["b_t"], rtol=1e-9, atol=1e-9 * np.abs(ref["b_t"]).max())
    np.testing.assert_allclose(got["b_s"], ref["b_s"], rtol=0, atol=0)
    un = api.expand_compact(compact, T, 0)
    assert not np.any(un["H_tt"]) and not np.any(un["H_ts"]) and not np.any(un["b_t"])


def test_header_is_plain_c99_and_the_c_example_links(tmp_path):
    """include/glim_amd.h must be consumable from C (the FFI of any host language binds it): the pure-C per-frame example compiles with
    -std=c99 -pedantic -Werror, links against the shared library and runs its no-device branch here."""
    import subprocess

    from glim_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    exe = str(tmp_path / "c_abi_frame")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_frame.c"), "-o", exe, "-L" + os.path.join(ROOT, "glim_amd"), "-lglim_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "glim_amd"), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "glim_amd ABI version" in out.stdout


def test_committed_traffic_files_were_measured_on_the_current_factor_kernel():
    """bench.py quotes `roofline.traffic` from the committed PMC passes (the counters need rocprofv3 runs of their own); the files carry the
    hash of the factor kernel's sources, and the newest one per workload must be the hash of the sources in the tree -- editing vgicp.hip /
    device_math.hpp / internal.hpp / the Makefile without re-running tools/round_evidence.sh fails here, not silently on the bench line."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for tag in ("odometry128k", "global256"):
        got = bench.measured_traffic(tag)
        assert got is not None, tag
        assert got[2], f"{got[1]} was measured on another version of the factor kernel ({tag})"


def test_committed_rocprof_average_agrees_with_the_bench_line_of_the_same_run():
    """Evidence hygiene (bench contract: the rocprofv3 --kernel-trace average of the dominant kernel must agree with the live HIP-event figure):
    the newest profiles/*/summary.json row of the batched factor kernel and the `roofline.kernel_ms` of the bench line taken INSIDE that profiler
    run (bench_under_rocprof.json) are within 5 % of each other, and the traffic file beside them is the one summarised from the same run."""
    import glob

    dirs = sorted(d for d in glob.glob(os.path.join(ROOT, "profiles", "r*")) if os.path.exists(os.path.join(d, "summary.json")))
    assert dirs
    d = dirs[-1]
    summary = json.load(open(os.path.join(d, "summary.json")))
    line = json.load(open(os.path.join(d, "bench_under_rocprof.json")))
    traffic = json.load(open(os.path.join(d, "traffic.json")))
    row = summary["kernel_trace_by_grid"][0]  # rows are ordered by total time: the dominant kernel first
    assert "vgicp_kernel" in row["kernel"] and row["calls"] >= 100
    live_us = line["roofline"]["kernel_ms"] * 1e3
    assert abs(row["avg_us"] - live_us) <= 0.05 * live_us, (row["avg_us"], live_us)
    assert abs(traffic["kernel_avg_us_rocprof"] - row["avg_us"]) < 1e-6


def test_every_diagnostic_key_is_documented():
    """Doc-drift guard: every key glim_amd_ctx_set_diag accepts (context.hip kDiagKeys + knn_debug) is named in the header comment of
    glim_amd_ctx_set_diag and in DESIGN.md 4.9, and the value words of the enumerated keys are listed there too."""
    src = open(os.path.join(ROOT, "glim_amd", "csrc", "context.hip")).read()
    keys = re.findall(r'\{"([a-z_0-9]+)", &Diag::', src) + ["knn_debug"]
    assert len(keys) >= 20
    header = open(os.path.join(ROOT, "include", "glim_amd_diag.h")).read()
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    for k in keys:
        assert k + "=" in header, f"diag key {k} is not documented in include/glim_amd_diag.h"
        assert k + "=" in design or f"`{k}`" in design, f"diag key {k} is not documented in DESIGN.md"
    for words in re.findall(r'k(?:Path|Kernel)Words\[\] = \{([^}]*)\}', src):
        for w in re.findall(r'"([a-z0-9]+)"', words):
            assert w in header and w in design, w


def test_design_md_is_the_template_filled_from_the_committed_evidence():
    """DESIGN.md is generated: tools/design/DESIGN.tpl.md with every number read from profiles/r05/ (tools/design/fill_design.py).  A number edited
    by hand in DESIGN.md, or evidence refreshed without refilling it, fails here."""
    import subprocess
    import sys

    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design", "fill_design.py"), "--check"], cwd=ROOT, capture_output=True, text=True)
    assert res.returncode == 0, "DESIGN.md differs from the filled template: run `python tools/design/fill_design.py`\n" + res.stderr[-500:]
