"""Static performance guard (no GPU): the hot loop of the default VGICP kernel variant must stay within its register and instruction
budget -- 96 VGPRs (five waves per SIMD), no scratch spills, and a bounded number of VECTOR-ALU instructions per point: with the gathers
pipelined the kernel is bound by VALU issue (DESIGN.md "K4"; scalar instructions issue beside the vector ones and are not counted).
Uses tools/isa_stats.py (hipcc -S for gfx950)."""
import ast
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VALU_BUDGET = 222  # 217 today: 123 FP32, 36 FP64, 43 integer, 15 compare / select (round 4, cofactor inverse: 225 = 133 + 36 + 40 + 16)


def test_vgicp_hot_loop_stays_within_budget():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_stats.py"), os.path.join(ROOT, "glim_amd", "csrc", "vgicp.hip"),
                          "vgicp_kernelILi0ELb0ELb1ELb0ELb0E", "v_rcp_f32"]  # linearise, plane-form, not inline, not the single-dispatch form
                         , capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"vgpr (\d+) sgpr \d+ \| longest loop: (\d+) instructions (\{.*\})", out.stdout)
    assert m, out.stdout
    vgpr, mix = int(m.group(1)), ast.literal_eval(m.group(3))
    valu = sum(v for k, v in mix.items() if k.startswith("valu"))
    assert vgpr <= 96, f"{vgpr} VGPRs: the kernel would drop below five waves per SIMD"
    assert valu <= VALU_BUDGET, f"{valu} vector-ALU instructions per point in the hot loop (budget {VALU_BUDGET}): {mix}"
    assert "scratch: False" in out.stdout, "register spills in the VGICP kernel"


def test_knn_query_group_kernel_stays_within_budget():
    """knn_qgroup.hip, k = 10, two queries per wavefront (the default kNN kernel from 49 152 points up): at most 64 VGPRs (eight wavefronts per SIMD: the kernel is a
    crowd of short independent work items and lives on occupancy), no scratch, no LDS permutes beyond the single lane ^ 32 stage of each sort
    (everything else is DPP / ds_swizzle: a ds_bpermute costs an LDS round trip inside a dependent chain), no exec-mask branches from
    short-circuit comparisons."""
    import collections
    import tempfile

    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "--cuda-device-only", "-S",
                               os.path.join(ROOT, "glim_amd", "csrc", "knn_qgroup.hip"), "-o", f.name], stderr=subprocess.DEVNULL, timeout=900)
        lines = open(f.name).read().splitlines()
    name = next(l.split()[1] for l in lines if l.strip().startswith(".amdhsa_kernel ") and "knn_qgroup_kernelILi10ELi2ELb0E" in l)
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i, l in enumerate(lines) if l.strip().startswith(".amdhsa_kernel " + name))
    body = [l.strip() for l in lines[start:end] if l.strip() and not l.strip().startswith((";", "."))]
    ops = collections.Counter(l.split()[0] for l in body)
    vgpr = int(next(l.split()[-1] for l in lines[end:end + 80] if ".amdhsa_next_free_vgpr" in l))
    assert vgpr <= 64, f"{vgpr} VGPRs: fewer than eight wavefronts per SIMD"
    assert not any("scratch_" in l for l in body), "register spills in the query-group kNN kernel"
    assert ops["ds_bpermute_b32"] <= 6, f"{ops['ds_bpermute_b32']} ds_bpermute in the kernel: a cross-lane move fell back to the LDS path"
    assert sum(1 for l in body if "dpp" in l) >= 150, "the bitonic network no longer compiles to DPP moves"
    assert ops["s_cbranch_execz"] <= 40, f"{ops['s_cbranch_execz']} exec-mask branches: a comparison compiles to control flow again"
