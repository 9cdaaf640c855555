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
VALU_BUDGET = 230  # 225 today: 133 FP32, 36 FP64, 40 integer, 16 compare / select


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
