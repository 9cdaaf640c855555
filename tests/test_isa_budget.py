"""Static performance guard (no GPU): the hot loop of the default VGICP kernel variant must stay within its register and instruction
budget -- 96 VGPRs (five waves per SIMD), no scratch spills, <= 245 instructions per point (DESIGN.md section 9 item 2 explains why the
instruction count is what bounds this kernel).  Uses tools/isa_stats.py (hipcc -S for gfx950)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_vgicp_hot_loop_stays_within_budget():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_stats.py"), os.path.join(ROOT, "glim_amd", "csrc", "vgicp.hip"),
                          "vgicp_kernelILi0ELb0ELb1ELb0E", "v_rcp_f32"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"vgpr (\d+) sgpr \d+ \| longest loop: (\d+) instructions", out.stdout)
    assert m, out.stdout
    vgpr, loop = int(m.group(1)), int(m.group(2))
    assert vgpr <= 96, f"{vgpr} VGPRs: the kernel would drop below five waves per SIMD"
    assert loop <= 245, f"{loop} instructions per point in the hot loop (was 240)"
    assert "scratch: False" in out.stdout, "register spills in the VGICP kernel"
