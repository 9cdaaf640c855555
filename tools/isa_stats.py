#!/usr/bin/env python3
"""Static look at a kernel's hot loop (no GPU needed): compiles a .hip file to gfx950 assembly and prints, for every kernel whose
mangled name contains the given substring, the VGPR / SGPR counts and the instruction mix of its longest loop.
  python tools/isa_stats.py glim_amd/csrc/vgicp.hip vgicp_kernelILi0ELb0ELb1ELb0E v_rcp_f32 [extra hipcc flags]
Also prints the loop's floating-point operation count per trip (= per point per lane): fma / fmac / mad count 2, every other f32 / f64 arithmetic
instruction 1, conversions / moves / compares 0 -- bench.py prices the VALU-bound general kernel against the 157.3 TFLOP/s FP32 vector peak with it.
ISA_STATS_JSON=<file>: append one JSON line per kernel (name, registers, instruction mix, flops) to <file>.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile


def main():
    src, pat = sys.argv[1], sys.argv[2]
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "--cuda-device-only", "-S", *sys.argv[4:], src, "-o", f.name],
                              stderr=subprocess.DEVNULL)
        lines = open(f.name).read().splitlines()
    names = [l.split()[1] for l in lines if l.strip().startswith(".amdhsa_kernel ") and pat in l]
    for name in names:
        start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
        end = next(i for i, l in enumerate(lines) if l.strip().startswith(".amdhsa_kernel " + name))
        body = lines[start:end]
        meta = {k: next((l.split()[-1] for l in lines[end:end + 80] if k in l), "?") for k in (".amdhsa_next_free_vgpr", ".amdhsa_next_free_sgpr", ".amdhsa_accum_offset")}
        labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        marker = sys.argv[3] if len(sys.argv) > 3 else None  # e.g. v_rcp_f32: the smallest loop that contains this instruction
        cands = [lp for lp in loops if marker is None or any(marker in body[i] for i in range(lp[0], lp[1] + 1))]
        a, b = (min if marker else max)(cands, key=lambda x: x[1] - x[0])
        # instructions of nested loops that do not contain the marker (e.g. a rare re-probe loop) are not part of every trip: leave them out
        skip = set()
        for (c, d) in loops:
            if a <= c and d <= b and (c, d) != (a, b) and not (marker and any(marker in body[i] for i in range(c, d + 1))):
                skip.update(range(c, d + 1))
        ops = collections.Counter(l.split()[0] for i, l in ((i, body[i].strip()) for i in range(a, b + 1) if i not in skip) if l and not l.startswith((";", ".")))
        cat = collections.Counter()
        for op, c in ops.items():
            if op.startswith("v_pk_"):
                cat["valu_packed_f32"] += c
            elif op.startswith("v_") and "f64" in op:
                cat["valu_f64"] += c
            elif op.startswith(("v_cmp", "v_cndmask")):
                cat["valu_cmp_sel"] += c
            elif op.startswith("v_") and ("f32" in op or op.startswith(("v_fma", "v_mac"))):
                cat["valu_f32"] += c
            elif op.startswith("v_"):
                cat["valu_int_other"] += c
            elif op.startswith("s_waitcnt"):
                cat["waitcnt"] += c
            elif op.startswith("s_"):
                cat["salu"] += c
            elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
                cat["vmem"] += c
            elif op.startswith("ds_"):
                cat["lds"] += c
            else:
                cat["other"] += c
        def flops(kind):
            total = 0
            for op, c in ops.items():
                if not op.startswith("v_") or kind not in op or op.startswith(("v_cvt", "v_cmp", "v_cndmask", "v_mov")):
                    continue
                total += c * (2 if re.match(r"v_(pk_)?(fma|fmac|mac|mad)", op) else 1) * (2 if op.startswith("v_pk_") else 1)
            return total
        fl32, fl64 = flops("f32"), flops("f64")
        if os.environ.get("ISA_STATS_JSON"):
            import json
            with open(os.environ["ISA_STATS_JSON"], "a") as jf:
                sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
                from bench import kernel_source_id  # (stamps the factor kernel's source id: bench.py says whether a count belongs to the kernel it times)
                jf.write(json.dumps({"kernel": name, "kernel_source_id": kernel_source_id(), "vgpr": meta[".amdhsa_next_free_vgpr"], "sgpr": meta[".amdhsa_next_free_sgpr"], "loop_instructions": sum(ops.values()),
                                     "mix": dict(cat), "fp32_flops_per_point": fl32, "fp64_flops_per_point": fl64, "ops": dict(ops)}) + "\n")
        print(name[:90])
        print("  fp32 flops per trip:", fl32, "| fp64 flops per trip:", fl64)
        print("  vgpr", meta[".amdhsa_next_free_vgpr"], "sgpr", meta[".amdhsa_next_free_sgpr"], "| longest loop:", sum(ops.values()), "instructions", dict(cat))
        print("  scratch:", any("scratch_" in l for l in body), "| top:", ", ".join(f"{c} {o}" for o, c in ops.most_common(12)))


if __name__ == "__main__":
    main()
