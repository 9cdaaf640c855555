"""Turn the raw rocprofv3 output of tools/profile.sh (gpurun_out/prof_<tag>/) into the committed summaries under profiles/<tag>/:
summary.json (per kernel x grid: calls, avg / min / max duration from the kernel trace; FETCH_SIZE / WRITE_SIZE averages from
the separate PMC passes), kernel_stats.csv (rocprofv3's own --stats table) and traffic.json (HBM bytes per launch of the dominant
kernel: bytes from the L2 read-request size counters, which need no correction; the doubled FETCH_SIZE of MI355X_MICROARCH.md is recorded next to it).

usage: python tools/summarize_profile.py gpurun_out/prof_r01 profiles/r01 [factors_per_launch]"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_id  # noqa: E402  (identity of the factor kernel these counters were measured on; bench.py checks it)

src, dst = sys.argv[1], sys.argv[2]
F = int(sys.argv[3]) if len(sys.argv) > 3 else 128
# optional 4th / 5th / 6th argument: workload tag (traffic file suffix + the tag bench.py looks up), stream bytes per point, points per launch
TAG = sys.argv[4] if len(sys.argv) > 4 else "odometry128k"
STREAM_BPP = float(sys.argv[5]) if len(sys.argv) > 5 else 24.0
POINTS = float(sys.argv[6]) if len(sys.argv) > 6 else F * 131072.0
os.makedirs(dst, exist_ok=True)


def one(pattern):
    m = glob.glob(os.path.join(src, pattern), recursive=True)
    return m[0] if m else None


def by_kernel_grid(path, value_col, scale=1.0):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
        acc[(r["Kernel_Name"], grid)].append(float(r[value_col]) * scale)
    return acc


summary = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 10 --warmup 2 --inner 16 --sync-calls 0 --no-cpu-baseline --no-m2 --no-resident-cost   "
                      "(tools/profile.sh; PMC passes, each its own run: --pmc FETCH_SIZE; --pmc WRITE_SIZE; --pmc TCC_EA0_RDREQ_{32,64,128}B_sum)"}
trace = one("stats/**/*kernel_trace.csv")
rows = []
if trace:
    acc = defaultdict(list)
    for r in csv.DictReader(open(trace)):
        acc[(r["Kernel_Name"], int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        rows.append({"kernel": k, "grid": g, "calls": len(v), "avg_us": round(sum(v) / len(v), 3), "min_us": round(min(v), 3), "max_us": round(max(v), 3)})
summary["kernel_trace_by_grid"] = rows
for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    path = one(f"{tag}/**/*counter_collection.csv")
    out = []
    if path:
        acc = defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            acc[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
        for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            out.append({"kernel": k, "grid": g, "counter": counter, "calls": len(v), "avg": sum(v) / len(v)})
    summary[tag] = out
json.dump(summary, open(os.path.join(dst, "summary.json"), "w"), indent=1)
stats = one("stats/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(dst, "kernel_stats.csv"))
bench = os.path.join(src, "bench_stats.json")
if os.path.exists(bench) and os.path.getsize(bench):
    shutil.copy(bench, os.path.join(dst, "bench_under_rocprof.json"))

# traffic of the dominant kernel (largest total time among vgicp_kernel instantiations), per launch.
# MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B although the requests of a wide stream are 128 B -- double it.  The
# request-size counters give the same correction directly: bytes = 32 RDREQ_32B + 64 RDREQ_64B + 128 RDREQ_128B (separate PMC pass); both are
# recorded, the request counters are the figure used (they need no assumption about which part of the reads is "wide").
tj = os.path.join(dst, "traffic.json" if TAG == "odometry128k" else f"traffic_{TAG}.json")
dom = next((r for r in rows if "vgicp_kernel" in r["kernel"]), None)
out = {}
if dom:
    def avg_of(tag, counter):
        path = one(f"{tag}/**/*counter_collection.csv")
        if not path:
            return None
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
             if r["Counter_Name"] == counter and r["Kernel_Name"] == dom["kernel"] and int(r["Grid_Size"]) == dom["grid"]]
        return sum(v) / len(v) if v else None

    fetch_kb, write_kb = avg_of("pmc_fetch", "FETCH_SIZE"), avg_of("pmc_write", "WRITE_SIZE")
    r32, r64, r128 = (avg_of("pmc_rdreq", f"TCC_EA0_RDREQ_{b}B_sum") for b in (32, 64, 128))
    read = None
    if r128 is not None:
        read = 32.0 * (r32 or 0.0) + 64.0 * (r64 or 0.0) + 128.0 * r128
    elif fetch_kb is not None:
        read = 2.0 * fetch_kb * 1024.0
    wr = (write_kb or 0.0) * 1024.0
    if read is not None:
        out = {
            "workload": (f"odometry128k (bench.py default, F={F}), plane-form source clouds" if TAG == "odometry128k"
                         else f"{TAG} (bench.py --workload {TAG}, {F} factors per launch, {STREAM_BPP:.0f} B/pt stream)"),
            "kernel": dom["kernel"][dom["kernel"].find("vgicp_kernel"):].split("(")[0],
            "kernel_avg_us_rocprof": dom["avg_us"], "kernel_source_id": kernel_source_id(),
            "source": "rocprofv3 --pmc passes of tools/profile.sh (FETCH_SIZE; WRITE_SIZE; TCC_EA0_RDREQ_{32,64,128}B_sum), averages over the launches of the dominant kernel",
            "fetch_size_kb_raw": fetch_kb, "write_size_kb_raw": write_kb,
            "rdreq_32B": r32, "rdreq_64B": r64, "rdreq_128B": r128,
            "read_bytes_per_launch": read,
            "read_bytes_per_launch_from_2x_fetch_size": None if fetch_kb is None else 2.0 * fetch_kb * 1024.0,
            "write_bytes_per_launch": wr, "traffic_bytes_per_launch": read + wr,
            "factors_per_launch_when_measured": F, "traffic_bytes_per_factor": (read + wr) / F,
            "stream_bytes_per_launch": STREAM_BPP * POINTS, "gather_bytes_per_launch": read - STREAM_BPP * POINTS,
        }
        json.dump(out, open(tj, "w"), indent=1)
print(json.dumps({"kernels": len(rows), "dominant": dom and {k: dom[k] for k in ("grid", "calls", "avg_us")}, "traffic_per_factor": out.get("traffic_bytes_per_factor")}))
