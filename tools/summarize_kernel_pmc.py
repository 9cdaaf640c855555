"""Per-kernel averages of rocprofv3 --pmc passes (tools/round_evidence_r06.sh) for ONE kernel picked by a name fragment: writes a small JSON.

  python tools/summarize_kernel_pmc.py <prof dir with pass sub-directories> <kernel name fragment> <out.json> [key=value ...]

Two uses: (a) traffic of the kNN query-group kernel for the kNN-led bench lines (bytes = 32 / 64 / 128 B read requests of the L2 to the fabric +
WRITE_SIZE, as tools/summarize_profile.py does for the factor kernel; MI355X_MICROARCH.md "HBM": FETCH_SIZE on gfx950 counts every request at 64 B, so
the request-size counters are used); (b) the SQ / TCP / TCC counters of the batched factor kernel ("co-limited" has to rest on
counters of THIS kernel version).  Every counter is averaged over the dispatches of the kernel with the most total time among the matches."""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

src, frag, dst = sys.argv[1], sys.argv[2], sys.argv[3]
extra = dict(a.split("=", 1) for a in sys.argv[4:])
counters = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values
for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if frag in r["Kernel_Name"]:
            counters[(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
durations = defaultdict(list)
for path in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if frag in r["Kernel_Name"]:
            g = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            durations[(r["Kernel_Name"], g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if not counters:
    print(json.dumps({"error": f"no dispatch of a kernel matching {frag!r} in {src}"}))
    sys.exit(1)
key = max(counters, key=lambda k: sum(durations.get(k, [0.0])) or len(next(iter(counters[k].values()))))
avg = {c: sum(v) / len(v) for c, v in counters[key].items()}
out = {"kernel": key[0][:200], "grid": key[1], "dispatches_per_counter": {c: len(v) for c, v in counters[key].items()}, "counters_avg_per_dispatch": avg,
       "kernel_avg_us_rocprof": (sum(durations[key]) / len(durations[key])) if durations.get(key) else None,
       "source": "rocprofv3 --pmc passes, one run per counter group (tools/round_evidence_r06.sh), averages over the dispatches of this kernel / grid"}
out.update(extra)
if "TCC_EA0_RDREQ_128B_sum" in avg:
    read = 32.0 * avg.get("TCC_EA0_RDREQ_32B_sum", 0.0) + 64.0 * avg.get("TCC_EA0_RDREQ_64B_sum", 0.0) + 128.0 * avg["TCC_EA0_RDREQ_128B_sum"]
    wr = avg.get("WRITE_SIZE", 0.0) * 1024.0
    out.update(read_bytes_per_launch=read, write_bytes_per_launch=wr, traffic_bytes_per_launch=read + wr)
if "knn" in frag:
    out["knn_source_id"] = bench.source_id(bench.KNN_SOURCES)
else:
    out["kernel_source_id"] = bench.kernel_source_id()
if "SQ_INSTS_VALU" in avg and "SQ_BUSY_CYCLES" in avg:
    # SQ_BUSY_CYCLES is summed over the shader engines' SQs; the per-SIMD issue rate below follows profiles/r03/probe/pmc_sq_global256.log's reading
    out["derived"] = {
        "valu_instructions_per_wave": avg["SQ_INSTS_VALU"] / avg["SQ_WAVES"] if avg.get("SQ_WAVES") else None,
        "share_of_wave_cycles_waiting_on_any_counter": avg["SQ_WAIT_INST_ANY"] / avg["SQ_WAVE_CYCLES"] if avg.get("SQ_WAIT_INST_ANY") and avg.get("SQ_WAVE_CYCLES") else None,
        "share_of_wave_cycles_issuing_vmem": avg["SQ_INST_CYCLES_VMEM"] / avg["SQ_WAVE_CYCLES"] if avg.get("SQ_INST_CYCLES_VMEM") and avg.get("SQ_WAVE_CYCLES") else None,
        "valu_busy_share_of_wave_cycles": avg["SQ_ACTIVE_INST_VALU"] / avg["SQ_WAVE_CYCLES"] if avg.get("SQ_ACTIVE_INST_VALU") and avg.get("SQ_WAVE_CYCLES") else None,
        "l2_hit_rate": avg["TCC_HIT_sum"] / (avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"]) if avg.get("TCC_HIT_sum") is not None and avg.get("TCC_MISS_sum") else None,
    }
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernel", "grid", "kernel_avg_us_rocprof") if k in out} | {"counters": sorted(avg)}))
