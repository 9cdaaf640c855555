cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
for v in pair w64; do
  if [ $v = w64 ]; then export GLIM_AMD_KNN_WAVE64=1; else unset GLIM_AMD_KNN_WAVE64; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/knnprof_$v -- python $R/tools/knn_time.py > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/knnpmc_$v -- python $R/tools/knn_time.py > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
for v in ("pair","w64"):
    for f in glob.glob("$R/gpurun_out/knnprof_%s/*/*kernel_stats.csv" % v):
        for r in csv.DictReader(open(f)):
            if "knn_" in r["Name"] or "half_box" in r["Name"] or "rs_" in r["Name"] or "curve" in r["Name"] or "bbox" in r["Name"]:
                print(v, r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
    for f in glob.glob("$R/gpurun_out/knnpmc_%s/*/*counter_collection.csv" % v):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "knn_pair" in r["Kernel_Name"] or "knn_chunk" in r["Kernel_Name"]:
                agg[(r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k2, vals in sorted(agg.items()): print(v, k2, len(vals), sum(vals)/len(vals))
PY
