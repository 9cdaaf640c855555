#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/c4
mkdir -p $OUT
cd $REPO
export GLIM_AMD_SCAN_CACHE=/tmp/glim_amd_scan_cache
for ppt in 32 16 8; do
  GLIM_AMD_DIAG="ppt=$ppt" timeout 200 python bench.py --workload global256 --no-cpu-baseline --no-predict --steps 10 --warmup 3 > $OUT/g256_ppt$ppt.json 2> $OUT/g256_ppt$ppt.err < /dev/null
done
for ppt in 0 32 16; do
  GLIM_AMD_DIAG="ppt=$ppt" timeout 200 python bench.py --no-m2 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/m1_ppt$ppt.json 2> $OUT/m1_ppt$ppt.err < /dev/null
  GLIM_AMD_DIAG="ppt=$ppt" timeout 200 python bench.py --workload submap20 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/s20_ppt$ppt.json 2> $OUT/s20_ppt$ppt.err < /dev/null
done
export PYTHONPATH=$REPO
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_knn -- python $REPO/tools/knn_time.py > $OUT/knn_time.txt 2>&1
cd $REPO
python - <<'PY'
import json,glob,csv,collections
for f in sorted(glob.glob('gpurun_out/c4/*_ppt*.json')):
    try:
        r=json.load(open(f)); print(f, round(r['ms_per_step'],3), round(r['roofline']['kernel_ms'],4), r['config'].get('lm_iteration_ms'))
    except Exception as e: print(f,'ERR',e)
t=glob.glob('gpurun_out/c4/prof_knn/**/*kernel_trace.csv',recursive=True)
if t:
    acc=collections.defaultdict(list)
    for row in csv.DictReader(open(t[0])):
        acc[(row['Kernel_Name'][:100], row.get('Grid_Size'))].append((int(row['End_Timestamp'])-int(row['Start_Timestamp']))/1e3)
    for (k,g),v in sorted(acc.items(), key=lambda kv:-sum(kv[1]))[:45]:
        print(f"{sum(v):10.1f} us total {len(v):5d} calls avg {sum(v)/len(v):8.2f} min {min(v):8.2f} max {max(v):8.2f}  grid {g}  {k}")
PY
rm -rf $OUT/prof_knn
grep -v "^\[\|^W2026\|^E2026" $OUT/knn_time.txt | tail -12
