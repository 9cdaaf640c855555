#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/c3
mkdir -p $OUT
cd $REPO
(timeout 700 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider 2>&1 | grep -v "^$" | tail -40) > $OUT/gputest.log
tail -12 $OUT/gputest.log
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null
timeout 200 python bench.py --workload odometry_frame > $OUT/bench_odometry_frame.json 2> $OUT/bench_odometry_frame.err < /dev/null
export GLIM_AMD_SCAN_CACHE=/tmp/glim_amd_scan_cache
for ppt in 0 32 64 128; do
  GLIM_AMD_DIAG="ppt=$ppt" timeout 200 python bench.py --workload global256 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/g256_ppt$ppt.json 2> $OUT/g256_ppt$ppt.err < /dev/null
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_knn -- python $REPO/tools/knn_time.py > $OUT/knn_time.txt 2>&1
cd $REPO
python - <<'PY'
import json,glob,csv,collections
for f in sorted(glob.glob('gpurun_out/c3/g256_*.json')):
    try:
        r=json.load(open(f)); ps=r.get('predicted_scaling') or {}
        tm=ps.get('pair_order_target_major',{}).get('cost_model_points',{})
        print(f, round(r['ms_per_step'],3), round(r['roofline']['kernel_ms'],3), {n:(round(x['compute_only_speedup_bound'],2),round(x['max_over_mean'],3)) for n,x in tm.items()})
    except Exception as e: print(f,'ERR',e)
r=json.load(open('gpurun_out/c3/bench.json'))
print(json.dumps(r['single_factor_loop'])); print(r['value'], r['roofline']['kernel_ms'], r['speedup_vs_cpu_baseline'])
o=json.load(open('gpurun_out/c3/bench_odometry_frame.json'))['config']
for k in ('frames_10000_pts','frames_131072_pts'):
    print(k, json.dumps(o[k]))
t=glob.glob('gpurun_out/c3/prof_knn/**/*kernel_trace.csv',recursive=True)
if t:
    acc=collections.defaultdict(list)
    for row in csv.DictReader(open(t[0])):
        acc[(row['Kernel_Name'][:90], row.get('Grid_Size'))].append((int(row['End_Timestamp'])-int(row['Start_Timestamp']))/1e3)
    for (k,g),v in sorted(acc.items(), key=lambda kv:-sum(kv[1]))[:40]:
        print(f"{sum(v):10.1f} us total {len(v):5d} calls avg {sum(v)/len(v):8.2f} min {min(v):8.2f} max {max(v):8.2f}  grid {g}  {k}")
PY
rm -rf $OUT/prof_knn
cat $OUT/knn_time.txt | grep -v "^\[" | tail -12
