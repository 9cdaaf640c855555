#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/c6
mkdir -p $OUT
cd $REPO
export GLIM_AMD_SCAN_CACHE=/tmp/glim_amd_scan_cache
for d in "knn_heavy=0" "knn_heavy=100000" "knn_heavy=256" "knn_heavy=128" "knn_heavy=96" "knn_heavy=64" "knn_heavy=48"; do
  echo "== $d" >> $OUT/knn.txt
  GLIM_AMD_DIAG="$d" timeout 100 python tools/knn_time.py 2>&1 | grep "knn ms\|vmap" >> $OUT/knn.txt
done
cat $OUT/knn.txt
(timeout 700 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider 2>&1 | grep -v "^$" | tail -40) > $OUT/gputest.log
tail -8 $OUT/gputest.log
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null
timeout 200 python bench.py --workload odometry_frame > $OUT/bench_odometry_frame.json 2> $OUT/bench_odometry_frame.err < /dev/null
python - <<'PY'
import json
r=json.load(open('gpurun_out/c6/bench.json'))
print(json.dumps(r['single_factor_loop'])); print(r['value'], r['roofline']['kernel_ms'], r['speedup_vs_cpu_baseline'])
m=r['m2_global256']; print(m['ms_per_step'], m['roofline']['kernel_ms'], json.dumps(m['parity']))
print({k:{w:(round(x['compute_only_speedup_bound'],2), round(x['max_over_mean'],3)) for w,x in v['cost_model_points'].items()} for k,v in m['predicted_scaling'].items() if k.startswith('pair_order')})
o=json.load(open('gpurun_out/c6/bench_odometry_frame.json'))['config']
for k in ('frames_10000_pts','frames_131072_pts'):
    print(k, json.dumps(o[k]))
PY
