#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/c5
mkdir -p $OUT
cd $REPO
export GLIM_AMD_SCAN_CACHE=/tmp/glim_amd_scan_cache
for d in "knn_heavy=0" "knn_heavy=8" "knn_heavy=12" "knn_heavy=16" "knn_heavy=24" "knn_heavy=32" "knn_heavy=48" "knn_heavy=0,knn_curve_bits=10" "knn_heavy=16,knn_curve_bits=10" "knn_heavy=16,knn_curve_bits=8"; do
  echo "== $d" >> $OUT/knn.txt
  GLIM_AMD_DIAG="$d" timeout 100 python tools/knn_time.py 2>&1 | grep "knn ms" >> $OUT/knn.txt
done
cat $OUT/knn.txt
timeout 200 python bench.py --no-m2 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/m1.json 2> $OUT/m1.err < /dev/null
timeout 200 python bench.py --workload submap20 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/s20.json 2> $OUT/s20.err < /dev/null
timeout 200 python bench.py --workload global256 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/g256.json 2> $OUT/g256.err < /dev/null
python - <<'PY'
import json
for n in ('m1','s20','g256'):
    try:
        r=json.load(open(f'gpurun_out/c5/{n}.json')); print(n, round(r['ms_per_step'],3), round(r['roofline']['kernel_ms'],4), r['config'].get('lm_iteration_ms'), r.get('single_factor_loop',{}).get('us_per_call'))
        ps=r.get('predicted_scaling')
        if ps: print({k:{w:(round(x['compute_only_speedup_bound'],2)) for w,x in v['cost_model_points'].items()} for k,v in ps.items() if k.startswith('pair_order')})
    except Exception as e: print(n,'ERR',e)
PY
