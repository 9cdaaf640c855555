#!/bin/bash
# GPU call 2 of round 3: full -m gpu suite, default bench, odometry_frame, then the FP32-transform A/B on global256 (+ parity tests with that library)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/c2
mkdir -p $OUT
cd $REPO
(timeout 700 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider 2>&1 | grep -v "^$" | tail -60) > $OUT/gputest.log
tail -25 $OUT/gputest.log
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null
timeout 200 python bench.py --workload odometry_frame > $OUT/bench_odometry_frame.json 2> $OUT/bench_odometry_frame.err < /dev/null
export GLIM_AMD_SCAN_CACHE=/tmp/glim_amd_scan_cache
for rep in 1 2; do
  for lib in main f32t; do
    if [ $lib = main ]; then unset GLIM_AMD_LIB; else export GLIM_AMD_LIB=$REPO/build/ab/$lib/libglim_amd.so; fi
    timeout 200 python bench.py --workload global256 --no-predict --steps 10 --warmup 3 > $OUT/g256_${lib}_$rep.json 2> $OUT/g256_${lib}_$rep.err < /dev/null
    timeout 200 python bench.py --no-m2 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/m1_${lib}_$rep.json 2> $OUT/m1_${lib}_$rep.err < /dev/null
  done
done
export GLIM_AMD_LIB=$REPO/build/ab/f32t/libglim_amd.so
(timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_edge_cases.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "not own_size" 2>&1 | tail -15) > $OUT/gputest_f32t.log
unset GLIM_AMD_LIB
tail -5 $OUT/gputest_f32t.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c2/g256_*.json')+glob.glob('gpurun_out/c2/m1_*.json')):
    try:
        r=json.load(open(f)); print(f, r['ms_per_step'], r['roofline']['kernel_ms'], r['config'].get('total_error'), r.get('parity'))
    except Exception as e: print(f,'ERR',e)
r=json.load(open('gpurun_out/c2/bench.json'))
print(json.dumps(r['single_factor_loop'])); print(r['value'], r['roofline']['kernel_ms'])
m=r['m2_global256']; print(m['ms_per_step'], json.dumps(m['parity']))
for k,v in m['predicted_scaling'].items():
    if k.startswith('pair_order'):
        for cm in ('cost_model_points','cost_model_fit'):
            w=v[cm] if cm=='cost_model_points' else v[cm].get('worlds',{})
            print(k,cm,{n:(round(x['compute_only_speedup_bound'],2),round(x['max_over_mean'],3)) for n,x in w.items()})
o=json.load(open('gpurun_out/c2/bench_odometry_frame.json'))['config']
for k in ('frames_10000_pts','frames_131072_pts'):
    print(k, json.dumps({a:b for a,b in o[k].items()}))
PY
