#!/bin/bash
# where the front end's linearise stage spends its time, call by call (diagnostic)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05n
mkdir -p $OUT
cd $REPO
BENCH_FRONTEND_FINE=1 timeout 300 python bench.py --workload frontend128k --no-cpu-baseline > $OUT/fine.json 2> $OUT/fine.err < /dev/null
BENCH_FRONTEND_FINE=1 GLIM_AMD_DIAG=plan_cache=0 timeout 300 python bench.py --workload frontend128k --no-cpu-baseline > $OUT/fine_nocache.json 2> $OUT/fine_nocache.err < /dev/null
grep -h "linearise stage\|slowest\|release, us" $OUT/*.err | cut -c1-900
