#!/bin/bash
# front-end workloads: shipped switches against plan_recycle=0 on one box (the evidence box showed slow outliers in their linearise stage)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05m
mkdir -p $OUT
cd $REPO
for w in frontend128k rgbd300k; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_${w}.json 2> $OUT/bench_${w}.err < /dev/null
  GLIM_AMD_DIAG=plan_recycle=0 timeout 300 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_${w}_norecycle.json 2> $OUT/bench_${w}_norecycle.err < /dev/null
  timeout 300 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_${w}_again.json 2> $OUT/bench_${w}_again.err < /dev/null
done
for f in $OUT/*.json; do echo "$f: $(head -c 200 $f)"; done
