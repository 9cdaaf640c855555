#!/bin/bash
# Profiling recipe used for profiles/ (run on the GPU box through gpurun): kernel trace + stats, then separate PMC passes.
set -x
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${1:-r01}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${BENCH_ARGS:---steps 10 --warmup 2 --no-cpu-baseline}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/bench_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $REPO/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $REPO/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/bench_write.err
find $OUT -name '*.csv' | head -50
