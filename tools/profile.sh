#!/bin/bash
# Profiling recipe used for profiles/ (run on the GPU box through gpurun): kernel trace + stats, then separate PMC passes.
set -x
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${1:-r01}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${BENCH_ARGS:---steps 10 --warmup 2 --inner 16 --sync-calls 0 --no-cpu-baseline --no-m2 --no-resident-cost}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/bench_stats.err
[ -z "$SKIP_FETCH_PASS" ] && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $REPO/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $REPO/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/bench_write.err
# read requests of the L2 to the fabric by size: the direct form of the gfx950 FETCH_SIZE correction (FETCH_SIZE tallies every request at 64 B)
rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc_rdreq -- python $REPO/bench.py $ARGS > $OUT/bench_rdreq.json 2> $OUT/bench_rdreq.err
find $OUT -name '*.csv' | head -50
