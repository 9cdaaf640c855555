#!/bin/bash
# the multi path with the many-block error sum: its tests, the default bench line, the native path on its own
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05l
mkdir -p $OUT
cd $REPO
(timeout 400 python -m pytest tests/test_multi_gpu.py tests/test_gpu_stress.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v '^$' | cut -c1-300 | tail -30) > $OUT/gputest.log
timeout 480 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null
timeout 300 python bench.py --gpus 1 --native > $OUT/bench_global256_native.json 2> $OUT/bench_global256_native.err < /dev/null
tail -4 $OUT/gputest.log
cut -c1-300 $OUT/bench.json
for f in $OUT/*.err; do tail -n 3 $f | cut -c1-300; done
