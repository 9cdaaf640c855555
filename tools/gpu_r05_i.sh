#!/bin/bash
# gated small-cloud pull, plan recycling, one-device multi path without the per-evaluation library call: the -m gpu suite, the live frame, configs[3]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05i
mkdir -p $OUT
cd $REPO
(timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v '^$' | cut -c1-300 | tail -60) > $OUT/gputest.log
timeout 500 python bench.py --workload odometry_frame > $OUT/bench_odometry_frame.json 2> $OUT/bench_odometry_frame.err < /dev/null
timeout 400 python bench.py --workload global256 --no-cpu-baseline --no-predict > $OUT/bench_global256.json 2> $OUT/bench_global256.err < /dev/null
timeout 300 python bench.py --gpus 1 --native > $OUT/bench_global256_native.json 2> $OUT/bench_global256_native.err < /dev/null
tail -5 $OUT/gputest.log
cut -c1-200 $OUT/bench_odometry_frame.json
tail -3 $OUT/*.err | cut -c1-300
