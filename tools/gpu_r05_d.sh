#!/bin/bash
# round 5, fourth GPU call: the -m gpu suite with durations, native path (8 pieces), live odometry frame loop
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05d
mkdir -p $OUT
cd $REPO
(timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 2>&1 | grep -v '^$' | tail -70) > $OUT/gputest.log
timeout 240 python bench.py --gpus 1 --native > $OUT/bench_native.json 2> $OUT/bench_native.err < /dev/null
timeout 400 python bench.py --workload odometry_frame > $OUT/bench_odometry_frame.json 2> $OUT/bench_odometry_frame.err < /dev/null
tail -5 $OUT/gputest.log
cut -c1-300 $OUT/bench_native.json
