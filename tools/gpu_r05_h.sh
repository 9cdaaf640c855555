#!/bin/bash
# piecewise small-cloud upload: the -m gpu suite and the odometry frame line again
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05h
mkdir -p $OUT
cd $REPO
(timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v '^$' | cut -c1-300 | tail -40) > $OUT/gputest.log
timeout 400 python bench.py --workload odometry_frame > $OUT/bench_odometry_frame.json 2> $OUT/bench_odometry_frame.err < /dev/null
timeout 300 python bench.py --workload frontend128k > $OUT/bench_frontend128k.json 2> $OUT/bench_frontend128k.err < /dev/null
tail -4 $OUT/gputest.log
cut -c1-200 $OUT/bench_odometry_frame.json
