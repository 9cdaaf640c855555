#!/bin/bash
# event pool of the voxel maps, completion word of glim_amd_frame_create from the last block of its build kernel: the -m gpu suite, the live frame line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05q
mkdir -p $OUT
cd $REPO
(timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v '^$' | cut -c1-300 | tail -15) > $OUT/gputest.log
timeout 500 python bench.py --workload odometry_frame > $OUT/bench_odometry_frame.json 2> $OUT/bench_odometry_frame.err < /dev/null
grep -h "passed\|failed" $OUT/gputest.log
cut -c1-200 $OUT/bench_odometry_frame.json
