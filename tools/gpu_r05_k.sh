#!/bin/bash
# fused frame kernels (pull + every level's build in one launch, every level's records in one): the suites that touch clouds / maps / frames, the live frame
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05k
mkdir -p $OUT
cd $REPO
(timeout 600 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_parity.py tests/test_gpu_cpp.py tests/test_gpu_stress.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v '^$' | cut -c1-300 | tail -40) > $OUT/gputest.log
timeout 500 python bench.py --workload odometry_frame > $OUT/bench_odometry_frame.json 2> $OUT/bench_odometry_frame.err < /dev/null
tail -5 $OUT/gputest.log
cut -c1-200 $OUT/bench_odometry_frame.json
for f in $OUT/*.err; do tail -n 3 $f | cut -c1-300; done
