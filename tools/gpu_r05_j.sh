#!/bin/bash
# stage account of glim_amd_frame_create inside the live loop; the multi path with records stored to the host by the kernels
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05j
mkdir -p $OUT
cd $REPO
(timeout 400 python -m pytest tests/test_multi_gpu.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v '^$' | cut -c1-300 | tail -30) > $OUT/gputest.log
timeout 500 python bench.py --workload odometry_frame > $OUT/bench_odometry_frame.json 2> $OUT/bench_odometry_frame.err < /dev/null
timeout 400 python bench.py --workload global256 --no-cpu-baseline --no-predict > $OUT/bench_global256.json 2> $OUT/bench_global256.err < /dev/null
tail -5 $OUT/gputest.log
cut -c1-200 $OUT/bench_odometry_frame.json
for f in $OUT/*.err; do tail -n 3 $f | cut -c1-300; done
