#!/bin/bash
# round 5, sixth GPU call: the -m gpu suite (polled voxel-map builds, spread overlap lanes), voxel-map timing, live odometry loop, default bench
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05f
mkdir -p $OUT
cd $REPO
(timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v '^$' | cut -c1-300 | tail -80) > $OUT/gputest.log
timeout 100 python tools/voxelmap_time.py 2>&1 | grep -v '^[WE]20' > $OUT/voxelmap_time.txt
timeout 400 python bench.py --workload odometry_frame > $OUT/bench_odometry_frame.json 2> $OUT/bench_odometry_frame.err < /dev/null
timeout 480 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null
tail -8 $OUT/gputest.log
cat $OUT/voxelmap_time.txt
cut -c1-300 $OUT/bench.json
