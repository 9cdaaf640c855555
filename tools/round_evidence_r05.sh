#!/bin/bash
# Everything profiles/r05/ is built from, in one gpurun call (run from the repo root on the GPU box):   tools/round_evidence_r05.sh
#  1. the -m gpu test-suite                      2. tools/profile.sh: rocprofv3 kernel trace + stats, then the separate PMC passes, of the batched bench
#  3. PMC passes of the global256 workload (general kernel)        4. the default bench line and the other workloads
# The raw rocprofv3 output is summarised HERE (tools/summarize_profile.py) and deleted: gpurun copies at most 64 MiB back.
# (Unchanged since round 4 and not re-taken: kNN kernel probe, preprocessing kernel breakdown, sync_floor, batch sweep -- profiles/r04/.)
TAG=r05
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/evidence_$TAG
mkdir -p $OUT
cd $REPO
(timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v '^$' | tail -15) > $OUT/gputest.log
timeout 500 bash $REPO/tools/profile.sh $TAG > $OUT/profile.log 2>&1 < /dev/null
cd $REPO
python tools/summarize_profile.py gpurun_out/prof_$TAG $OUT 128 > $OUT/summarize.log 2>&1
rm -rf gpurun_out/prof_$TAG
# the traffic files go where bench.py looks for them (this scratch copy of the repo) BEFORE the bench lines are taken, so that the lines
# carry traffic measured on exactly the kernels they time (`traffic_measured_on_this_kernel_version`)
mkdir -p profiles/$TAG && cp $OUT/traffic.json profiles/$TAG/traffic.json 2>/dev/null
SKIP_FETCH_PASS=1 BENCH_ARGS="--workload global256 --steps 3 --warmup 1 --no-cpu-baseline --no-predict --no-native" timeout 600 bash $REPO/tools/profile.sh ${TAG}_g > $OUT/profile_g.log 2>&1 < /dev/null
cd $REPO
mkdir -p $OUT/global256
python tools/summarize_profile.py gpurun_out/prof_${TAG}_g $OUT/global256 32640 global256 36 2029810471 >> $OUT/summarize.log 2>&1
cp $OUT/global256/traffic_global256.json profiles/$TAG/traffic_global256.json 2>/dev/null
rm -rf gpurun_out/prof_${TAG}_g
timeout 480 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null
for w in odometry_frame odometry_under_load submap20 rgbd300k frontend128k; do
  timeout 300 python bench.py --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err < /dev/null
done
# configs[3] through the native multi-device C-ABI path (world 1 here)
timeout 300 python bench.py --gpus 1 --native > $OUT/bench_global256_native.json 2> $OUT/bench_global256_native.err < /dev/null
timeout 100 python tools/voxelmap_time.py 2>&1 | grep -v '^[WE]20' > $OUT/voxelmap_time.txt
du -sh $REPO/gpurun_out
cat $OUT/gputest.log | tail -3
cat $OUT/summarize.log
cut -c1-400 $OUT/bench.json
