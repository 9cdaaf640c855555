#!/bin/bash
# Everything profiles/<tag>/ is built from, in one gpurun call (run from the repo root on the GPU box):   tools/round_evidence.sh r03
#  1. the -m gpu test-suite                      2. tools/profile.sh: rocprofv3 kernel trace + stats, then the separate PMC passes, of the default bench
#  3. PMC passes of the global256 workload (general kernel)
#  4. the default bench line (with cpu_baseline and m2_global256) and the other workloads; tools/batch_sweep.py
# The raw rocprofv3 output is summarised HERE (tools/summarize_profile.py) and deleted: gpurun copies at most 64 MiB back.
TAG=${1:-r05}
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
if [ -z "$SKIP_GLOBAL256_PMC" ]; then
  SKIP_FETCH_PASS=1 BENCH_ARGS="--workload global256 --steps 3 --warmup 1 --no-cpu-baseline --no-predict --no-native" timeout 600 bash $REPO/tools/profile.sh ${TAG}_g > $OUT/profile_g.log 2>&1 < /dev/null
  cd $REPO
  mkdir -p $OUT/global256
  python tools/summarize_profile.py gpurun_out/prof_${TAG}_g $OUT/global256 32640 global256 36 2029810471 >> $OUT/summarize.log 2>&1
  cp $OUT/global256/traffic_global256.json profiles/$TAG/traffic_global256.json 2>/dev/null
  # the kernel trace of the whole-frame pipelines (kNN, covariance, voxel map, preprocessing kernels): per-kernel durations only
  cd /tmp && export TMPDIR=/tmp
  PYTHONPATH=$REPO timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_${TAG}_frame -- python $REPO/tools/knn_time.py > $OUT/knn_time.txt 2>&1
  cd $REPO
  cp $(find gpurun_out/prof_${TAG}_frame -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_frame_pipeline.csv 2>/dev/null
  rm -rf gpurun_out/prof_${TAG}_frame
  rm -rf gpurun_out/prof_${TAG}_g
fi
timeout 250 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null
for w in odometry_frame submap20 global256 rgbd300k frontend128k odometry_under_load; do
  timeout 300 python bench.py --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err < /dev/null
done
# configs[3] through the native multi-device C-ABI path (world 1 here)
timeout 300 python bench.py --gpus 1 --native > $OUT/bench_global256_native.json 2> $OUT/bench_global256_native.err < /dev/null
# kernel breakdown of the scan preprocessing (rocprofv3 kernel trace of 45 calls on a raw 131 072-pt scan) and its wall time
python tools/preprocess_profile.py > $OUT/preprocess_time.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && PYTHONPATH=$REPO timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_${TAG}_pre -- python $REPO/tools/preprocess_profile.py > /dev/null 2>&1)
cp $(find gpurun_out/prof_${TAG}_pre -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_preprocess.csv 2>/dev/null
rm -rf gpurun_out/prof_${TAG}_pre
# the floor of a synchronous call on this box (launch / mailbox round trips) and the resident session's lifecycle
(hipcc -O3 --offload-arch=gfx950 tools/ubench/sync_floor.hip -o /tmp/sync_floor && timeout 120 /tmp/sync_floor) > $OUT/sync_floor.txt 2>&1
timeout 120 python tools/res_probe.py > $OUT/resident_probe.txt 2>&1
# kNN kernels against each other and the oracle (lists, wall time, per-wavefront counters of the query-group kernel); voxel-map insert wall time
mkdir -p $OUT/probe
timeout 300 python tools/knn_qgroup_probe.py 2>&1 | grep -v '^[WE]20' > $OUT/probe/knn_qgroup.txt
timeout 100 python tools/voxelmap_time.py 2>&1 | grep -v '^[WE]20' > $OUT/voxelmap_time.txt
timeout 200 python tools/batch_sweep.py > $OUT/batch_sweep.json 2> $OUT/batch_sweep.err < /dev/null
du -sh $REPO/gpurun_out
cat $OUT/gputest.log | tail -3
cut -c1-400 $OUT/bench.json
