#!/bin/bash
# Everything profiles/r06/ is built from, in one gpurun call (run from the repo root on the GPU box):   tools/round_evidence_r06.sh
#  1. the -m gpu test-suite                      2. tools/profile.sh: rocprofv3 kernel trace + stats, then the separate PMC passes, of the batched bench
#  3. PMC passes of the global256 workload (general kernel) and of the submap20 bundle       4. SQ / TCP / TCC counters of the batched plane-form kernel
#  5. traffic of the kNN query-group kernel for the rgbd300k / frontend128k lines            6. the default bench line and the other workloads
# The raw rocprofv3 output is summarised HERE and deleted: gpurun copies at most 64 MiB back.  Every rocprofv3 / bench step has its own hard time-out.
TAG=r06
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/evidence_$TAG
mkdir -p $OUT
cd $REPO
(timeout -k 5 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v '^$' | tail -15) > $OUT/gputest.log
timeout -k 5 600 bash $REPO/tools/profile.sh $TAG > $OUT/profile.log 2>&1 < /dev/null
cd $REPO
python tools/summarize_profile.py gpurun_out/prof_$TAG $OUT 128 > $OUT/summarize.log 2>&1
rm -rf gpurun_out/prof_$TAG
# the traffic files go where bench.py looks for them (this scratch copy of the repo) BEFORE the bench lines are taken, so that the lines
# carry traffic measured on exactly the kernels they time (`traffic_measured_on_this_kernel_version`)
mkdir -p profiles/$TAG && cp $OUT/traffic.json profiles/$TAG/traffic.json 2>/dev/null
SKIP_FETCH_PASS=1 BENCH_ARGS="--workload global256 --steps 3 --warmup 1 --no-cpu-baseline --no-predict --no-native" timeout -k 5 700 bash $REPO/tools/profile.sh ${TAG}_g > $OUT/profile_g.log 2>&1 < /dev/null
cd $REPO
mkdir -p $OUT/global256
PTS=$(python -c "import json;print(int(json.load(open('gpurun_out/prof_${TAG}_g/bench_stats.json'))['config']['mean_points_per_submap']*32640))" 2>/dev/null || echo 2180000000)
python tools/summarize_profile.py gpurun_out/prof_${TAG}_g $OUT/global256 32640 global256 36 $PTS >> $OUT/summarize.log 2>&1
cp $OUT/global256/traffic_global256.json profiles/$TAG/traffic_global256.json 2>/dev/null
rm -rf gpurun_out/prof_${TAG}_g
SKIP_FETCH_PASS=1 BENCH_ARGS="--workload submap20 --steps 10 --warmup 2 --no-cpu-baseline" timeout -k 5 400 bash $REPO/tools/profile.sh ${TAG}_s > $OUT/profile_s.log 2>&1 < /dev/null
cd $REPO
mkdir -p $OUT/submap20
python tools/summarize_profile.py gpurun_out/prof_${TAG}_s $OUT/submap20 380 submap20 24 24903680 >> $OUT/summarize.log 2>&1
cp $OUT/submap20/traffic_submap20.json profiles/$TAG/traffic_submap20.json 2>/dev/null
rm -rf gpurun_out/prof_${TAG}_s
# SQ / TCP / TCC counters of the batched plane-form kernel (one run per group: the SQ block holds few counters at a time)
cd /tmp && export TMPDIR=/tmp
M1ARGS="--steps 6 --warmup 2 --inner 16 --sync-calls 0 --no-cpu-baseline --no-m2 --no-resident-cost"
P=$REPO/gpurun_out/prof_${TAG}_c; mkdir -p $P
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $P/trace -- python $REPO/bench.py $M1ARGS > /dev/null 2> $P/trace.err
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $P/p1 -- python $REPO/bench.py $M1ARGS > /dev/null 2> $P/p1.err
timeout -k 5 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INST_CYCLES_VMEM --output-format csv -d $P/p2 -- python $REPO/bench.py $M1ARGS > /dev/null 2> $P/p2.err
timeout -k 5 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $P/p3 -- python $REPO/bench.py $M1ARGS > /dev/null 2> $P/p3.err
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_SALU --output-format csv -d $P/p4 -- python $REPO/bench.py $M1ARGS > /dev/null 2> $P/p4.err
cd $REPO
python tools/summarize_kernel_pmc.py $P "vgicp_kernel<0, false, true, false, false" $OUT/m1_counters.json workload=odometry128k_batched_128_factors >> $OUT/summarize.log 2>&1
tail -2 $P/p1.err $P/p3.err >> $OUT/summarize.log 2>&1
rm -rf $P
# traffic of the kNN query-group kernel for the two kNN-led lines
for w in rgbd300k frontend128k; do
  cd /tmp
  P=$REPO/gpurun_out/prof_${TAG}_k_$w; mkdir -p $P
  KARGS="--workload $w --frames 40 --no-cpu-baseline"
  timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $P/trace -- python $REPO/bench.py $KARGS > /dev/null 2> $P/trace.err
  timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/w -- python $REPO/bench.py $KARGS > /dev/null 2> $P/w.err
  timeout -k 5 200 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $P/r -- python $REPO/bench.py $KARGS > /dev/null 2> $P/r.err
  cd $REPO
  python tools/summarize_kernel_pmc.py $P knn_qgroup_kernel $OUT/traffic_knn_$w.json workload=$w >> $OUT/summarize.log 2>&1
  cp $OUT/traffic_knn_$w.json profiles/$TAG/ 2>/dev/null
  rm -rf $P
done
timeout -k 5 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null
for w in odometry_frame odometry_under_load submap20 rgbd300k frontend128k; do
  timeout -k 5 300 python bench.py --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err < /dev/null
done
# configs[3] through the native multi-device C-ABI path (world 1 here) + the N > 1 path over 8 virtual devices
timeout -k 5 500 python bench.py --gpus 1 --native > $OUT/bench_global256_native.json 2> $OUT/bench_global256_native.err < /dev/null
# the driver's N > 1 launch form with ONE rank (the only world this box has): must print one JSON line, rc 0
timeout -k 5 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_torchrun_world1.json 2> $OUT/bench_torchrun_world1.err < /dev/null; echo "torchrun world 1 rc $?" >> $OUT/summarize.log
BENCH_FRONTEND_FINE=1 timeout -k 5 200 python bench.py --workload frontend128k --no-cpu-baseline > /dev/null 2> $OUT/frontend_linearize_stage_fine.txt < /dev/null
timeout -k 5 100 python tools/voxelmap_time.py 2>&1 | grep -v '^[WE]20' > $OUT/voxelmap_time.txt
du -sh $REPO/gpurun_out
cat $OUT/gputest.log | tail -3
cat $OUT/summarize.log
cut -c1-400 $OUT/bench.json
