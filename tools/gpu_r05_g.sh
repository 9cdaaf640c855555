#!/bin/bash
# the N > 1 code paths of bench.py on one GPU: BENCH_FORCE_DIST=1 initialises the RCCL process group with one rank and takes every collective branch
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05g
mkdir -p $OUT
cd $REPO
BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 400 python bench.py --workload global256 --steps 5 --warmup 2 > $OUT/bench_global256_forcedist.json 2> $OUT/bench_global256_forcedist.err < /dev/null
BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 timeout 400 python bench.py --workload odometry128k --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_odometry128k_forcedist.json 2> $OUT/bench_odometry128k_forcedist.err < /dev/null
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-m2 > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err < /dev/null
for f in $OUT/*.json; do echo "$f: $(head -c 300 $f)"; done
tail -5 $OUT/*.err | cut -c1-300
