#!/bin/bash
# the driver's N > 1 launch form with one rank (torchrun, RCCL process group initialised, every collective branch of bench.py taken) on the shipped tree
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05t
mkdir -p $OUT
cd $REPO
BENCH_FORCE_DIST=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err < /dev/null
echo rc=$?
head -c 400 $OUT/bench_torchrun1.json; echo; tail -n 4 $OUT/bench_torchrun1.err | cut -c1-300
