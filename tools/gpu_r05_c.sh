#!/bin/bash
# round 5, third GPU call: the -m gpu suite (plane view, twins, multi pieces), the default bench line, the native path alone
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05c
mkdir -p $OUT
cd $REPO
(timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v '^$' | tail -40) > $OUT/gputest.log
timeout 480 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null
timeout 240 python bench.py --gpus 1 --native > $OUT/bench_native.json 2> $OUT/bench_native.err < /dev/null
tail -5 $OUT/gputest.log
cut -c1-300 $OUT/bench.json
