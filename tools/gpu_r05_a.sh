#!/bin/bash
# round 5, first GPU call: the -m gpu suite at the new library, the default bench line (new headline form, native breakdown, warmed shard simulation)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05a
mkdir -p $OUT
cd $REPO
(timeout 480 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v '^$' | tail -25) > $OUT/gputest.log
timeout 420 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null
timeout 240 python bench.py --gpus 1 --native > $OUT/bench_native.json 2> $OUT/bench_native.err < /dev/null
tail -3 $OUT/gputest.log
cut -c1-600 $OUT/bench.json
