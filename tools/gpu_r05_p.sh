#!/bin/bash
# last checks of the round: smoke(), the -m gpu suite, the live frame line once more (with the gate-alone variant)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05p
mkdir -p $OUT
cd $REPO
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
(timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v '^$' | cut -c1-300 | tail -15) > $OUT/gputest.log
timeout 500 python bench.py --workload odometry_frame > $OUT/bench_odometry_frame.json 2> $OUT/bench_odometry_frame.err < /dev/null
tail -2 $OUT/smoke.log; grep -h "passed\|failed" $OUT/gputest.log
cut -c1-200 $OUT/bench_odometry_frame.json
