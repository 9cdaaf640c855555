#!/bin/bash
# the shipped tree once more: smoke() and the -m gpu suite
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05r
mkdir -p $OUT
cd $REPO
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
(timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v '^$' | cut -c1-300 | tail -15) > $OUT/gputest.log
tail -2 $OUT/smoke.log; grep -h "passed\|failed" $OUT/gputest.log
