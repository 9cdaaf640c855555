#!/bin/bash
# One gpurun call: the -m gpu suite, then the bench lines listed as arguments (default: the default line + odometry_frame).
# usage (repo root, on the GPU box):  tools/gpu_call.sh <tag> [workload ...]      -> gpurun_out/<tag>/
TAG=${1:-call}
shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
if [ -z "$SKIP_TESTS" ]; then
  (timeout ${TEST_TIMEOUT:-700} python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider -rP 2>&1 | grep -v "^$" | tail -150) > $OUT/gputest.log
  tail -30 $OUT/gputest.log
fi
for w in "${@:-default odometry_frame}"; do
  for ww in $w; do
    if [ "$ww" = default ]; then
      timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null
    else
      timeout 300 python bench.py --workload $ww > $OUT/bench_$ww.json 2> $OUT/bench_$ww.err < /dev/null
    fi
  done
done
timeout 100 python tools/knn_time.py > $OUT/knn_time.txt 2>&1
for f in $OUT/bench*.json; do echo "== $f"; cut -c1-3000 $f; done
tail -3 $OUT/bench*.err
cat $OUT/knn_time.txt
