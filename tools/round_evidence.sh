#!/bin/bash
# Everything profiles/<tag>/ is built from, in one gpurun call (run from the repo root on the GPU box):
#   tools/round_evidence.sh r01
# 1. tools/profile.sh: rocprofv3 kernel trace + stats, then the separate FETCH_SIZE / WRITE_SIZE passes, of the default bench
# 2. the default bench line (with cpu_baseline) and the other workloads
# 3. rocprofv3 kernel stats of the preprocessing front end
# Summaries are made afterwards with tools/summarize_profile.py (CPU side).
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/evidence_$TAG
mkdir -p $OUT
timeout 400 bash $REPO/tools/profile.sh $TAG > $OUT/profile.log 2>&1 < /dev/null
cd $REPO
timeout 200 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null
for w in submap20 global256 rgbd300k frontend128k; do
  timeout 300 python bench.py --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err < /dev/null
done
cd /tmp && export TMPDIR=/tmp
for m in random voxelgrid; do
  timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pp_$m -- python $REPO/tools/preprocess_prof.py $m > $OUT/pp_$m.log 2>&1 < /dev/null
done
ls $OUT
tail -2 $OUT/bench.err
cat $OUT/bench.json
