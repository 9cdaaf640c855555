#!/bin/bash
# round 5, second GPU call: K4 Sherman-Morrison A/B (same box, alternating), the -m gpu suite, the default bench line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05b
mkdir -p $OUT
cd $REPO
for rep in 1 2; do
  timeout 200 python tools/k4_ab.py >> $OUT/k4_ab.jsonl 2>> $OUT/k4_ab.err
  GLIM_AMD_LIB=$REPO/build/ab/sm0/libglim_amd.so timeout 200 python tools/k4_ab.py >> $OUT/k4_ab.jsonl 2>> $OUT/k4_ab.err
done
(timeout 480 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v '^$' | tail -40) > $OUT/gputest.log
timeout 420 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null
cat $OUT/k4_ab.jsonl | cut -c1-400
tail -5 $OUT/gputest.log
cut -c1-300 $OUT/bench.json
