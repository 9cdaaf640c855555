#!/bin/bash
# One gpurun call: A/B of the factor-kernel builds (build/ab/*), of the kNN variants, then the GPU test-suite on the main build.
cd "$(dirname "$0")/.."
OUT=gpurun_out/ab_r02; mkdir -p $OUT
export KEXP='[{}, {"GLIM_AMD_NO_PLANE": 1}]'
LIBS="${LIBS:-old p2 p2g5 p2w4}" REPS=1 timeout 600 bash tools/kexp.sh > $OUT/kexp.log 2>&1
for w in 0 2 4; do
  if [ $w = 0 ]; then unset GLIM_AMD_KNN_WPC; else export GLIM_AMD_KNN_WPC=$w; fi
  echo "== WPC=$w" >> $OUT/knn.log
  timeout 300 python tools/knn_time.py >> $OUT/knn.log 2>&1
done
unset GLIM_AMD_KNN_WPC
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $OUT/gputest.log
cat $OUT/kexp.log | grep kernel_us; cat $OUT/knn.log; cat $OUT/gputest.log
