#!/bin/bash
# GPU side of a kernel A/B: for every build/ab/<name>/libglim_amd.so named in $LIBS run tools/kexp.py with $KEXP; results in gpurun_out/kexp/
cd "$(dirname "$0")/.."
OUT=gpurun_out/kexp; mkdir -p $OUT
export GLIM_AMD_SCAN_CACHE=/tmp/glim_amd_scan_cache
for rep in $(seq 1 ${REPS:-2}); do
for name in ${LIBS:-main}; do
  GLIM_AMD_LIB=$PWD/build/ab/$name/libglim_amd.so KEXP_TAG=$name timeout 300 python tools/kexp.py < /dev/null 2> $OUT/$name.err | tee -a $OUT/results.jsonl
done
done
