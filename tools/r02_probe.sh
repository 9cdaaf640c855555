#!/bin/bash
# One gpurun call: instruction-rate and memory-system microbenchmarks, table-footprint sweep of K4, SQ / TLB / L2 counters of K4 for the
# library variants named in $LIBS (build/ab/<name>/libglim_amd.so from tools/ab_variant.sh; round 2 ran it with the two loop forms).
cd "$(dirname "$0")/.."
R=$PWD; OUT=gpurun_out/probe_r02; mkdir -p $OUT
timeout 120 build/ubench/valu_rate > $OUT/valu_rate.txt 2>&1
timeout 200 build/ubench/gather_rate > $OUT/gather_rate.txt 2>&1
export KEXP='[{}, {"GLIM_AMD_BUCKET_FACTOR": 3}, {"GLIM_AMD_BUCKET_FACTOR": 12}, {"GLIM_AMD_TARGET_BLOCKS": 2560}]'
LIBS="${LIBS:-main}" REPS=1 timeout 400 bash tools/kexp.sh > $OUT/kexp.log 2>&1
unset KEXP
export PMC_GROUPS_FILE=$R/tools/pmc_groups_short.txt
for lib in ${LIBS:-main}; do timeout 400 bash tools/pmc_kexp.sh $lib $lib > $OUT/pmc_$lib.txt 2>&1; done
cd $R
cat $OUT/valu_rate.txt; cat $OUT/gather_rate.txt; grep kernel_us $OUT/kexp.log; tail -25 $OUT/pmc_*.txt
