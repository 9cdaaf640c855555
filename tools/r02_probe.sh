#!/bin/bash
# One gpurun call: instruction-rate and memory-system microbenchmarks, table-footprint sweep of K4, SQ / TLB / L2 counters of K4 (both loop forms).
cd "$(dirname "$0")/.."
R=$PWD; OUT=gpurun_out/probe_r02; mkdir -p $OUT
timeout 120 build/ubench/valu_rate > $OUT/valu_rate.txt 2>&1
timeout 200 build/ubench/gather_rate > $OUT/gather_rate.txt 2>&1
export KEXP='[{}, {"GLIM_AMD_BUCKET_FACTOR": 3}, {"GLIM_AMD_BUCKET_FACTOR": 12}, {"GLIM_AMD_TARGET_BLOCKS": 2560}]'
LIBS="p2 old" REPS=1 timeout 400 bash tools/kexp.sh > $OUT/kexp.log 2>&1
unset KEXP
export PMC_GROUPS_FILE=$R/tools/pmc_groups_short.txt
timeout 400 bash tools/pmc_kexp.sh p2 p2 > $OUT/pmc_p2.txt 2>&1
timeout 400 bash tools/pmc_kexp.sh old old > $OUT/pmc_old.txt 2>&1
cd $R
cat $OUT/valu_rate.txt; cat $OUT/gather_rate.txt; grep kernel_us $OUT/kexp.log; tail -25 $OUT/pmc_p2.txt; tail -25 $OUT/pmc_old.txt
