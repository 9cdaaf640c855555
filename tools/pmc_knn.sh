#!/bin/bash
# SQ-level counters for the kNN walker (separate PMC passes; no tracing domains combined).  Run on the GPU box.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_knn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -- python $REPO/tools/knn_prof.py lidar > /dev/null 2> $OUT/a.err < /dev/null
timeout 100 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python $REPO/tools/knn_prof.py lidar > /dev/null 2> $OUT/b.err < /dev/null
python3 - <<PY
import csv, glob, collections
for sub in ('a','b'):
    for f in glob.glob('$OUT/'+sub+'/*/*counter_collection.csv'):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if ('knn_grid_kernel' in r['Kernel_Name'] or 'knn_chunk_kernel' in r['Kernel_Name']) and int(r['Grid_Size']) > 100000:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()): print(k, len(v), sum(v)/len(v))
PY
