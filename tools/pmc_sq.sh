#!/bin/bash
# SQ-level counters for the fused VGICP kernel (separate PMC pass; no tracing domains combined).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_sq_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${BENCH_ARGS:---steps 5 --warmup 2 --no-cpu-baseline}"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -- python $REPO/bench.py $ARGS > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python $REPO/bench.py $ARGS > $OUT/b.json 2> $OUT/b.err
python3 - <<PY
import csv, glob, collections
for sub in ('a','b'):
    for f in glob.glob('$OUT/'+sub+'/*/*counter_collection.csv'):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'vgicp_kernel' in r['Kernel_Name'] and int(r['Grid_Size']) > 500000:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()): print(k, len(v), sum(v)/len(v))
PY
