#!/bin/bash
# Memory-pipeline counters of the fused VGICP kernel on the bench workload (tools/kexp.py), one rocprofv3 --pmc pass per counter group
# (PMC passes only: never combined with tracing domains).  usage (GPU box): bash tools/pmc_kexp.sh <tag> [lib-name under build/ab]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}; LIB=${2:-}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
export GLIM_AMD_SCAN_CACHE=/tmp/glim_amd_scan_cache
[ -n "$LIB" ] && export GLIM_AMD_LIB=$REPO/build/ab/$LIB/libglim_amd.so
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $group --output-format csv -d $OUT/g$i -- python $REPO/tools/kexp.py > $OUT/g$i.out 2> $OUT/g$i.err
done < <(if [ -n "$PMC_GROUPS_FILE" ]; then cat "$PMC_GROUPS_FILE"; else cat <<'GROUPS'
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_REQ_sum
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES
TD_TD_BUSY_sum TD_TC_STALL_sum TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_READ_SECTORS_sum
GROUPS
fi)
python3 - <<PY
import csv, glob, collections, json
agg = collections.OrderedDict()
for f in sorted(glob.glob('$OUT/g*/*/*counter_collection.csv')):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'vgicp_kernel' in r['Kernel_Name'] and int(r['Grid_Size']) > 200000:
            per[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in per.items():
        agg[k] = {"launches": len(v), "mean": sum(v) / len(v)}
json.dump(agg, open('$OUT/summary.json', 'w'), indent=1)
for k, v in agg.items():
    print(f"{k:45s} {v['launches']:5d} {v['mean']:.4g}")
PY
