#!/bin/bash
# FETCH_SIZE calibration (GPU box): the no-gather ablation reads a KNOWN byte count (40 B per point, coalesced), which gives the
# gfx950 FETCH_SIZE correction for this access mix; the shipped kernel's counter then yields its real HBM read traffic.
cd "$(dirname "$0")/.."
REPO=$(pwd)
mkdir -p /tmp/abl1 gpurun_out/fetch_calib
for f in context cloud voxelmap vgicp covariance knn; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -DGLIM_AMD_ABLATE=1 -c glim_amd/csrc/$f.hip -o /tmp/abl1/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/abl1/*.o -o /tmp/abl1/libglim_amd.so
cd /tmp && export TMPDIR=/tmp
S='[{"GLIM_AMD_U":1}]'
for v in base abl1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    if [ $v = abl1 ]; then export GLIM_AMD_LIB=/tmp/abl1/libglim_amd.so; else unset GLIM_AMD_LIB; fi
    SWEEP=$S rocprofv3 --pmc $c --output-format csv -d $REPO/gpurun_out/fetch_calib/${v}_$c -- python $REPO/tools/sweep.py > /dev/null 2>&1
  done
done
python3 - <<PY
import csv, glob, collections
for v in ('base','abl1'):
    for c in ('FETCH_SIZE','WRITE_SIZE'):
        for f in glob.glob('$REPO/gpurun_out/fetch_calib/%s_%s/*/*counter_collection.csv' % (v,c)):
            vals=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'vgicp_kernel' in r['Kernel_Name']]
            print(v, c, len(vals), sum(vals)/max(1,len(vals)))
PY
