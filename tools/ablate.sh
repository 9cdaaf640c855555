#!/bin/bash
# Profiling-only: builds ablated variants of the VGICP kernel (glim_amd/csrc/vgicp.hip, GLIM_AMD_ABLATE) and times them.
# Run on the GPU box.  The results of ablated kernels are wrong by construction; only their timing is meaningful.
set -e
cd "$(dirname "$0")/.."
SRC=glim_amd/csrc
for a in ${ABLATIONS:-1 2 3}; do
  mkdir -p /tmp/abl$a
  for f in context cloud voxelmap vgicp covariance knn; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -DGLIM_AMD_ABLATE=$a -c $SRC/$f.hip -o /tmp/abl$a/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/abl$a/*.o -o /tmp/abl$a/libglim_amd.so
done
S='[{"GLIM_AMD_U":1}]'
echo "baseline"; SWEEP=$S python tools/sweep.py 2>&1 | grep kernel_us
for a in ${ABLATIONS:-1 2 3}; do echo "ablate $a"; GLIM_AMD_LIB=/tmp/abl$a/libglim_amd.so SWEEP=$S python tools/sweep.py 2>&1 | grep kernel_us; done
