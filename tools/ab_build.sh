#!/bin/bash
# usage: tools/ab_build.sh <name> <extra hipcc flags...>   -> /tmp/ab_<name>/libglim_amd.so  (profiling A/B builds)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p /tmp/ab_$name
for f in context cloud voxelmap vgicp covariance knn; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize "$@" -c glim_amd/csrc/$f.hip -o /tmp/ab_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/ab_$name/*.o -o /tmp/ab_$name/libglim_amd.so
echo /tmp/ab_$name/libglim_amd.so
