#!/bin/bash
# usage: tools/ab_variant.sh <name> [extra hipcc flags...]  -> build/ab/<name>/libglim_amd.so
# A/B build of the factor kernels only: vgicp.hip is recompiled with the extra flags and linked with the objects of the main build
# (glim_amd/csrc/*.o, `make -C glim_amd/csrc` first).  build/ is git-ignored but travels with gpurun; tools/kexp.sh picks the library per process.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/ab/$name
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize "$@" -c glim_amd/csrc/vgicp.hip -o build/ab/$name/vgicp.o
objs=$(ls glim_amd/csrc/*.o | grep -v /vgicp.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/ab/$name/vgicp.o $objs -ldl -lpthread -o build/ab/$name/libglim_amd.so
rm build/ab/$name/vgicp.o
echo build/ab/$name/libglim_amd.so
