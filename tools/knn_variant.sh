#!/bin/bash
# usage: knn_variant.sh <name> [flags]  -> build/ab/<name>/libglim_amd.so (k = 10 kernels only)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/ab/$name
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -DGLIM_AMD_DEV_K10 "$@" -c glim_amd/csrc/knn.hip -o build/ab/$name/knn.o
objs=$(ls glim_amd/csrc/*.o | grep -v /knn.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/ab/$name/knn.o $objs -ldl -lpthread -o build/ab/$name/libglim_amd.so
rm build/ab/$name/knn.o
echo build/ab/$name/libglim_amd.so
