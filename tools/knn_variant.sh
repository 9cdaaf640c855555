#!/bin/bash
# usage: knn_variant.sh <name> [flags]  -> build/ab/<name>/libglim_amd.so (k = 10 kernels only)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/ab/$name
# the three translation units of kernel group K2 are recompiled with the flags (in parallel), everything else comes from the main build
for f in knn knn_chunks knn_pairs; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -DGLIM_AMD_DEV_K10 "$@" -c glim_amd/csrc/$f.hip -o build/ab/$name/$f.o &
done
wait
objs=$(ls glim_amd/csrc/*.o | grep -v '/knn[_a-z]*\.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/ab/$name/knn.o build/ab/$name/knn_chunks.o build/ab/$name/knn_pairs.o $objs -ldl -lpthread -o build/ab/$name/libglim_amd.so
rm build/ab/$name/knn.o build/ab/$name/knn_chunks.o build/ab/$name/knn_pairs.o
echo build/ab/$name/libglim_amd.so
