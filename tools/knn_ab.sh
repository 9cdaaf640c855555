#!/bin/bash
# Same-box A/B of the kNN chunk-kernel variants that are staged behind compile-time macros in glim_amd/csrc/knn_chunks.hip / knn_pairs.hip (k = 10 builds).
#   here (CPU):   tools/knn_ab.sh build            -> build/ab/{kp0,sel,gb_sel,pk,all3}/libglim_amd.so
#   on the GPU:   gpurun --timeout 120 -- 'tools/knn_ab.sh run'      (timings + exactness against the oracle; then the kNN parity tests with the
#                                                                    full variant library, which needs all k: tools/knn_ab.sh full builds it)
set -e
cd "$(dirname "$0")/.."
case "$1" in
  build)
    tools/knn_variant.sh kp0
    tools/knn_variant.sh sel -DGLIM_AMD_KNN_SELECT
    tools/knn_variant.sh gb_sel -DGLIM_AMD_KNN_SELECT -DGLIM_AMD_KNN_GROUPBOX
    tools/knn_variant.sh pk -DGLIM_AMD_KNN_PKMASK
    tools/knn_variant.sh all3 -DGLIM_AMD_KNN_SELECT -DGLIM_AMD_KNN_GROUPBOX -DGLIM_AMD_KNN_PKMASK
    ;;
  full)  # every k, all three macros: the library the parity tests should be run with before the macros become the default
    mkdir -p build/ab/full
    for f in knn knn_chunks knn_pairs; do
      /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -DGLIM_AMD_KNN_SELECT -DGLIM_AMD_KNN_GROUPBOX -DGLIM_AMD_KNN_PKMASK \
        -c glim_amd/csrc/$f.hip -o build/ab/full/$f.o &
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/ab/full/knn.o build/ab/full/knn_chunks.o build/ab/full/knn_pairs.o \
      $(ls glim_amd/csrc/*.o | grep -v '/knn[_a-z]*\.o') -ldl -lpthread -o build/ab/full/libglim_amd.so
    rm build/ab/full/knn.o build/ab/full/knn_chunks.o build/ab/full/knn_pairs.o
    ;;
  run)
    mkdir -p gpurun_out/knn_ab
    for v in kp0 sel gb_sel pk all3 kp0; do
      echo "== $v" >> gpurun_out/knn_ab/ab.txt
      GLIM_AMD_LIB=build/ab/$v/libglim_amd.so timeout 30 python tools/knn_time.py 2>&1 | grep knn >> gpurun_out/knn_ab/ab.txt
    done
    # the shipped library: default, and the run-time switches of the staged selection with either chunk kernel forced
    for e in "" "GLIM_AMD_KNN_SELECT=1" "GLIM_AMD_KNN_SELECT=1 GLIM_AMD_KNN_WAVE64=1" "GLIM_AMD_KNN_SELECT=1 GLIM_AMD_KNN_PAIR=1" "GLIM_AMD_KNN_WAVE64=1" "GLIM_AMD_KNN_PAIR=1"; do
      echo "== shipped library, env: ${e:-default}" >> gpurun_out/knn_ab/ab.txt
      env $e timeout 30 python tools/knn_time.py 2>&1 | grep knn >> gpurun_out/knn_ab/ab.txt
    done
    cat gpurun_out/knn_ab/ab.txt
    if [ -f build/ab/full/libglim_amd.so ]; then
      GLIM_AMD_LIB=build/ab/full/libglim_amd.so timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_edge_cases.py tests/test_preprocess.py -q -x -m gpu -k "knn or neighb or covar or frontend or rgbd or preprocess" 2>&1 | tail -5 | tee gpurun_out/knn_ab/tests.txt
    fi
    ;;
  *) echo "usage: $0 build|full|run"; exit 2;;
esac
