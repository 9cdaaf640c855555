#!/bin/bash
# A/B of the untested experiments on branch exp/next-round against main, prepared on the CPU side and run in ONE gpurun call.
#
#   CPU side:   tools/ab_exp.sh build          -> build/ab/exp/libglim_amd.so from the branch (git worktree; build/ travels with gpurun)
#   GPU side:   gpurun --timeout 700 -- 'bash tools/ab_exp.sh run'
#               1. parity: the whole GPU suite against the experimental library (GLIM_AMD_LIB selects it, glim_amd/_lib.py)
#               2. timing: tools/batch_sweep.py (1 / 8 / 64 factors per launch) and the default bench, main vs exp, interleaved
#               results in gpurun_out/ab_exp/
set -u
cd "$(dirname "$0")/.."
REPO=$(pwd)
case "${1:-}" in
  build)
    set -e
    rm -rf build/ab/exp_src && mkdir -p build/ab
    git worktree remove --force build/ab/exp_src 2>/dev/null || true
    git worktree add --force build/ab/exp_src exp/next-round
    make -C build/ab/exp_src/glim_amd/csrc -j8 > /dev/null
    mkdir -p build/ab/exp && cp build/ab/exp_src/glim_amd/libglim_amd.so build/ab/exp/libglim_amd.so
    git worktree remove --force build/ab/exp_src
    ls -la build/ab/exp/libglim_amd.so
    ;;
  run)
    OUT=$REPO/gpurun_out/ab_exp
    mkdir -p $OUT
    EXP=$REPO/build/ab/exp/libglim_amd.so
    [ -f "$EXP" ] || { echo "missing $EXP: run 'tools/ab_exp.sh build' first"; exit 1; }
    GLIM_AMD_LIB=$EXP timeout 150 python -m pytest tests -m gpu -x -q -p no:cacheprovider < /dev/null > $OUT/parity_exp.log 2>&1
    tail -3 $OUT/parity_exp.log
    GLIM_AMD_LIB=$EXP GLIM_AMD_FUSED_FINALIZE=1 timeout 100 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_edge_cases.py -m gpu -x -q -p no:cacheprovider < /dev/null > $OUT/parity_expfused.log 2>&1
    tail -3 $OUT/parity_expfused.log
    GLIM_AMD_LIB=$EXP GLIM_AMD_U=2 timeout 100 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_edge_cases.py -m gpu -x -q -p no:cacheprovider < /dev/null > $OUT/parity_expu2.log 2>&1
    tail -3 $OUT/parity_expu2.log
    export GLIM_AMD_SCAN_CACHE=/tmp/glim_amd_scan_cache   # scene generation (host numpy) once, not once per variant
    for rep in 1 2; do
      # expfused: the branch's single-dispatch finalisation (GLIM_AMD_FUSED_FINALIZE=1); expu2: two points per lane, packed FP32 (GLIM_AMD_U=2)
      for v in main exp expfused expu2; do
        unset GLIM_AMD_LIB GLIM_AMD_FUSED_FINALIZE GLIM_AMD_U
        if [ $v != main ]; then export GLIM_AMD_LIB=$EXP; fi
        if [ $v = expfused ]; then export GLIM_AMD_FUSED_FINALIZE=1; fi
        if [ $v = expu2 ]; then export GLIM_AMD_U=2; fi
        if [ $rep = 1 ]; then timeout 60 python tools/batch_sweep.py < /dev/null > $OUT/sweep_${v}_$rep.json 2> $OUT/sweep_${v}_$rep.err; fi
        timeout 120 python bench.py --no-cpu-baseline < /dev/null > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err
      done
    done
    unset GLIM_AMD_LIB GLIM_AMD_FUSED_FINALIZE GLIM_AMD_U
    python - <<'PY'
import glob, json, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "ab_exp")
for f in sorted(glob.glob(os.path.join(out, "bench_*.json"))):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), round(d["value"]), "calls/s, kernel", round(d["roofline"]["kernel_ms"] * 1e3, 1), "us, sync", round(d["sync_single_factor_calls_per_s"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
    ;;
  *)
    echo "usage: tools/ab_exp.sh build | run"; exit 2;;
esac
