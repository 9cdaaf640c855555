#!/bin/bash
# round 5, fifth GPU call: the two failing tests with full tracebacks, then the new parity tests, the default bench line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05e
mkdir -p $OUT
cd $REPO
(timeout 300 python -m pytest tests/test_gpu_edge_cases.py::test_frame_create_equals_the_separate_calls tests/test_twins.py tests/test_gpu_parity.py::test_voxelmap_lru_horizon_matches_oracle_over_an_insert_sequence tests/test_gpu_parity.py::test_plane_view_written_with_the_map_equals_the_one_built_on_first_use -m gpu -q -p no:cacheprovider -s 2>&1 | grep -v '^$' | cut -c1-400 | tail -120) > $OUT/gputest_focus.log
timeout 480 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null
tail -40 $OUT/gputest_focus.log
cut -c1-300 $OUT/bench.json
