#!/bin/bash
# Profiling-only: ablated builds of the fused VGICP kernel (glim_amd/csrc/vgicp.hip, GLIM_AMD_ABLATE) for tools/kexp.sh.
#   1  no gathers (keys and records synthesised in registers: stream + algebra only)      3  no algebra (loads only)
#   4  key gather only (records synthesised)
# The results of ablated kernels are wrong by construction; only their timing is meaningful.
#   here:        tools/ablate.sh            (builds build/ab/abl{1,3,4})
#   on the GPU:  LIBS="main abl1 abl3 abl4" bash tools/kexp.sh      (main = tools/ab_variant.sh main)
set -e
cd "$(dirname "$0")/.."
for a in ${ABLATIONS:-1 3 4}; do tools/ab_variant.sh abl$a -DGLIM_AMD_ABLATE=$a; done
