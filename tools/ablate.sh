#!/bin/bash
# Diagnostic builds of the library with compile-time knobs (phases of the MFMA slab loop removed - results WRONG by
# construction - or alternative code shapes), timed with tools/kbench.py.  Run on the GPU box:
# (tools/variant_build.sh builds one-file variants HERE instead, in seconds, and ships them with gpurun.)
#   bash tools/ablate.sh "-DCPG_ABLATE=1" "-DCPG_ABLATE=7" "-DCPG_FWD_PREFETCH=0" ...
set -e
cd "$(dirname "$0")/.."
SRC=controlled-peptide-generation_amd/csrc
i=0
for flags in "" "$@"; do
  i=$((i+1))
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden $flags -I $SRC $SRC/*.hip -o /tmp/libcpg_var_$i.so
  echo "== flags: '$flags'"
  CPG_LIB_PATH=/tmp/libcpg_var_$i.so python tools/kbench.py --iters 5 $KBENCH_ARGS | grep "^\[1\]"
done
