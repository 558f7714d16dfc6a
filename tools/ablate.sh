#!/bin/bash
# Diagnostic: build ablated variants of the library (phases of the MFMA slab loop removed, results WRONG by construction)
# and time the recurrent kernels with each, to see which phase costs what.  Run on the GPU box: bash tools/ablate.sh
set -e
cd "$(dirname "$0")/.."
SRC=controlled-peptide-generation_amd/csrc
mkdir -p gpurun_out
for m in 0 1 2 3 4 7; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -DCPG_ABLATE=$m -I $SRC \
     $SRC/api.hip $SRC/gemm.hip $SRC/gru.hip $SRC/lstm.hip $SRC/decode.hip $SRC/losses.hip $SRC/optim.hip $SRC/rng.hip $SRC/class.hip \
     -o /tmp/libcpg_ablate_$m.so
  echo "== CPG_ABLATE=$m  (1: no global loads/LDS writes, 2: no LDS fragment reads, 4: no barrier)"
  CPG_LIB_PATH=/tmp/libcpg_ablate_$m.so python tools/kbench.py --iters 5 "$@" | grep "^\[1\]"
done
