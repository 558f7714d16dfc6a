#!/bin/bash
# Diagnostic builds of the persistent GRU kernel (phases removed - results WRONG by construction), built HERE with hipcc and
# shipped in build_variants/; on the GPU box:  bash tools/ablate_persist.sh run
set -e
cd "$(dirname "$0")/.."
SRC=controlled-peptide-generation_amd/csrc
VARIANTS=${VARIANTS:-"0 1"}
if [ "$1" = "run" ]; then
  for v in $VARIANTS; do
    echo "== CPG_PERSIST_ABLATE=$v"
    CPG_LIB_PATH=$PWD/build_variants/libcpg_pa_$v.so python tools/kbench.py --iters 5 ${KBENCH_ARGS} 2>&1 | grep "^\[1\].*persistent"
  done
  exit 0
fi
mkdir -p build_variants
OBJS=""
for f in api gemm gru lstm decode decode_fused losses optim rng class classifier; do OBJS="$OBJS controlled-peptide-generation_amd/_build/$f.hip.o"; done
for v in $VARIANTS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DCPG_PERSIST_ABLATE=$v $EXTRA -I $SRC -c $SRC/gru_persist.hip -o /tmp/gp_$v.o &
done
wait
for v in $VARIANTS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/gp_$v.o -o build_variants/libcpg_pa_$v.so
done
ls -la build_variants/
