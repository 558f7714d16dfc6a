#!/bin/bash
# Diagnostic builds of the one-launch BPTT kernel (csrc/gru.hip, CPG_CHAIN_ABLATE bits; results WRONG by construction), built
# HERE with hipcc and shipped in build_variants/; on the GPU box:  bash tools/ablate_chain.sh run
set -e
cd "$(dirname "$0")/.."
SRC=controlled-peptide-generation_amd/csrc
VARIANTS=${VARIANTS:-"0 1 2 17"}
if [ "$1" = "run" ]; then
  for v in $VARIANTS; do
    echo "== CPG_CHAIN_ABLATE=$v"
    CPG_LIB_PATH=$PWD/build_variants/libcpg_ca_$v.so timeout 120 python tools/kbench.py --iters 5 ${KBENCH_ARGS} 2>&1 | grep "^\[1\].*chain"
  done
  exit 0
fi
mkdir -p build_variants
OBJS=""
for f in api gemm gru_persist lstm decode decode_fused losses optim rng class classifier; do OBJS="$OBJS controlled-peptide-generation_amd/_build/$f.hip.o"; done
for v in $VARIANTS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DCPG_CHAIN_ABLATE=$v $EXTRA -I $SRC -c $SRC/gru.hip -o /tmp/gc_$v.o &
done
wait
for v in $VARIANTS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/gc_$v.o -o build_variants/libcpg_ca_$v.so
done
ls -la build_variants/
