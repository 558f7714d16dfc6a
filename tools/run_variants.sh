#!/bin/bash
# On the GPU box: time every diagnostic library under build_variants/ (built HERE with tools/variant_build.sh, shipped by gpurun)
# with tools/kb.py.   bash tools/run_variants.sh [kb.py arguments, e.g. --only fwdp]
cd "$(dirname "$0")/.."
python tools/kb.py "$@" 2>&1 | grep "^\["
for so in build_variants/libcpg_*.so; do
  [ -e "$so" ] || continue
  CPG_LIB_PATH=$PWD/$so timeout 300 python tools/kb.py "$@" 2>&1 | grep "^\["
done
