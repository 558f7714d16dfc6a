#!/bin/bash
# On the GPU box: time every diagnostic library under build_variants/ (built locally) with tools/kbench.py.
cd "$(dirname "$0")/.."
for so in build_variants/libcpg_v_*.so; do
  n=${so#build_variants/libcpg_v_}; n=${n%.so}
  echo "== $n: $(cat build_variants/v_$n.txt)"
  CPG_LIB_PATH=$PWD/$so timeout 120 python tools/kbench.py --iters 5 ${KBENCH_ARGS} 2>&1 | grep "^\[1\].*persistent"
done
