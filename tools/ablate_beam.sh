#!/bin/bash
# On the GPU box: decode time of one 262 144-proposal beam round with each diagnostic build of the fused beam kernel
# (build_variants/libcpg_v_ba<bits>.so, CPG_BEAM_ABLATE bits: see csrc/decode_fused.hip; results wrong by construction).
cd "$(dirname "$0")/.."
for so in build_variants/libcpg_v_ba*.so; do
  n=${so#build_variants/libcpg_v_ba}; n=${n%.so}
  echo "== CPG_BEAM_ABLATE=$n: $(CPG_LIB_PATH=$PWD/$so timeout 200 python tools/class_profile.py 262144 2>&1 | grep decode_ids | tail -1)"
done
