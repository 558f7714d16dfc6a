#!/bin/bash
# PMC passes over one micro-benchmark command:  bash tools/micro/pmc_micro.sh OUTDIR -- build_variants/wgrad_planes 512 51200 1024 1 2
set -e
OUT=$1; shift; shift
cd /tmp 2>/dev/null || true
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/pass$i" -o p -- "$@" > "$OUT/pass$i.log" 2>&1 || { tail -5 "$OUT/pass$i.log"; exit 1; }
done
python tools/pmc_summary.py "$OUT" --skip-first 1
