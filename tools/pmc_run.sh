#!/bin/bash
# PMC passes (counters only, with --kernel-trace; never combined with other trace domains) over one command.
#   bash tools/pmc_run.sh OUTDIR -- python tools/kb.py --iters 2
# Passes follow the SQ (8) / TCC (4: FETCH_SIZE costs 3, WRITE_SIZE 2) / GRBM (2) slot limits of MI355X_MICROARCH.md.
set -e
OUT=$1; shift; shift
cd /tmp 2>/dev/null || true
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
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
