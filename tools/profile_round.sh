#!/bin/bash
# One command that regenerates the round's profile evidence on the GPU box (outputs under gpurun_out/prof_<tag>/; copy the
# summaries into profiles/).   bash tools/profile_round.sh r02
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (the judged line)  -> kernel stats csv + bench line
#   2. PMC passes (tools/pmc_run.sh: SQ / LDS / FETCH_SIZE / WRITE_SIZE, separate runs) over the training leg only
#      -> per-kernel counter means as JSON (profiles/<tag>_pmc.json is what bench.py reads `roofline.traffic` from)
set -e
TAG=${1:-r04}
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export CPG_BENCH_NO_TORCH_PROFILER=1    # bench.py's launch count uses torch.profiler (roctracer): not under rocprofv3
# --no-extra-legs: the headline leg + CPU baseline + CLaSS leg (the bf16-mode / config-C legs and the sustained region would mix
# their launches into the per-step table); the default command's own line is kept next to it
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --steps 20 --warmup 5 --no-extra-legs > $OUT/bench_line.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
bash tools/pmc_run.sh $OUT/pmc -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-class --no-extra-legs > $OUT/pmc.log 2>&1
python tools/pmc_summary.py --skip-first 2 --stamp-csrc --json $OUT/${TAG}_pmc.json $OUT/pmc > $OUT/pmc_summary.txt
find $OUT/trace -name "*_kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
echo done
