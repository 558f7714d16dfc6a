cd /root/repo
rm -f gpurun_out/ab_*.json
i=0
for m in rnn z 0 rnn z 0; do
i=$((i+1))
CPG_ASIDE=$m python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-class --no-extra-legs 2>> gpurun_out/b_r6b.err > gpurun_out/ab_f32_${m}_$i.json
done
for m in rnn z 0 rnn z 0; do
i=$((i+1))
CPG_ASIDE=$m python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-class --no-extra-legs --dtype bf16 2>> gpurun_out/b_r6b.err > gpurun_out/ab_bf16_${m}_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['value'])
PY
