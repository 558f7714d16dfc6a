set -x
mkdir -p gpurun_out/r3g
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_persistent.py -x -q -m gpu 2>&1 | tail -12
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3g/bench_n1.json 2> gpurun_out/r3g/bench_n1.err; tail -4 gpurun_out/r3g/bench_n1.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3g/bench_n1.json') if l.startswith('{')][-1]); e=d['extra']
print(d['ms_per_step'], {x['family']: x['avg_launch_us'] for x in [d['roofline']]+e['kernel_families']})
for k in ('graph_replay','config_a_batch32','sustained'): print(k, e.get(k))
print('bf16', e['bf16_mode']['ms_per_step'], 'C', e['config_c']['ms_per_step'], e['config_c']['roofline']['kernel'])
PY
