set -x
CPG_LIB_PATH=$PWD/build_variants/libcpg_sub2.so timeout 900 python -m pytest tests/test_gpu_persistent.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_persistent.py -x -q -m gpu 2>&1 | tail -2
bash tools/insitu.sh 2>&1 | tail -6
for v in sub2 sub2d4; do CPG_LIB_PATH=$PWD/build_variants/libcpg_$v.so python bench.py --steps 10 --warmup 3 --no-extra-legs --no-class --no-cpu-baseline --dtype bf16 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); f=[d['roofline']]+d['extra']['kernel_families']
print('bf16 $v', d['ms_per_step'], {x['family']: x['avg_launch_us'] for x in f})"; done
python bench.py --steps 10 --warmup 3 --no-extra-legs --no-class --no-cpu-baseline --dtype bf16 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); f=[d['roofline']]+d['extra']['kernel_families']
print('bf16 current', d['ms_per_step'], {x['family']: x['avg_launch_us'] for x in f})"
