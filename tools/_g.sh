set -x
mkdir -p gpurun_out/r3n
bash tools/profile_round.sh r03 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3n/bench_n1.json 2> gpurun_out/r3n/bench_n1.err; tail -1 gpurun_out/r3n/bench_n1.err
timeout 600 env -u WORLD_SIZE python bench.py --gpus 2 --steps 5 --warmup 2 --no-extra-legs --class-proposals 131072 > gpurun_out/r3n/bench_n2.json 2> gpurun_out/r3n/bench_n2.err; tail -1 gpurun_out/r3n/bench_n2.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
