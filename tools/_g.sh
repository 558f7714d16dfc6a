set -x
mkdir -p gpurun_out/r3e
for i in 1 2; do
( cd build_variants/old && python tools/kb.py --only fwdp --tag old-f32 2>&1 | grep "^\[1" )
python tools/kb.py --only fwdp --tag new-f32 2>&1 | grep "^\[1"
( cd build_variants/old && python tools/kb.py --only fwdp --dtype bf16 --tag old-bf16 2>&1 | grep "^\[1" )
python tools/kb.py --only fwdp --dtype bf16 --tag new-bf16 2>&1 | grep "^\[1"
done > gpurun_out/r3e/ab2.log 2>&1
cat gpurun_out/r3e/ab2.log
timeout 900 python -m pytest tests/test_gpu_persistent.py tests/test_lstm.py -x -q -m gpu 2>&1 | tail -5
python tools/kb.py --only fwdp,fwd --B 512 --H 1024 --T 50 2>&1 | grep "^\["
python tools/kb.py --only fwdp,fwd --B 256 --H 1024 --T 50 2>&1 | grep "^\["
