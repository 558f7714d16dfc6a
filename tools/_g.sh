set -x
mkdir -p gpurun_out/r3h
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r3h/pytest_all.log 2>&1; tail -15 gpurun_out/r3h/pytest_all.log
bash tools/insitu.sh "tn_tile=128x128" "tn_tile=128x64" "no_overlap=1" 2>&1 | tail -8
