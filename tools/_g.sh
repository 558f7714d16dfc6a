timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -3
cat gpurun_out/persist_handoff_report.json
bash tools/insitu.sh 2>&1 | tail -2
