timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tiles.py -q -x -m gpu 2>&1 | tail -3
bash tools/insitu.sh 2>&1 | tail -5
bash tools/insitu.sh 2>&1 | tail -5
