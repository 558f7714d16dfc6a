bash tools/insitu.sh 2>&1 | tail -4
