export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "connectivity or slic_edge or slic_full or starved" 2>&1 | tail -8 > gpurun_out/r2_c3.log
python3 tools/prof_stages.py --batch 30 --reps 3 2>&1 | grep -v "^[WE]2026" >> gpurun_out/r2_c3.log
cd superpixel-align_amd/csrc && touch spa_connect.hip && make EXTRA=-DSPA_CONN_TIMING > /dev/null 2>&1; cd ../..
python3 tools/prof_stages.py --batch 30 --reps 1 2>&1 | grep "bfs tier" | sort -k 14 -n -r | head -30 >> gpurun_out/r2_c3.log
cat gpurun_out/r2_c3.log
