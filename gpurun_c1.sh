export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "connectivity or slic_edge or slic_full or starved" 2>&1 | tail -8 > gpurun_out/r2_c3.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_stats_p/runc -- python3 tools/prof_stages.py --batch 30 --reps 3 2>&1 | grep -v "^[WE]2026" >> gpurun_out/r2_c3.log
python3 tools/conn_timeline.py gpurun_out/r2_stats_p >> gpurun_out/r2_c3.log
cat gpurun_out/r2_c3.log
