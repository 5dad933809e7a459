python -m pytest tests/test_gpu_parity.py tests/test_gpu_baselines.py -q -m gpu -x -k "kmeans or baseline or direct or anchor_pipeline" 2>&1 | tail -3 > gpurun_out/r2_km.log
python3 tools/km_bench.py 4600 514 2 2>&1 | grep "DIV  32" >> gpurun_out/r2_km.log
python3 tools/km_bench.py 12000 514 2 2>&1 | grep "DIV  32" >> gpurun_out/r2_km.log
python3 tools/km_bench.py 12000 514 4 2>&1 | grep "DIV  32" >> gpurun_out/r2_km.log
cat gpurun_out/r2_km.log
