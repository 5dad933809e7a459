python -m pytest tests/test_gpu_baselines.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r2_s64.log
cat gpurun_out/r2_s64.log
