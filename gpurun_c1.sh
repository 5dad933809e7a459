python3 tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r2_cv4.log
cat gpurun_out/r2_cv4.log
