python3 tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r2_cv5.log
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_config1.py -q -m gpu -x -k "bf16 or conv or drn or config5" 2>&1 | tail -4 >> gpurun_out/r2_cv5.log
python bench.py --dtype bf16 --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 > gpurun_out/r2_cv5_bf16.json
python3 -c "
import json
d=json.loads(open('gpurun_out/r2_cv5_bf16.json').read())
print(d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['kernels']['k_conv3x3_bf16(all)'], d['kernels']['k_bias_act(all)'])
" >> gpurun_out/r2_cv5.log
cat gpurun_out/r2_cv5.log
