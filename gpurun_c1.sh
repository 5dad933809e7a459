python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5 > gpurun_out/final_gpu_tests.log
bash tools/make_profiles.sh > gpurun_out/final_make_profiles.log 2>&1
cat gpurun_out/final_gpu_tests.log; for f in gpurun_out/final_variant_*.json gpurun_out/final_bench_line.json; do python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], (d.get('host_to_host') or {}).get('value'))
except Exception as e: print('$f', 'ERR', e)
"; done
