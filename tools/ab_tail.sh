# same-box A/B: the batch's tail on the second stream (next batch's forward under it) against the tail on the main stream
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_config1.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
SPA_PIPE_TAIL_AUX=0 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('tail on main', d['value'], d['device_resident_value'], d['quality'])"
python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('tail on aux ', d['value'], d['device_resident_value'], d['quality'])"
done
