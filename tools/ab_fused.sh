# same-box A/B in the default two-stream layout: fused Winograd launches (default) against three launches per layer (SPA_WINO_FUSED=0)
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fused        ', d['value'], d['device_resident_value'])"
SPA_WINO_FUSED=0 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('three launch ', d['value'], d['device_resident_value'])"
done
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no_cpu_baseline --one_stream 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('one stream, fused        ', d['value'], d['device_resident_value'])"
SPA_WINO_FUSED=0 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --one_stream 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('one stream, three launch ', d['value'], d['device_resident_value'])"
done
