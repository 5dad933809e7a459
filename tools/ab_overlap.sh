# same-box A/B: the superpixel branch on a second stream (--overlap) against the one-stream default, alternating
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no_cpu_baseline --one_stream 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('one stream ', d['value'], d['device_resident_value'])"
python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('two streams', d['value'], d['device_resident_value'])"
done
