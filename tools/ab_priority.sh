# same-box A/B: queue priority of the pipeline's second stream (superpixel branch) against the forward's stream
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else None)"
for i in 1 2; do
for p in 0 1 -1; do
SPA_AUX_PRIORITY=$p python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('aux priority $p', d['value'], d['device_resident_value'], d['stage_ms_per_step'])"
done
done
