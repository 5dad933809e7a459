# stem kernel: tests, LDS conflict counters and a same-box A/B against ab/libspalign_old.so
python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "stem or network or drn_c" 2>&1 | tail -2
for i in 1 2; do
SPA_LIB_PATH=$PWD/ab/libspalign_old.so python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_host_loop 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('old', d['value'], d['kernels']['k_drn_stem_d(+normalise)']['avg_ms'])"
python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_host_loop 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new', d['value'], d['kernels']['k_drn_stem_d(+normalise)']['avg_ms'])"
done
bash tools/pmc_lds.sh > /dev/null 2>&1; grep -E "stem|layer2" gpurun_out/sq_lds.txt | awk '{print $1,$2,$(NF)}'
