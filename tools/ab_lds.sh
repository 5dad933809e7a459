# Same-box A/B of two builds: ab/libspalign_old.so (another build of csrc/, SPA_LIB_PATH) against the in-tree library.
for i in 1 2; do
SPA_LIB_PATH=$PWD/ab/libspalign_old.so python tools/winof_bench.py --shapes 512:512:4,256:256:2 2>&1 | grep -v amdgpu | sed 's/^/old /'
python tools/winof_bench.py --shapes 512:512:4,256:256:2 2>&1 | grep -v amdgpu | sed 's/^/new /'
done
SPA_LIB_PATH=$PWD/ab/libspalign_old.so python tools/conv16_one.py 2>&1 | grep -v amdgpu | sed 's/.*; \([0-9.]* ms vs [0-9.]* ms\)/old \1/'
python tools/conv16_one.py 2>&1 | grep -v amdgpu | sed 's/.*; \([0-9.]* ms vs [0-9.]* ms\)/new \1/'
SPA_LIB_PATH=$PWD/ab/libspalign_old.so python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_host_loop 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('old', d['value'], d['stage_ms_per_step'])"
python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_host_loop 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new', d['value'], d['stage_ms_per_step'])"
SPA_LIB_PATH=$PWD/ab/libspalign_old.so python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_host_loop 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('old', d['value'], d['stage_ms_per_step'])"
python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_host_loop 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new', d['value'], d['stage_ms_per_step'])"
