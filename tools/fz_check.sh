python -m pytest tests/test_gpu_felzenszwalb.py tests/test_gpu_baselines.py -m gpu -x -q 2>&1 | tail -2
python tools/fz_rounds.py 1024 2048 2>&1 | grep -v amdgpu
python tools/fz_fullres.py 1 2>&1 | grep -v amdgpu
python tools/fz_fullres.py 8 2>&1 | grep -v amdgpu
python tools/fz_fullres.py 30 2>&1 | grep -v amdgpu
python tools/fz_small.py 2>&1 | grep -v amdgpu | tail -4
