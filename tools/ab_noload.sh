# compute-only timing of the direct split-plane kernels: ab/libspalign_noload.so = the same build with the K loop's global loads compiled out
SPA_LIB_PATH=$PWD/ab/libspalign_noload.so python tools/conv16_one.py 2>&1 | grep -v amdgpu | sed 's/: f32.*; \([0-9.]* ms vs [0-9.]* ms\)/ noload \1/'
python tools/conv16_one.py 2>&1 | grep -v amdgpu | sed 's/: f32.*; \([0-9.]* ms vs [0-9.]* ms\)/ asis   \1/'
