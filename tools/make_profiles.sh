#!/bin/bash
# Refresh the judged profiles (run on the GPU box from the repo root):
#   1. the default bench line                                   -> gpurun_out/final_bench_line.json
#   2. rocprofv3 --kernel-trace --stats of the default bench    -> gpurun_out/final_stats/
#   3. HBM traffic PMC passes (FETCH_SIZE, WRITE_SIZE; separate passes) over tools/prof_stages.py
#   4. the other operating points (one JSON line each)          -> gpurun_out/final_variant_*.json
# then, in the build container:  python tools/write_profiles.py r6_final
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/final_stats gpurun_out/final_pmc_* gpurun_out/final_variant_*
python3 bench.py --no_cpu_baseline --no_exact_fp32 --steps 10 --warmup 3 > /dev/null 2>&1      # a fresh box's first process reads 4-8 % low (allocator, clocks): not the line
python3 bench.py --steps 20 --warmup 5 > gpurun_out/final_bench_line.json 2> gpurun_out/final_bench.err      # the step counts the round driver uses
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_stats -- python3 bench.py --no_cpu_baseline --no_exact_fp32 --steps 20 --warmup 5 > gpurun_out/final_stats_bench_line.json 2> gpurun_out/final_stats.err
for P in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/final_pmc_$P -- python3 tools/prof_stages.py --batch 8 --reps 2 --bias_act --wino > gpurun_out/final_pmc_$P.log 2>&1
done
python3 tools/pmc_summary.py gpurun_out/final_pmc_FETCH_SIZE > gpurun_out/final_pmc_fetch.txt
python3 tools/pmc_summary.py gpurun_out/final_pmc_WRITE_SIZE > gpurun_out/final_pmc_write.txt
rm -rf gpurun_out/final_pmc_FETCH_SIZE gpurun_out/final_pmc_WRITE_SIZE          # raw counter dumps: tens of MB
v() { name=$1; shift; python3 bench.py --no_cpu_baseline --no_exact_fp32 --steps 20 --warmup 5 "$@" 2> gpurun_out/final_variant_$name.err | tail -1 > gpurun_out/final_variant_$name.json; }
v bf16 --dtype bf16
v cfg5 --dtype bf16 --n_slic_segments 400
v cfg5_one_stream --dtype bf16 --n_slic_segments 400 --one_stream
v one_stream --one_stream
v anchor --pool_mode anchor
v anchor_bf16 --pool_mode anchor --dtype bf16
v anchor_bf16_device_rng --pool_mode anchor --dtype bf16 --device_rng
v float_images --float_images
v fp32_mfma_gemm --fp32_mfma_gemm
v drn_c_26 --arch drn_c_26
v drn_c_26_bf16 --arch drn_c_26 --dtype bf16
v reference_operating_point --superpixel_method felzenszwalb --height 224 --width 224 --arch drn_c_26 --pool_mode anchor --n_clusters 4
v k4 --n_clusters 4
SPA_KM_HOST_INIT=1 python3 bench.py --no_cpu_baseline --no_exact_fp32 --steps 20 --warmup 5 --n_clusters 4 2> gpurun_out/final_variant_k4_host_init.err | tail -1 > gpurun_out/final_variant_k4_host_init.json
SPA_GEMM16_STAGGER=0 SPA_CONV16_STAGGER=0 python3 bench.py --no_cpu_baseline --no_exact_fp32 --steps 20 --warmup 5 2> gpurun_out/final_variant_round4_kernels.err | tail -1 > gpurun_out/final_variant_round4_kernels.json
python3 tools/h2h_probe2.py --steps 10 2>&1 | grep -E "device resident|host loop" > gpurun_out/final_h2h_probe.txt
SPA_LATE_DOWNLOAD=0 python3 tools/h2h_probe2.py --steps 10 2>&1 | grep -E "device resident|host loop" | sed "s/^/[downloads enqueued at once, behind an event] /" >> gpurun_out/final_h2h_probe.txt
python3 -m pytest tests/test_gpu_hostile.py tests/test_gpu_descriptor_parity.py -q -s 2>&1 | grep -E "hostile|per-channel|bulk error|descriptors|passed|failed" > gpurun_out/final_hostile.txt
python3 tools/driver300.py --n 600 --decode_procs 32 2>/dev/null | tail -1 > gpurun_out/final_driver600_procs32.json
ls -la gpurun_out | tail -20
