#!/bin/bash
# Refresh the judged profiles (run on the GPU box from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench  -> gpurun_out/final_stats/
#   2. the default bench line itself                           -> gpurun_out/final_bench_line.json
#   3. HBM traffic PMC passes (FETCH_SIZE, WRITE_SIZE; separate passes) over tools/prof_stages.py
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/final_stats gpurun_out/final_pmc_*
python3 bench.py > gpurun_out/final_bench_line.json 2> gpurun_out/final_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_stats -- python3 bench.py --no_cpu_baseline > gpurun_out/final_stats_bench_line.json 2> gpurun_out/final_stats.err
for P in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/final_pmc_$P -- python3 tools/prof_stages.py --batch 8 --reps 2 > gpurun_out/final_pmc_$P.log 2>&1
done
python3 tools/pmc_summary.py gpurun_out/final_pmc_FETCH_SIZE > gpurun_out/final_pmc_fetch.txt
python3 tools/pmc_summary.py gpurun_out/final_pmc_WRITE_SIZE > gpurun_out/final_pmc_write.txt
