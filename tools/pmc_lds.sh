#!/bin/bash
# LDS bank-conflict counters of the DRN kernels (one rocprofv3 --pmc pass over a two-step bench) -> gpurun_out/sq_lds.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/sqd_*
D=gpurun_out/sqd_1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_host_loop > $D.log 2>&1
python3 tools/pmc_summary.py gpurun_out/ > gpurun_out/sq_all.txt
grep -E "^(void )?(k_gemm_f16x3|k_conv3x3_f32|k_drn_stem_d_f16x3|k_drn_layer2|k_conv_small)" gpurun_out/sq_all.txt | grep -E "SQ_LDS|SQ_ACTIVE_INST_LDS|SQ_INSTS_LDS" > gpurun_out/sq_lds.txt
rm -rf gpurun_out/sqd_*/
wc -l gpurun_out/sq_lds.txt
