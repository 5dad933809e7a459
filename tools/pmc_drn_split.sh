#!/bin/bash
# SQ counter passes over the split-plane DRN kernels (run on the GPU box from the repo root): the Winograd layer of
# tools/prof_stages.py --wino (k_wino4_in, k_gemm_f16x3, k_wino4_out_s), its direct 128 -> 128 layer, and the stem + the
# remaining kernels through a two-step bench.  Summary -> gpurun_out/sq_drn_split.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/sqd_*
i=0
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  i=$((i+1)); D=gpurun_out/sqd_$i
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_exact_fp32 --one_stream > $D.log 2>&1
done
python3 tools/pmc_summary.py gpurun_out/ > gpurun_out/sq_all.txt
grep -E "^(void )?(k_gemm_f16x3|k_conv3x3_bf16|k_wino4_in|k_wino4_out_s|k_conv3x3_f32|k_drn_stem_d_f16x3|k_bias_act_f32|k_amax)" gpurun_out/sq_all.txt > gpurun_out/sq_drn_split.txt
rm -rf gpurun_out/sqd_*/
wc -l gpurun_out/sq_drn_split.txt
