#!/bin/bash
# SQ counter passes over tools/convp_ab.py (round 6: the planes-in-LDS direct kernel against the one it replaces); run on the
# GPU box from the repo root.  Summary -> gpurun_out/sq_convp_<0|1>.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for V in ${CONVP_SET:-0 1}; do
  export SPA_CONVP=$V
  rm -rf gpurun_out/sqd_*
  i=0
  for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16"; do
    i=$((i+1)); D=gpurun_out/sqd_$i
    timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 tools/convp_ab.py --big > $D.log 2>&1
  done
  python3 tools/pmc_summary.py gpurun_out/ > gpurun_out/sq_all.txt
  grep -E "^(void )?(k_conv3x3_p16|k_conv3x3_f32)" gpurun_out/sq_all.txt > gpurun_out/sq_convp_$V.txt
  python3 tools/sq_table.py gpurun_out/sq_convp_$V.txt > gpurun_out/sq_convp_table_$V.txt
  rm -rf gpurun_out/sqd_*/
done
cat gpurun_out/sq_convp_table_*.txt
