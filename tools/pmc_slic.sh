#!/bin/bash
# PMC passes over tools/prof_stages.py (run on the GPU box from the repo root); summary of the SLIC kernels
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_*
i=0
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  i=$((i+1)); D=gpurun_out/pmc_$i
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 tools/prof_stages.py --batch ${BATCH:-30} --reps 2 > $D.log 2>&1
done
python3 tools/pmc_summary.py gpurun_out ${1:-k_slic} > gpurun_out/pmc_summary.txt
