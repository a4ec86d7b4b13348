#!/bin/bash
# SQ counters of the fused top-K kernel (GPU box, repo root): bash scripts/topk_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/topk_pmc; mkdir -p $R/gpurun_out/topk_pmc
python3 $R/scripts/topk_only.py yelp2018 5 > $R/gpurun_out/topk_pmc/time_masked.txt 2>&1
i=0
for ctr in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/topk_pmc/p$i -o p -- python3 $R/scripts/topk_only.py yelp2018 2 > $R/gpurun_out/topk_pmc/p$i.log 2>&1 || echo "pass $i failed"
done
cat $R/gpurun_out/topk_pmc/time_*.txt | grep evaluation
