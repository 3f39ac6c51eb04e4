#!/bin/bash
# round 6 contact U: LDS and MFMA counters of the 256-wide step's kernels at HEAD, with the backward kernel's unpadded (default) and padded LDS copy
# (RLHIP_W3_DZF_PAD=0 / 1): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, SQ_INSTS_LDS / SQ_ACTIVE_INST_LDS, SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES,
# SQ_INSTS_VALU / SQ_WAVE_CYCLES -- separate passes, kernel-trace only
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
R=$PWD; O=gpurun_out/r6_u; mkdir -p $O; rm -f $O/*.txt
for pad in 0 1; do
  export RLHIP_W3_DZF_PAD=$pad
  i=0
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/p -o pmc -- python $R/tools/ppo3w_time.py 4096 128 3 > $R/$O/log_${pad}_$i.log 2>&1)
    for k in ppo3w_fwd_kernel ppo3w_bwd_kernel ppo3w_dw2_kernel; do python3 tools/pmc_last.py $O/p $k 32 2>/dev/null | grep -v "0, 4>" | sed "s/^/pad=$pad /" | tee -a $O/counters.txt; done
    rm -rf $O/p
  done
done
