#!/bin/bash
# PMC passes of the headline workload (tools/ppo2_one.py) for two builds of the library: gpurun_ab/libOLD.so (start of the tile
# work: packed FMA pairs, records through the scalar cache, counter barrier in the tail) and gpurun_ab/libNEW.so
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
L=reinforcementlearning.jl_amd/lib/librlhip.so
cp $L /tmp/lib_keep.so
for v in OLD NEW; do
    cp gpurun_ab/lib$v.so $L
    for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES"; do
        echo "== $v: $pass"
        bash tools/pmc.sh tile_$v tools/ppo2_one.py $pass 2>&1 | grep "ppo_grad_kernel\|reduce_apply_kernel" | cut -c1-400
    done
done
cp /tmp/lib_keep.so $L
