#!/bin/bash
# round 4, contact F: PMC of the two-wave rollout kernel (two passes), Pendulum T = 128 rollout time
export PYTHONPATH=$GRAFT_REPO_ROOT/reinforcementlearning.jl_amd:$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
bash tools/pmc.sh r4roll_a tools/rollout_one.py SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
bash tools/pmc.sh r4roll_b tools/rollout_one.py SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAVES
