#!/bin/bash
# round 4, contact J: Float64 trig of Float32 arguments on a medium-range reduction -- exhaustive micro check, env parity, timings
export PYTHONPATH=$GRAFT_REPO_ROOT/reinforcementlearning.jl_amd:$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_j; mkdir -p $O
cd $GRAFT_REPO_ROOT
( time ./tools/micro/trig_f32arg.bin ) 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_learners.py tests/test_gpu_run.py tests/test_gpu_ppo3.py tests/test_gpu_ppo3w.py -q -x -m gpu > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
python tools/pendulum_pitch_ab.py 300 2>&1 | tail -2
python tools/envkinds_time.py 2>&1 | tail -6
