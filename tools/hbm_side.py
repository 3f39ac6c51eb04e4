"""round 4: the rest of SURVEY 8(d)'s HBM list at the bench sizes (bench.roofline_hbm_side: Pendulum / MountainCar env-step at
2^24 envs, Adam / Polyak at 2^22 and 2^26 parameters, the max-pool push of 4096 x 28 KB frames, the 2^20-sample small gather)
for the rocprofv3 PMC traffic passes (tools/r4_pmc.sh): every kernel is launched >= 10 times, the last launches are averaged"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
import bench
out = bench.roofline_hbm_side(torch, rlhip)
print({k: (v["us_per_launch"], v["frac"]) for k, v in out.items()})
