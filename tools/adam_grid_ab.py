"""Adam / Polyak streaming kernels of bench.roofline_hbm_side alone (round 4: grid-size A / B through a temporary environment hook; the result is in the comment at the launches in csrc/optim.hip)"""
import os, sys
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"reinforcementlearning.jl_amd")]
import torch, rlhip, bench
out = bench.roofline_hbm_side(torch, rlhip)
print({k:(v["us_per_launch"], v["frac"]) for k,v in out.items() if "adam" in k or "polyak" in k})
