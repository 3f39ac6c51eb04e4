"""round 4 A / B: store policy of the env-step kernel's WRITE-ONLY arrays (reward, done, observation planes) at 2^24 envs --
measured through a temporary environment hook; the result is the comment at the streaming launch in csrc/envs.hip"""
import os, sys
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"reinforcementlearning.jl_amd")]
import torch, rlhip, bench
r = bench.roofline_env_step(torch, rlhip)
side = bench.roofline_hbm_side(torch, rlhip)
print(os.environ.get("RLHIP_ENV_WO_ORDINARY", "0"), "cartpole", r["us_per_launch"], r["frac"], "no-term", r["without_terminations"]["us_per_launch"],
      {k: (v["us_per_launch"], v["frac"]) for k, v in side.items() if k.startswith("env_step")})
