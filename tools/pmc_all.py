"""every HBM-bound kernel of the bench line at the bench's sizes, for the PMC traffic passes of tools/pmc_all.sh
(bench.roofline_extras(hbm_only = True): GAE + returns, the u8 frame gathers; bench.roofline_hbm_side: Pendulum / MountainCar
env-step, Adam / Polyak, the max-pool push, the small gather).  The CartPole env-step has its own script (tools/envstep.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
import bench
out = bench.roofline_extras(torch, rlhip, hbm_only=True)
out.update(bench.roofline_hbm_side(torch, rlhip))
print({k: (v.get("us_per_launch"), v.get("frac"), v.get("algorithmic_bytes")) for k, v in out.items() if isinstance(v, dict)})
