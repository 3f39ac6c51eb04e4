"""the HBM side kernels of bench.roofline_hbm_side alone (round 4 A / B of store policy / unroll: RLHIP_STREAM_NT_STORES,
RLHIP_STREAM_UNROLL2, RLHIP_GATHER_GENERIC are read once per process)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
import bench
out = bench.roofline_hbm_side(torch, rlhip)
tag = " ".join(f"{k}={os.environ[k]}" for k in ("RLHIP_STREAM_NT_STORES", "RLHIP_STREAM_UNROLL2", "RLHIP_GATHER_GENERIC") if k in os.environ) or "default"
print(tag, {k: (v["us_per_launch"], v["frac"]) for k, v in out.items()})
