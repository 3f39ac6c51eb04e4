"""CPU parity oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (reinforcementlearning.jl_amd/) never imports it and has no CPU fallback.
"""
from .binding import *  # noqa: F401,F403
from .agent import DQNRun, dqn_run  # noqa: F401,E402  (the per-stage DQN agent loop sequenced from the functions above)
