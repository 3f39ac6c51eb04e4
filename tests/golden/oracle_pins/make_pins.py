#!/usr/bin/env python3
"""Regression pins of the parts of the oracle whose parity is UNPINNED upstream (no reference KAT exists): the Acrobot
RK4 step and the Gaussian policy heads.  NOT reference-derived -- these freeze what the oracle computes today (values as
C99 hex floats), so that a later edit of oracle/*.c or of the shared Philox streams cannot drift silently; the GPU suite
checks the kernels against the same numbers.  Regenerate only on purpose:  python tests/golden/oracle_pins/make_pins.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", ".."))
import oracle  # noqa: E402


def hx(a):
    return [float(x).hex() for x in np.asarray(a, np.float64).ravel()]


def acrobot():
    out = []
    for dtype, name in ((np.float64, "f64"), (np.float32, "f32")):
        for kw in ({}, {"nips": 1}, {"max_torque_noise": 0.5}):
            env = oracle.VecEnv("acrobot", 6, seed=21, env_id_base=3, dtype=dtype, **kw)
            acts = [[0, 1, 2, 2, 1, 0], [2, 2, 0, 1, 0, 1], [1, 0, 2, 0, 2, 1]]
            steps = []
            for a in acts:
                env.step(np.array(a, np.int32))
                steps.append({"s": [hx(x) for x in env.s], "reward": hx(env.reward), "done": env.done.tolist()})
            out.append({"dtype": name, "kw": kw, "actions": acts, "steps": steps})
    return out


def heads():
    rng = np.random.default_rng(5)
    mu = rng.normal(size=(3, 4)).astype(np.float32)
    raw = np.log1p(np.exp(rng.normal(size=(3, 4)))).astype(np.float32)
    out = {"mu": hx(mu), "raw_sigma": hx(raw), "shape": [3, 4], "cases": []}
    for squash, soft in ((0, 0), (1, 0), (1, 1)):
        a, lp = oracle.gaussian_head_sample(mu, raw, 2, 0.2, 1.5, squash, soft, seed=9, env_id_base=1, step=4)
        out["cases"].append({"squash": squash, "soft": soft, "action": hx(a), "logp": hx(lp),
                             "logp_of_action": hx(oracle.gaussian_head_logp(mu, raw, a, 0.2, 1.5, squash, soft))})
    return out


if __name__ == "__main__":
    json.dump({"acrobot": acrobot(), "heads": heads()}, open(os.path.join(HERE, "pins.json"), "w"), indent=1)
    print("wrote", os.path.join(HERE, "pins.json"))
