"""The oracle against its own frozen outputs for the parts without a reference KAT (tests/golden/oracle_pins/): the
Acrobot RK4 step and the Gaussian heads.  Transcendentals come from the host libm, so values are compared to 1 ulp-ish
tolerances, flags exactly."""
import json
import os

import numpy as np

import oracle

PINS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_pins", "pins.json")))


def unhx(xs, dtype=np.float64):
    return np.array([float.fromhex(x) for x in xs], np.float64).astype(dtype)


def test_acrobot_oracle_matches_its_pins():
    for case in PINS["acrobot"]:
        dt = np.float64 if case["dtype"] == "f64" else np.float32
        env = oracle.VecEnv("acrobot", 6, seed=21, env_id_base=3, dtype=dt, **case["kw"])
        for a, want in zip(case["actions"], case["steps"]):
            env.step(np.array(a, np.int32))
            for k in range(4):
                np.testing.assert_allclose(env.s[k], unhx(want["s"][k], dt), rtol=1e-12 if dt == np.float64 else 1e-6, atol=0)
            assert env.done.tolist() == want["done"] and np.array_equal(env.reward, unhx(want["reward"], dt))


def test_gaussian_heads_oracle_matches_its_pins():
    h = PINS["heads"]
    d, n = h["shape"]
    mu, raw = unhx(h["mu"], np.float32).reshape(d, n), unhx(h["raw_sigma"], np.float32).reshape(d, n)
    for c in h["cases"]:
        a, lp = oracle.gaussian_head_sample(mu, raw, 2, 0.2, 1.5, c["squash"], c["soft"], seed=9, env_id_base=1, step=4)
        np.testing.assert_allclose(a.ravel(), unhx(c["action"], np.float32), rtol=2e-7, atol=0)
        np.testing.assert_allclose(lp.ravel(), unhx(c["logp"], np.float32), rtol=1e-5, atol=1e-6)
        lp2 = oracle.gaussian_head_logp(mu, raw, a, 0.2, 1.5, c["squash"], c["soft"])
        np.testing.assert_allclose(lp2.ravel(), unhx(c["logp_of_action"], np.float32), rtol=1e-5, atol=1e-5)
