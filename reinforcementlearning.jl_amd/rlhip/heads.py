"""Stochastic Gaussian policy heads (host-side mirror).

Reference: RLCore/src/utils/networks.jl -- `GaussianNetwork(pre, μ, σ; min_σ, max_σ, squash)` :44-116 and
`SoftGaussianNetwork(pre, μ, σ; min_σ, max_σ)` :130-199.  `pre`, `μ`, `σ` are callables on device tensors
(`HipApproximator.forward`, `ops.mlp2_forward`, ...) that return (features, batch) matrices in the reference's
column-major sense, i.e. torch tensors of shape (batch, features) whose memory is (d x n) column-major; the head itself
-- clamp, sampling, squash, log-probability with the tanh correction -- is one launch (csrc/heads.hip).  `rng` is the
shared Philox NORMAL stream: (seed, env_id_base + column, step) with `step` advancing by one per sampling call."""
import torch

from ._lib import call
from .ops import ptr, stream_ptr


def _cm(x):
    """(d, n) Julia-shaped view -> contiguous (n, d) storage (= d x n column-major)"""
    return x.t().contiguous() if x.dim() == 2 else x.contiguous()


class GaussianNetwork:
    soft = 0

    def __init__(self, pre=None, mu=None, sigma=None, min_sigma=0.0, max_sigma=float("inf"), squash="identity", seed=0,
                 env_id_base=0):
        if squash not in ("identity", "tanh"):
            raise ValueError("squash must be identity or tanh")  # "Other squashing functions are not supported" :36
        self.pre = pre if pre is not None else (lambda x: x)
        self.mu, self.sigma = mu, sigma
        self.min_sigma, self.max_sigma = float(min_sigma), float(max_sigma)
        self.squash = 1 if squash == "tanh" else 0
        self.seed, self.env_id_base, self.step = int(seed), int(env_id_base), 0

    # -- (model)(state): mu, sigma (d, n)
    def heads(self, state):
        x = self.pre(state)
        mu, raw = self.mu(x), self.sigma(x)
        return mu, raw

    def __call__(self, state, action=None, is_sampling=False, is_return_log_prob=False):
        if action is not None:
            return self.logp(state, action)
        mu, raw = self.heads(state)
        if not is_sampling:
            return mu, raw.clamp(self.min_sigma, self.max_sigma)  # :65-67,79
        a, lp = self._sample(mu, raw, 1, is_return_log_prob)
        a = a[:, 0, :].t()  # (d, n)
        return (a, lp.reshape(1, -1)) if is_return_log_prob else a

    def sample(self, state, action_samples):
        """(model)(state::(ns, 1, n), action_samples::Int) -> actions (d, K, n), logp (1, K, n)  :90-100"""
        if state.dim() == 3:
            state = state[:, 0, :]
        mu, raw = self.heads(state)
        a, lp = self._sample(mu, raw, int(action_samples), True)
        return a.permute(2, 1, 0), lp.t().unsqueeze(0)

    def logp(self, state, action):
        """(model)(state, action) -> logp (1, n) or (1, K, n)  :110-116"""
        three = action.dim() == 3
        if state.dim() == 3:
            state = state[:, 0, :]
        mu, raw = self.heads(state)
        d, n = mu.shape
        act = action.permute(2, 1, 0).contiguous() if three else action.t().contiguous().reshape(n, 1, d)
        K = act.shape[1]
        out = torch.empty((n, K), dtype=torch.float32, device=mu.device)
        mu_s, raw_s = _cm(mu), _cm(raw)  # keep the transposed copies alive across the (asynchronous) call
        call("rlhip_gaussian_head_logp_f32", ptr(mu_s), ptr(raw_s), ptr(act), d, n, K, self.min_sigma,
             self.max_sigma, self.squash, self.soft, ptr(out), stream_ptr())
        return out.t().unsqueeze(0) if three else out.reshape(1, n)

    def _sample(self, mu, raw, K, want_logp):
        d, n = mu.shape
        act = torch.empty((n, K, d), dtype=torch.float32, device=mu.device)
        lp = torch.empty((n, K), dtype=torch.float32, device=mu.device) if want_logp else None
        mu_s, raw_s = _cm(mu), _cm(raw)  # keep the transposed copies alive across the (asynchronous) call
        call("rlhip_gaussian_head_sample_f32", ptr(mu_s), ptr(raw_s), d, n, K, self.min_sigma, self.max_sigma,
             self.squash, self.soft, self.seed, self.env_id_base, self.step, ptr(act), ptr(lp), stream_ptr())
        self.step += 1
        return act, lp


class SoftGaussianNetwork(GaussianNetwork):
    """SoftGaussianNetwork: tanh-squashed actions, logp = sum(normlogpdf - 2 (log 2 - z - softplus(-2 z)))  :147-198"""
    soft = 1

    def __init__(self, pre=None, mu=None, sigma=None, min_sigma=0.0, max_sigma=float("inf"), seed=0, env_id_base=0):
        super().__init__(pre, mu, sigma, min_sigma, max_sigma, "tanh", seed, env_id_base)


class CategoricalNetwork:
    """CategoricalNetwork(model)  RLCore/src/utils/networks.jl:405-432 (+ masked methods :459-472).

    `model` maps a state batch to logits shaped (na, n).  `(net)(state; is_sampling, is_return_log_prob)` returns the
    logits, or a one-hot `z` (na, n) drawn with the Gumbel-max trick (`sample_categorical` :425-432, one launch:
    rlhip_categorical_sample_f32 on the GUMBEL Philox stream), or `(z, logits)`; with a Bool `mask` (na, n) the masked
    logits are `logits + ifelse(mask, 0, typemin)` (:461) and masked actions are never drawn.  `actions` holds the
    0-based indices of the last draw (what the env kernels take)."""

    def __init__(self, model, seed=0, env_id_base=0):
        self.model, self.seed, self.env_id_base, self.step = model, int(seed), int(env_id_base), 0
        self.actions = None

    def __call__(self, state, mask=None, is_sampling=False, is_return_log_prob=False):
        from ._lib import call
        from .ops import _u8, ptr, stream_ptr

        lg = self.model(state).contiguous()
        na, n = lg.shape
        m8 = _u8(mask)
        logits = torch.empty_like(lg) if mask is not None else lg
        z = torch.empty_like(lg) if is_sampling else None
        if is_sampling:
            self.actions = torch.empty(n, dtype=torch.int32, device=lg.device)
        if mask is not None or is_sampling:  # masked logits, Gumbel-max draw and one-hot in ONE launch behind the ABI
            call("rlhip_categorical_network_f32", ptr(lg), na, n, ptr(m8), self.seed, self.env_id_base, self.step,
                 ptr(logits) if mask is not None else None, ptr(self.actions) if is_sampling else None, ptr(z),
                 stream_ptr())
        if not is_sampling:
            return logits
        self.step += 1
        return (z, logits) if is_return_log_prob else z
