"""Thin functional layer over the C ABI: torch device tensors in, torch device tensors out.

Every function here is a direct call into librlhip.so (HIP kernels); torch only owns the memory and
the stream.  Names follow the reference (RLCore/src/utils/basic.jl, .../distributions.jl, ...).
Matrix arguments follow Julia's column-major convention: pass tensors whose *storage* is the
column-major matrix, i.e. a torch tensor of shape (n2, n1) C-contiguous == Julia (n1, n2).  The
helpers `from_julia` / `to_julia` convert from/to the mathematical (n1, n2) view.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import call

_DT = {torch.float32: "f32", torch.float64: "f64"}


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.RLHipError("rlhip ops need CUDA/HIP device tensors (there is no CPU fallback)")
    if not t.is_contiguous():
        raise _lib.RLHipArgumentError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def from_julia(a, dtype=None, device="cuda"):
    """(n1, n2) mathematical matrix -> tensor holding its column-major storage (shape (n2, n1))."""
    t = torch.as_tensor(a, dtype=dtype, device=device)
    return t.t().contiguous() if t.dim() == 2 else t.contiguous()


def to_julia(t):
    return t.t() if t.dim() == 2 else t


def _dims_shape(t):
    """tensor holding column-major storage of an n1 x n2 matrix (shape (n2, n1)) or a vector."""
    if t.dim() == 1:
        return t.shape[0], 1
    if t.dim() == 2:
        return t.shape[1], t.shape[0]
    raise _lib.RLHipArgumentError("expected a vector or a matrix")


def _u8(t):
    if t is None:
        return None
    if t.dtype == torch.bool:
        t = t.view(torch.uint8)
    if t.dtype != torch.uint8:
        raise _lib.RLHipArgumentError("terminal / mask must be bool or uint8")
    return t


# --------------------------------------------------------------------------------------- scans
def discount_rewards(rewards, gamma, terminal=None, init=None, dims=0):
    """discount_rewards(rewards, gamma; terminal, init, dims)  RLCore/src/utils/basic.jl:138-235."""
    sfx = _DT[rewards.dtype]
    n1, n2 = _dims_shape(rewards)
    out = torch.empty_like(rewards)
    if init is not None and not torch.is_tensor(init):
        init = torch.tensor([init], dtype=rewards.dtype, device=rewards.device)
    call(f"rlhip_discount_rewards_{sfx}", ptr(out), ptr(rewards), n1, n2, gamma, ptr(_u8(terminal)),
         ptr(init), dims, stream_ptr())
    return out


def discount_rewards_reduced(rewards, gamma, terminal=None, init=None, dims=0):
    """discount_rewards_reduced  RLCore/src/utils/basic.jl:237-319."""
    sfx = _DT[rewards.dtype]
    n1, n2 = _dims_shape(rewards)
    if rewards.dim() == 2 and dims not in (1, 2):
        raise _lib.RLHipArgumentError("matrix input requires dims = 1 or 2")
    n_out = 1 if rewards.dim() == 1 else (n2 if dims == 1 else n1)
    out = torch.empty(n_out, dtype=rewards.dtype, device=rewards.device)
    if init is not None and not torch.is_tensor(init):
        init = torch.tensor([init], dtype=rewards.dtype, device=rewards.device)
    call(f"rlhip_discount_rewards_reduced_{sfx}", ptr(out), ptr(rewards), n1, n2, gamma,
         ptr(_u8(terminal)), ptr(init), dims, stream_ptr())
    return out


def generalized_advantage_estimation(rewards, values, gamma, lam, terminal=None, dims=0):
    """generalized_advantage_estimation  RLCore/src/utils/basic.jl:334-417."""
    sfx = _DT[rewards.dtype]
    n1, n2 = _dims_shape(rewards)
    out = torch.empty_like(rewards)
    call(f"rlhip_gae_{sfx}", ptr(out), ptr(rewards), ptr(values), n1, n2, gamma, lam,
         ptr(_u8(terminal)), dims, stream_ptr())
    return out


def gae_returns(rewards, values, terminal, gamma, lam):
    """PPO fusion on time-major (T, n) tensors: returns (advantages, returns)."""
    T, n = rewards.shape
    adv = torch.empty_like(rewards)
    ret = torch.empty_like(rewards)
    call("rlhip_gae_returns_f32", ptr(adv), ptr(ret), ptr(rewards), ptr(values), ptr(_u8(terminal)), n, T,
         gamma, lam, stream_ptr())
    return adv, ret


# ----------------------------------------------------------------------------------- selection
def get_eps(kind, eps_stable, eps_init, warmup_steps, decay_steps, step):
    return _lib.lib.rlhip_get_eps({"linear": 0, "exp": 1}[kind], eps_stable, eps_init, warmup_steps,
                                  decay_steps, step)


def eps_greedy_select(values, eps, seed, step, env_id_base=0, mask=None, is_break_tie=False, soa=True):
    """values: (na, n) tensor.  soa=True: C-contiguous (na, n) (component-major, this library's layout);
    soa=False: storage of a Julia (na, N) column-major matrix, i.e. torch shape (n, na)."""
    if soa:
        na, n = values.shape
        ks, is_ = n, 1
    else:
        n, na = values.shape
        ks, is_ = 1, na
    out = torch.empty(n, dtype=torch.int32, device=values.device)
    call("rlhip_eps_greedy_select_f32", ptr(values), na, n, ks, is_, ptr(_u8(mask)), float(eps),
         int(is_break_tie), seed, env_id_base, step, ptr(out), stream_ptr())
    return out


def eps_greedy_prob(values, eps, mask=None, is_break_tie=False, soa=True):
    """prob(::EpsilonGreedyExplorer, values[, mask]) for every env of a (na, n) device tensor: Float64 probabilities in the
    layout of `values` (epsilon_greedy_explorer.jl:141-194)."""
    if soa:
        na, n = values.shape
        ks, is_ = n, 1
    else:
        n, na = values.shape
        ks, is_ = 1, na
    out = torch.empty(values.shape, dtype=torch.float64, device=values.device)
    m8 = _u8(mask)
    call("rlhip_eps_greedy_prob_f32", ptr(values), na, n, ks, is_, ptr(m8), float(eps), int(is_break_tie), ptr(out),
         stream_ptr())
    return out


def categorical_sample(logits, seed, step, env_id_base=0, mask=None, soa=True):
    if soa:
        na, n = logits.shape
        ks, is_ = n, 1
    else:
        n, na = logits.shape
        ks, is_ = 1, na
    a = torch.empty(n, dtype=torch.int32, device=logits.device)
    lp = torch.empty(n, dtype=torch.float32, device=logits.device)
    call("rlhip_categorical_sample_f32", ptr(logits), na, n, ks, is_, ptr(_u8(mask)), seed, env_id_base,
         step, ptr(a), ptr(lp), stream_ptr())
    return a, lp


# ------------------------------------------------------------------------------------- updates
def polyak_(dst, src, rho):
    call("rlhip_polyak_f32", ptr(dst), ptr(src), dst.numel(), rho, stream_ptr())
    return dst


def clip_by_global_norm_(grad, clip_norm):
    """clip_by_global_norm!(gs, ps, clip_norm)  RLCore/src/utils/basic.jl:21-29 -> device scalar gn."""
    gn = torch.empty(1, dtype=torch.float32, device=grad.device)
    call("rlhip_clip_by_global_norm_f32", ptr(grad), grad.numel(), clip_norm, ptr(gn), stream_ptr())
    return gn


def adam_(params, grad, m, v, beta_pow, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    call("rlhip_adam_f32", ptr(params), ptr(grad), ptr(m), ptr(v), ptr(beta_pow), params.numel(), lr, beta1,
         beta2, eps, stream_ptr())


def clip_adam_(params, grad, m, v, beta_pow, grad_scale=1.0, clip_norm=0.0, lr=1e-3, beta1=0.9,
               beta2=0.999, eps=1e-8, gn_out=None):
    call("rlhip_clip_adam_f32", ptr(params), ptr(grad), ptr(m), ptr(v), ptr(beta_pow), params.numel(),
         grad_scale, clip_norm, lr, beta1, beta2, eps, ptr(gn_out), stream_ptr())


def normlogpdf(mu, sigma, x):
    out = torch.empty_like(x)
    call("rlhip_normlogpdf_f32", ptr(mu), ptr(sigma), ptr(x), ptr(out), x.numel(), stream_ptr())
    return out


def diagnormlogpdf(mu, sigma, x):
    """arrays are storages of Julia (d, n) matrices: torch shape (n, d)."""
    n, d = mu.shape
    out = torch.empty(n, dtype=torch.float32, device=mu.device)
    call("rlhip_diagnormlogpdf_f32", ptr(mu), ptr(sigma), ptr(x), d, n, ptr(out), stream_ptr())
    return out


def huber_loss(q, target, delta=1.0, with_grad=True):
    loss = torch.empty(1, dtype=torch.float32, device=q.device)
    dq = torch.empty_like(q) if with_grad else None
    call("rlhip_huber_f32", ptr(q), ptr(target), q.numel(), delta, ptr(loss), ptr(dq), stream_ptr())
    return loss, dq


def td_target(qt_next, reward, terminal, gamma):
    """qt_next: SoA (na, n)."""
    na, n = qt_next.shape
    out = torch.empty(n, dtype=torch.float32, device=qt_next.device)
    call("rlhip_td_target_f32", ptr(qt_next), na, n, n, 1, ptr(reward), ptr(_u8(terminal)), gamma, ptr(out),
         stream_ptr())
    return out


# ----------------------------------------------------------------------------------------- misc
def fill_uniform(n, seed, t, tag, device="cuda"):
    out = torch.empty(n, dtype=torch.float32, device=device)
    call("rlhip_fill_uniform_f32", ptr(out), n, seed, t, tag, stream_ptr())
    return out


def permutation(n, seed, epoch, device="cuda"):
    out = torch.empty(n, dtype=torch.int32, device=device)
    call("rlhip_permutation", ptr(out), n, seed, epoch, stream_ptr())
    return out


def mlp2_nparams(n_in, h, n_out):
    return int(_lib.lib.rlhip_mlp2_nparams(n_in, h, n_out))


def mlp2_init(n_in, h, n_out, seed, net_id, device="cuda"):
    p = torch.empty(mlp2_nparams(n_in, h, n_out), dtype=torch.float32, device=device)
    call("rlhip_mlp2_init_f32", ptr(p), n_in, h, n_out, seed, net_id, stream_ptr())
    return p


def mlp2_forward(params, n_in, h, n_out, act, x):
    """x: SoA (n_in, batch) -> (n_out, batch)."""
    batch = x.shape[1]
    out = torch.empty((n_out, batch), dtype=torch.float32, device=x.device)
    call("rlhip_mlp2_forward_f32", ptr(params), n_in, h, n_out, act, ptr(x), batch, ptr(out), stream_ptr())
    return out


