"""The remaining explorers of RLCore/src/policies/explorers on (na, N) device value matrices -- host mirror; the
selection itself is select.hip behind the C ABI.  Every class is already "batched": one column per env instance,
which is what the reference's BatchExplorer (batch_explorer.jl:14-21) does with a comprehension.

    WeightedExplorer          weighted_explorer.jl:19-45
    WeightedSoftmaxExplorer   weighted_softmax_explorer.jl:13-34
    GumbelSoftmaxExplorer     gumbel_softmax_explorer.jl:6-24
    UCBExplorer               UCB_explorer.jl:5-30
    BatchExplorer             batch_explorer.jl:6-21
"""
import ctypes as C

import torch

from ._lib import RLHipError, call
from .ops import ptr, stream_ptr


def _strided_ptr(t):
    """base pointer of a possibly strided (na, N) view: the selection kernels take both strides"""
    if not t.is_cuda:
        raise RLHipError("rlhip ops need CUDA/HIP device tensors (there is no CPU fallback)")
    return C.c_void_p(t.data_ptr())


class _SamplingExplorer:
    KIND = None

    def __init__(self, seed=0, step=1, is_normalized=False):
        self.seed, self.step, self.is_normalized = int(seed), int(step), bool(is_normalized)

    def plan_(self, values, mask=None, env_id_base=0):
        """plan!(s, values[, mask]) for a (na, N) float32 device tensor -> 1-based actions (N,)"""
        if values.dtype != torch.float32 or values.dim() != 2:
            raise TypeError("values must be a (na, N) float32 matrix")
        na, n = values.shape
        out = torch.empty(n, dtype=torch.int32, device=values.device)
        if mask is not None:
            mask = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
        if mask is not None and mask.stride() != values.stride():
            mask = mask.contiguous()
            values = values.contiguous()
        call("rlhip_explorer_select_f32", self.KIND, _strided_ptr(values), na, n, values.stride(0), values.stride(1),
             None if mask is None else _strided_ptr(mask), int(self.is_normalized), self.seed, env_id_base, self.step, ptr(out),
             stream_ptr())
        self.step += 1
        return out + 1


class WeightedExplorer(_SamplingExplorer):
    """WeightedExplorer(; is_normalized = false, rng): sample(rng, Weights(values)); elements assumed >= 0."""
    KIND = 0

    def __init__(self, is_normalized=False, seed=0):
        super().__init__(seed, 1, is_normalized)


class WeightedSoftmaxExplorer(_SamplingExplorer):
    """WeightedSoftmaxExplorer(; rng): sample(rng, Weights(softmax(values), 1))."""
    KIND = 1


class GumbelSoftmaxExplorer(_SamplingExplorer):
    """GumbelSoftmaxExplorer(; rng): argmax(logsoftmax(v) .- log.(-log.(rand(rng, T, n))))."""
    KIND = 2


class UCBExplorer:
    """UCBExplorer(na; c = 2.0, ϵ = 1e-10, step = 1): one set of action counts per env instance."""

    def __init__(self, na, n_env=1, c=2.0, eps=1e-10, seed=0, device="cuda"):
        self.c, self.step, self.seed = float(c), 1, int(seed)
        self.actioncounts = torch.full((na, n_env), eps, dtype=torch.float64, device=device)

    def plan_(self, values, env_id_base=0):
        na, n = values.shape
        if (na, n) != tuple(self.actioncounts.shape):
            raise ValueError("values shape does not match the action counts")
        out = torch.empty(n, dtype=torch.int32, device=values.device)
        call("rlhip_ucb_select_f32", _strided_ptr(values), na, n, values.stride(0), values.stride(1), self.c,
             ptr(self.actioncounts), self.step, self.seed, env_id_base, ptr(out), stream_ptr())
        self.step += 1
        return out + 1


class BatchExplorer:
    """BatchExplorer(explorer): every explorer here already maps over the columns of a value matrix."""

    def __init__(self, explorer):
        self.explorer = explorer

    def plan_(self, values, mask=None, **kw):
        if values.dim() == 1:
            values = values[:, None]
            mask = None if mask is None else mask[:, None]
        return self.explorer.plan_(values, **kw) if mask is None else self.explorer.plan_(values, mask, **kw)
