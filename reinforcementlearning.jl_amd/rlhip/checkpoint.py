"""Checkpoint export / import of everything the hot path keeps on the device.

Reference: no built-in checkpointing; the documented recipe is `JLD2.@save` of the policy's approximator from a
`DoEveryNSteps` hook (docs/src/How_to_use_hooks.md:122-167, test RLEnvs/test/environments/examples/random_walk_1d.jl
:78-118).  With the state in HBM that hook needs an export: `state_dict(obj)` walks an env / policy / learner / agent /
trajectory and returns a FLAT {"path/to/field": numpy array or scalar} dict -- parameters, Adam moments and running
beta powers, target network, env state and episode (RNG) counters, vec-step / update / sampler counters, ring-buffer
storage with its head / length fields, sum-tree priorities -- which is what JLD2 (or np.savez) stores as is.
`load_state_dict(obj, d)` copies it back IN PLACE (device pointers, captured graphs and C structs stay valid), after
which the run continues bit-identically (tests/test_gpu_run.py).  Device <-> host copies go through torch here; the
Julia glue uses rlhip_memcpy_d2h / rlhip_memcpy_h2d of the C ABI on the same buffers."""
import ctypes as C

import numpy as np
import torch

_SCALARS = (bool, int, float, str)


def _is_ours(v):
    return type(v).__module__.split(".")[0] in ("rlhip",) and not isinstance(v, type)


def _struct_fields(st):
    for name, ftype in st._fields_:
        if isinstance(ftype, type) and issubclass(ftype, C._SimpleCData) and ftype not in (C.c_void_p, C.c_char_p):
            yield name


def _walk(obj, prefix, visit, seen, skip):
    """calls visit(kind, path, owner, key, value) for every tensor / scalar / C-struct field reachable from obj"""
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, dict):
        items = [(str(k), obj, k, v) for k, v in obj.items()]
    elif isinstance(obj, (list, tuple)):
        items = [(str(i), obj, i, v) for i, v in enumerate(obj)]
    else:
        items = [(k, obj, k, v) for k, v in sorted(vars(obj).items())]
    for name, owner, key, v in items:
        path = f"{prefix}{name}"
        if any(s in path for s in skip) or name == "_backing":  # _backing: the allocation the env's arrays are views of
            continue
        if isinstance(v, torch.Tensor):
            visit("tensor", path, owner, key, v)
        elif isinstance(v, _SCALARS) or v is None:
            if not isinstance(owner, tuple) and v is not None:
                visit("scalar", path, owner, key, v)
        elif isinstance(v, C.Structure):
            for f in _struct_fields(v):
                visit("field", f"{path}.{f}", v, f, getattr(v, f))
        elif isinstance(v, (dict, list, tuple)) or _is_ours(v):
            _walk(v, path + "/", visit, seen, skip)
        # anything else (process groups, captured graphs, ctypes pointers, modules) is not state of the path


def state_dict(obj, skip=()):
    """Flat {"path": np.ndarray | scalar} snapshot of obj (an rlhip object, or a dict / list of them).
    skip: substrings of paths to leave out (e.g. ("trajectory/container/state",) for a 30 GB frame ring)."""
    out = {}

    def visit(kind, path, owner, key, v):
        out[path] = v.detach().cpu().numpy().copy() if kind == "tensor" else v

    if torch.cuda.is_available():
        torch.cuda.synchronize()
    _walk(obj, "", visit, set(), tuple(skip))
    return out


def load_state_dict(obj, d, skip=(), strict=True):
    """Copy a state_dict back into obj in place.  strict: every entry of obj must be present in d, with equal shape."""
    missing = []

    def visit(kind, path, owner, key, v):
        if path not in d:
            missing.append(path)
            return
        x = d[path]
        if kind == "tensor":
            x = torch.as_tensor(np.asarray(x))
            if tuple(x.shape) != tuple(v.shape) or x.dtype != v.dtype:
                raise ValueError(f"checkpoint entry {path}: {tuple(x.shape)} {x.dtype} does not fit {tuple(v.shape)} {v.dtype}")
            v.copy_(x)
        else:
            x = x.item() if isinstance(x, np.ndarray) else x
            x = type(v)(x) if isinstance(v, _SCALARS) else x
            if kind == "field":
                setattr(owner, key, x)
            elif isinstance(owner, (dict, list)):
                owner[key] = x
            else:
                setattr(owner, key, x)

    _walk(obj, "", visit, set(), tuple(skip))
    if strict and missing:
        raise KeyError(f"checkpoint lacks {len(missing)} entries, e.g. {missing[:3]}")
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return obj


def save_checkpoint(path, obj, skip=()):
    """np.savez of state_dict(obj): plain arrays + 0-d scalars under their path names (JLD2-friendly layout)."""
    d = state_dict(obj, skip)
    np.savez(path, **{k: np.asarray(v) for k, v in d.items()})
    return len(d)


def load_checkpoint(path, obj, skip=(), strict=True):
    with np.load(path, allow_pickle=False) as z:
        d = {k: z[k] for k in z.files}
    return load_state_dict(obj, d, skip, strict)
