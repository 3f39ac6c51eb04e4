#!/usr/bin/env python3
"""Build librlhip.so (the C-ABI shared library) for gfx950 with hipcc, in-tree.

    python reinforcementlearning.jl_amd/build.py [--force]

Output: reinforcementlearning.jl_amd/lib/librlhip.so  (git-ignored, travels with gpurun snapshots).
Flags: -ffp-contract=off so the parity kernels never fuse a*b+c behind the reference's back
(MLP code uses explicit fmaf); gfx950 only -- no other offload arch, no compatibility layers.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "librlhip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result"]
# per-file additions.  ppo3.hip: the SLP vectorizer pairs adjacent f32 accumulations of the learner tiles into v_pk_fma_f32;
# beside MFMAs a packed f32 op costs more than the two scalar ones it replaces (MI355X_MICROARCH.md), the register pairs
# it needs pushed the 256-register producer / consumer tile into scratch, and one op_sel form of it produced run-to-run
# different dW1 sums at two waves per SIMD (tools/ppo3_determinism.py) -- scalar f32 code is exact, smaller and faster here
EXTRA = {"ppo3.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "rlhip.h"))
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    if _stale(obj, [os.path.join(CSRC, src)] + headers()):
        cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + os.environ.get("RLHIP_EXTRA_FLAGS", "").split() + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(_compile, sources()))
    if _stale(SO, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
