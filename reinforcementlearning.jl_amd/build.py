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
# per-file additions.  Every source whose kernels issue MFMAs is built WITHOUT the SLP vectorizer: it pairs adjacent f32
# accumulations into v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, and beside MFMAs a packed f32 op costs more than the two
# scalar ones it replaces (MI355X_MICROARCH.md), the even-aligned register pairs pushed a 256-register tile into scratch,
# and three wrong-result sightings in rounds 1-2 (a deterministic one in dqn3.hip, a deterministic one and a run-to-run
# one in the MFMA PPO tiles: DESIGN.md section 5) all sat on SLP-packed ops beside MFMAs -- root cause not established, so the
# combination is banned outright: tests/test_no_packed_f32_beside_mfma.py disassembles build/*.o and fails on any packed
# f32 VALU op inside a kernel that contains an MFMA.  (The two-layer PPO learner -- ppo_grad.hip -- runs
# layer 1 of both of its phases on the f32 MFMA since round 3 and falls under the same rule: its actor / critic pairs are
# two scalar FMAs, not one v_pk_fma_f32.)
NO_SLP = ["-fno-slp-vectorize", "-fno-vectorize"]  # (the loop vectorizer packs 2-trip loops over the actions the same way)
EXTRA = {"ppo3.hip": NO_SLP, "dqn3.hip": NO_SLP, "ppo3w.hip": NO_SLP, "ppo_grad.hip": NO_SLP}


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


# named variants of the library: (object directory, library file, extra flags).  "bounds" = the debug build of SURVEY.md section 5:
# every entry point that takes caller-supplied gather indices validates them first (rlhip_ring_check_indices; include/rlhip.h).
# Use it with RLHIP_LIB_PATH=<...>/lib/librlhip_bounds.so (rlhip/_lib.py) or point the Julia glue's `librlhip` at it.
VARIANTS = {"bounds": (os.path.join(HERE, "build_bounds"), os.path.join(LIBDIR, "librlhip_bounds.so"), ["-DRLHIP_BOUNDS_CHECK"]),
            # per-phase cycle sums of the two-wave PPO rollout, printed by workgroup 0 (a profiling aid: tools/rollout_one.py)
            "rollout_timing": (os.path.join(HERE, "build_rt"), os.path.join(LIBDIR, "librlhip_rt.so"), ["-DRLHIP_ROLLOUT_TIMING"])}


def _compile(src, obj_dir=None, more_flags=()):
    obj = os.path.join(obj_dir or OBJ, src.replace(".hip", ".o"))
    flags = FLAGS + EXTRA.get(src, []) + list(more_flags) + os.environ.get("RLHIP_EXTRA_FLAGS", "").split()
    stamp = obj + ".flags"  # an object is also stale when it was built with other flags
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
    if not same_flags or _stale(obj, [os.path.join(CSRC, src)] + headers()):
        cmd = [HIPCC] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
        with open(stamp, "w") as fh:
            fh.write(" ".join(flags))
    return obj


def build(force=False, variant=None):
    if variant is not None:
        return _build_variant(variant, force)
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    live = {f.replace(".hip", ".o") for f in sources()}
    for f in os.listdir(OBJ):  # objects of sources that no longer exist
        if (f.endswith(".o") and f not in live) or (f.endswith(".o.flags") and f[:-6] not in live):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(_compile, sources()))
    if _stale(SO, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return SO


def _build_variant(variant, force=False):
    """the default library's objects, except for the sources that look at the variant's macro: those are compiled again with it"""
    obj_dir, so, more = VARIANTS[variant]
    build(force=False)  # the shared objects
    os.makedirs(obj_dir, exist_ok=True)
    if force:
        for f in os.listdir(obj_dir):
            os.remove(os.path.join(obj_dir, f))
    macros = [m[2:].split("=")[0] for m in more if m.startswith("-D")] + ["RLHIP_CHECK_GATHER_INDICES"]

    def affected(src):
        text = open(os.path.join(CSRC, src)).read()
        return any(m in text for m in macros)

    objs = []
    for src in sources():
        objs.append(_compile(src, obj_dir, more) if affected(src) else os.path.join(OBJ, src.replace(".hip", ".o")))
    if _stale(so, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return so


if __name__ == "__main__":
    variants = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--variant=")]
    print(build(force="--force" in sys.argv, variant=variants[0] if variants else None))
