"""The two-wave PPO rollout (csrc/ppo.hip rollout_split_kernel) is built on three resource facts: 512-thread workgroups whose
actor wave and critic wave share a SIMD need <= 256 registers per lane; a spilled register in the step loop is a scratch round
trip per vec-step on the dependent chain; and the step records + two noise chunks must fit one workgroup's LDS.  CPU only:
hipcc cross-compiles, llvm-readelf reads the kernel descriptors' metadata of the gfx950 code object."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "reinforcementlearning.jl_amd", "build", "ppo.o")
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernel_meta(obj, tmp):
    fat, co = os.path.join(tmp, "ppo.fatbin"), os.path.join(tmp, "ppo.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], check=True, capture_output=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={fat}", f"--output={co}"], check=True, capture_output=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out = {}
    for blk in notes.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        num = lambda key: int(re.search(key + r":\s+(\d+)", blk).group(1))  # noqa: E731
        out[name] = {"agpr": int(re.match(r":\s+(\d+)", blk).group(1)), "vgpr": num(r"\.vgpr_count"),
                     "vspill": num(r"\.vgpr_spill_count"), "lds": num(r"\.group_segment_fixed_size"),
                     "scratch": num(r"\.private_segment_fixed_size"), "wg": num(r"\.max_flat_workgroup_size")}
    return out


@pytest.mark.skipif(not os.path.exists(f"{LLVM}/llvm-readelf"), reason="no llvm-readelf")
def test_two_wave_rollout_fits_two_waves_per_simd_without_scratch(tmp_path):
    import __graft_entry__ as g

    g.build()
    meta = {k: v for k, v in _kernel_meta(OBJ, str(tmp_path)).items() if "rollout_split_kernel" in k}
    assert len(meta) >= 3 * 3 * 7, f"expected every env x width x head instantiation, found {len(meta)}"
    bad = []
    for name, m in meta.items():
        regs = m["vgpr"] + m["agpr"]
        # ACT = 0 and a head kind the env really has (CartPole: 2 actions / Gaussian; Pendulum, MountainCar: 3 actions / Gaussian --
        # the dispatcher also instantiates the combinations no env produces, e.g. a 3-action CartPole)
        m_ = re.search(r"INS_\d+(CartPole|Pendulum|MountainCar)ParamsIfEELi\d+ELi\d+ELi0ELi\dELi([123])E", name)
        relu_known_head = m_ is not None and int(m_.group(2)) in ({"CartPole": (1, 2)}.get(m_.group(1), (1, 3)))
        if m["wg"] != 512:
            bad.append(f"{name[:90]}: workgroup bound {m['wg']}")
        if regs > 256:
            bad.append(f"{name[:90]}: {regs} registers per lane (two waves per SIMD need <= 256)")
        if m["lds"] > 160 * 1024:
            bad.append(f"{name[:90]}: {m['lds']} bytes of LDS")
        if relu_known_head and (m["vspill"] or m["scratch"]):  # the headline family: nothing of the chain in scratch
            bad.append(f"{name[:90]}: {m['vspill']} spilled VGPRs, {m['scratch']} B of scratch")
    assert not bad, "\n".join(bad)
    # no second copy of the wide rollout in the library (the one-wave kernel was removed with the A / B that retired it)
    assert not [k for k in _kernel_meta(OBJ, str(tmp_path)) if "rollout_wide_kernel" in k]


@pytest.mark.skipif(not os.path.exists(f"{LLVM}/llvm-objdump"), reason="no llvm-objdump")
def test_two_wave_rollout_keeps_its_barrier_pairing(tmp_path):
    """rollout_split_kernel places its T + 2 workgroup barriers in two DIFFERENT branches (actor waves / critic
    waves: threadIdx.x >> 8) and relies on s_barrier counting waves -- outside the HIP programming model, so a compiler that
    tail-merges, sinks or duplicates a barrier call in one branch only would break the pairing (deadlock, or a hand-off
    through LDS that is no longer ordered).  What the pairing needs from the code object: every instantiation carries exactly
    the SIX static barrier sites of the source (one before the step loop, one inside it, one after it, per role), three on
    each side of the role branch."""
    import __graft_entry__ as g

    g.build()
    fat, co = os.path.join(str(tmp_path), "ppo.fatbin"), os.path.join(str(tmp_path), "ppo.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", OBJ], check=True, capture_output=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={fat}", f"--output={co}"], check=True, capture_output=True)
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--demangle", co], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif cur is not None:
            m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
            if m:
                kernels[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    split = {k: v for k, v in kernels.items() if "rollout_split_kernel" in k}
    assert len(split) >= 3 * 3 * 7
    bad = []
    for name, ins in split.items():
        bars = [i for i, (_, op, _) in enumerate(ins) if op == "s_barrier"]
        if len(bars) != 6:
            bad.append(f"{name[:100]}: {len(bars)} s_barrier sites (expected 6)")
            continue
        # backward branches = loops; a barrier `in a loop` lies between a backward branch and its target
        addr = {a: i for i, (a, _, _) in enumerate(ins)}
        loops = []
        for i, (a, op, args) in enumerate(ins):
            if op.startswith("s_cbranch") or op == "s_branch":
                try:
                    off = int(args.split()[0])
                except (ValueError, IndexError):
                    continue
                off = off - 65536 if off >= 32768 else off
                tgt = a + 4 + 4 * off
                if off < 0 and tgt in addr:
                    loops.append((addr[tgt], i))
        in_loop = [sum(1 for lo, hi in loops if lo <= b <= hi) > 0 for b in bars]
        # per role: one site before the step loop, one inside it, one after it.  Block placement may put a site that is outside
        # every loop in the source inside the ADDRESS range of an unrelated loop (the noise-fill loop), so only the robust part
        # is asserted: at least the two step-loop sites are in a loop, and at least two sites are in none
        if sum(in_loop) < 2 or sum(not x for x in in_loop) < 2:
            bad.append(f"{name[:100]}: barrier sites in loops = {in_loop}")
    assert not bad, "\n".join(bad[:10])
