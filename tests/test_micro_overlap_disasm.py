"""tools/micro/mfma_valu_overlap.hip is only a measurement of "v_fma_f32 beside MFMAs" if its timed loops
contain exactly that.  Round 3's copy (plain -O3) timed 16 v_pk_fma_f32 and an s_nop 11.  This test disassembles the binary
__graft_entry__.build() produces (CPU only) and checks every kernel's loop: scalar v_fma_f32 / v_mfma only, no packed f32 op,
no s_nop >= 4, and the instruction counts the file's header states."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
EXE = os.path.join(ROOT, "tools", "micro", "mfma_valu_overlap.bin")


def _loops(tmp):
    """{kernel: [loop bodies]}: a loop body = the instructions between a backward branch's target and the branch"""
    co = os.path.join(tmp, "ovl.co")
    fat = os.path.join(tmp, "ovl.fatbin")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", EXE], check=True, capture_output=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={fat}", f"--output={co}"], check=True, capture_output=True)
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-Fa-f]+):", ln)
        if cur is not None and m:
            kernels[cur].append((int(m.group(2), 16), m.group(1).strip()))
    out = {}
    for name, ins in kernels.items():
        bodies = []
        for k, (addr, text) in enumerate(ins):
            m = re.match(r"s_cbranch_\w+\s+(\d+)", text)
            if not m:
                continue
            # SOPP branch target = address of the next instruction + 4 * simm16 (two's complement)
            off = int(m.group(1))
            off = off - 65536 if off >= 32768 else off
            tgt = addr + 4 + 4 * off
            if tgt <= addr:
                bodies.append([t for a, t in ins if tgt <= a <= addr])
        out[name] = bodies
    return out


@pytest.mark.skipif(not os.path.exists(f"{LLVM}/llvm-objdump"), reason="no llvm-objdump")
def test_overlap_micro_times_scalar_fmas_and_hazard_free_mfmas(tmp_path):
    import __graft_entry__ as g

    g.build_micro_checks()
    loops = _loops(str(tmp_path))
    ovl = {k: v for k, v in loops.items() if "ovl" in k}
    assert len(ovl) >= 10, sorted(loops)
    n_checked = 0
    for name, bodies in ovl.items():
        assert bodies, f"{name}: no loop found"
        for body in bodies:
            txt = "\n".join(body)
            n_fma = len(re.findall(r"\bv_fma_f32\b", txt))
            n_mfma = len(re.findall(r"\bv_mfma_", txt))
            if n_fma + n_mfma == 0:
                continue  # the epilogue's reduction loops, if the compiler kept any
            n_checked += 1
            assert not re.search(r"\bv_pk_\w+_f32\b", txt), f"{name}: packed f32 op in a timed loop\n{txt}"
            for m in re.finditer(r"\bs_nop\s+(\d+)", txt):
                assert int(m.group(1)) < 4, f"{name}: s_nop {m.group(1)} in a timed loop\n{txt}"
            # every timed loop is one iteration of the header's streams: NV in {16, 32} FMAs and / or 4 MFMAs
            assert n_fma in (0, 16, 32) and n_mfma in (0, 4), (name, n_fma, n_mfma)
            other = [t for t in body if not re.match(r"(v_fma_f32|v_mfma_|s_add|s_sub|s_cmp|s_cbranch|s_nop)", t)]
            assert len(other) <= 2, f"{name}: unexpected instructions in a timed loop: {other}"
    assert n_checked >= 16, n_checked
