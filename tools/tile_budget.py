"""Static instruction budget of a kernel's steady-state loop from the disassembly of its object (runs here, no GPU).

    python tools/tile_budget.py <object.o> <kernel name substring> [more substrings ...]

Finds the kernel, takes the LARGEST backward-branch loop body (the per-tile loop of the persistent learner kernels: its branch
target .. the s_cbranch that closes it), and counts instructions by class: MFMA, VALU by opcode family, LDS, global memory, SALU,
waits.  Cycle estimates use the issue rates measured on this chip (profiles/r04_overlap.md): one wave per SIMD issues a f32
VALU instruction every 5.8 cycles (2.9 with two or more waves), a 32x32x16 bf16 MFMA occupies the matrix pipe for 32 cycles (16
passes x 2; `v_mfma_f32_32x32x2_f32` 64)."""
import collections, os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    tmp = tempfile.mkdtemp()
    fat, co = os.path.join(tmp, "a.fatbin"), os.path.join(tmp, "a.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], check=True, capture_output=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={fat}", f"--output={co}"], check=True, capture_output=True)
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--demangle", co], check=True, capture_output=True, text=True).stdout


def kernels(dis):
    cur, out = None, collections.OrderedDict()
    for ln in dis.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:", ln)
        if m:
            cur = m.group(2)
            out[cur] = []
        elif cur is not None:
            m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
            if m:
                out[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma:" + op
    if op.startswith("v_cvt_pk_bf16") or op.startswith("v_cvt"):
        return "valu:cvt"
    if op.startswith(("v_fma", "v_fmac", "v_mac", "v_mad")):
        return "valu:fma"
    if op.startswith(("v_add_f", "v_sub_f", "v_mul_f", "v_pk_")):
        return "valu:add/mul"
    if op.startswith(("v_max", "v_min", "v_cmp", "v_cndmask", "v_med")):
        return "valu:max/cmp/select"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos", "v_ldexp", "v_frexp", "v_div")):
        return "valu:transcendental"
    if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_writelane", "v_readfirstlane", "v_perm", "v_swap", "v_bfe", "v_and",
                      "v_or", "v_xor", "v_lshl", "v_lshr", "v_ashr", "v_add_u", "v_add_co", "v_addc", "v_sub_u", "v_mul_lo",
                      "v_mul_hi", "v_mad_u", "v_mad_i", "v_add3", "v_lshl_add", "v_bfi", "v_alignbit", "v_not", "v_dot")):
        return "valu:move/int"
    if op.startswith("v_"):
        return "valu:other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier") or op.startswith("s_sleep"):
        return "wait:" + op
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    obj, pats = sys.argv[1], sys.argv[2:]
    ks = kernels(disassemble(obj))
    for name, ins in ks.items():
        if not all(p in name for p in pats) or not ins:
            continue
        addr = {a: i for i, (a, _, _) in enumerate(ins)}
        best = None
        for i, (a, op, args) in enumerate(ins):
            if op.startswith("s_cbranch") or op == "s_branch":
                try:
                    off = int(args.split()[0])
                except (ValueError, IndexError):
                    continue
                if off >= 32768:
                    off -= 65536  # simm16, in dwords, relative to the next instruction
                tgt = a + 4 + 4 * off
                if off < 0 and tgt in addr:
                    body = (addr[tgt], i)
                    if best is None or body[1] - body[0] > best[1] - best[0]:
                        best = body
        if best is None:
            print(f"{name[:110]}: no loop found ({len(ins)} instructions)")
            continue
        cnt = collections.Counter(classify(op) for _, op, _ in ins[best[0]:best[1] + 1])
        n_valu = sum(v for k, v in cnt.items() if k.startswith("valu"))
        n_mfma = sum(v for k, v in cnt.items() if k.startswith("mfma"))
        mfma_cycles = sum(v * (64 if "x2_f32" in k or "x2f32" in k else 32) for k, v in cnt.items() if k.startswith("mfma"))
        print(f"== {name[:150]}\n   loop body: {best[1] - best[0] + 1} of {len(ins)} instructions")
        for k in sorted(cnt):
            print(f"   {k:34s} {cnt[k]:6d}")
        print(f"   VALU total {n_valu}, MFMA {n_mfma} ({mfma_cycles} matrix-pipe cycles); VALU issue at one wave per SIMD: "
              f"{n_valu * 5.8:.0f} cycles, at two: {n_valu * 2.9:.0f}")


if __name__ == "__main__":
    main()
