"""Static check of the Julia glue (reinforcementlearning.jl_amd/julia/RLHip.jl) against include/rlhip.h.

There is no `julia` binary in the image, so the glue cannot be executed here.  What CAN be verified mechanically is
what breaks a `ccall` binding silently: the symbol name, the number of arguments, the C type of every argument and of
the return value, and the layout (field order and types) of every struct passed by reference.  This test parses both
files and compares them.  (The same call SEQUENCE is executed from a PyTorch-free C process on the GPU:
tests/abi_host/abi_host.c, tests/test_gpu_abi_host.py.)"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rlhip.h")
GLUE = os.path.join(ROOT, "reinforcementlearning.jl_amd", "julia", "RLHip.jl")

SCALARS = {"int32_t": "Int32", "int64_t": "Int64", "uint32_t": "UInt32", "uint64_t": "UInt64", "float": "Float32",
           "double": "Float64", "size_t": "Csize_t", "uint8_t": "UInt8", "uint16_t": "UInt16", "char": "UInt8"}
HANDLES = {"rlhip_stream_t", "rlhip_event_t", "rlhip_comm_t"}  # typedef void*


def _header_src():
    with open(HEADER) as f:
        src = f.read()
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def c_type_class(decl):
    """class of one C parameter / field declaration: 'ptr' or the Julia scalar name"""
    decl = decl.strip()
    if "*" in decl or "[" in decl:
        return "ptr"
    toks = [t for t in re.split(r"\s+", decl) if t not in ("const", "struct")]
    base = toks[0]
    if base in HANDLES:
        return "ptr"
    return SCALARS[base]


def header_prototypes():
    protos = {}
    for ret, name, params in re.findall(r"\b(int32_t|int64_t|double|float|const char\s*\*)\s+(rlhip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;",
                                        _header_src()):
        params = params.strip()
        plist = [] if params in ("", "void") else [c_type_class(p) for p in params.split(",")]
        rt = "Cstring" if "char" in ret else SCALARS[ret]
        protos[name] = (rt, plist)
    return protos


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "{(":
            depth += 1
        elif ch in "})":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out if x.strip()]


def jl_type_class(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring",):
        return "ptr"
    return t


def glue_ccalls():
    with open(GLUE) as f:
        src = f.read()
    src = re.sub(r"#[^\n]*", "", src)
    calls = []
    for m in re.finditer(r"ccall\(\(:(rlhip_[a-z0-9_]+),\s*LIB\),\s*([A-Za-z0-9_{}]+),\s*\(", src):
        i = m.end()
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        args = split_top(src[i:j - 1])
        calls.append((m.group(1), m.group(2), [jl_type_class(a) for a in args], src.count("\n", 0, m.start()) + 1))
    return calls


def test_every_ccall_matches_its_prototype():
    protos = header_prototypes()
    assert len(protos) >= 120
    calls = glue_ccalls()
    assert len(calls) >= 60
    for name, ret, args, line in calls:
        assert name in protos, f"RLHip.jl:{line}: {name} is not declared in include/rlhip.h"
        rt, plist = protos[name]
        assert (ret if ret != "Cstring" else "Cstring") == rt, f"RLHip.jl:{line}: {name} returns {rt}, glue says {ret}"
        assert len(args) == len(plist), f"RLHip.jl:{line}: {name} takes {len(plist)} arguments, glue passes {len(args)}"
        for k, (a, p) in enumerate(zip(args, plist)):
            assert a == p, f"RLHip.jl:{line}: {name} argument {k + 1}: header {p}, glue {a}"


def test_the_integration_table_entry_points_are_bound():
    """every ABI function INTEGRATION.md names as the replacement of a reference call site has a ccall in the glue"""
    bound = {c[0] for c in glue_ccalls()}
    needed = ["rlhip_env_reset", "rlhip_env_step", "rlhip_env_obs", "rlhip_cartpole_default", "rlhip_pendulum_default",
              "rlhip_mountaincar_default", "rlhip_acrobot_default", "rlhip_ring_init", "rlhip_ring_push_state",
              "rlhip_ring_push_transition", "rlhip_ring_length", "rlhip_ring_sample_indices", "rlhip_ring_gather",
              "rlhip_dqn_plan_f32", "rlhip_dqn3_plan_f32", "rlhip_dqn_update_f32", "rlhip_dqn3_update_f32",
              "rlhip_dqn_vec_step_f32", "rlhip_mlp2_init_f32", "rlhip_mlp3_init_f32", "rlhip_mlp3_pack_bf16",
              "rlhip_mlp2_forward_f32", "rlhip_clip_adam_f32", "rlhip_polyak_f32", "rlhip_clip_by_global_norm_f32",
              "rlhip_ppo_default", "rlhip_ppo_nparams", "rlhip_ppo_plan_f32", "rlhip_ppo_rollout_f32", "rlhip_ppo_gae_f32",
              "rlhip_ppo_update_f32", "rlhip_ppo_update_comm_f32", "rlhip_ppo_workspace_bytes", "rlhip_gae_f32",
              "rlhip_discount_rewards_f32", "rlhip_hook_episode_stats", "rlhip_comm_unique_id", "rlhip_comm_init",
              "rlhip_comm_export", "rlhip_p2p_setup", "rlhip_allreduce_grads", "rlhip_comm_check", "rlhip_comm_info",
              "rlhip_comm_destroy", "rlhip_malloc", "rlhip_free", "rlhip_memset", "rlhip_memcpy_h2d", "rlhip_memcpy_d2h",
              "rlhip_memcpy_d2d", "rlhip_stream_create", "rlhip_stream_sync", "rlhip_event_create", "rlhip_event_record",
              "rlhip_event_elapsed_ms", "rlhip_device_count", "rlhip_set_device", "rlhip_abi_version", "rlhip_last_error"]
    missing = [n for n in needed if n not in bound]
    assert not missing, missing
    # and the reference-side types of SURVEY 8b exist
    with open(GLUE) as f:
        src = f.read()
    for t in ("mutable struct HipVecEnv", "mutable struct HipTrajectory", "mutable struct HipApproximator",
              "mutable struct HipTargetNetwork", "mutable struct HipQBasedPolicy", "mutable struct HipPPOPolicy",
              "function HipCartPoleEnv", "function HipPendulumEnv", "function HipMountainCarEnv", "function HipAcrobotRK4Env",
              "state_space(env::HipVecEnv{:cartpole})", "Base.copy(env::HipVecEnv", "Random.seed!(env::HipVecEnv",
              "function Base.iterate(t::HipTrajectory", "function _run(policy::AbstractPolicy, env::HipVecEnv"):
        assert t in src, f"RLHip.jl lacks `{t}`"


def c_structs():
    out = {}
    src = _header_src()
    for body, name in re.findall(r"typedef struct(?:\s+\w+)?\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"((?:const\s+)?(?:struct\s+)?\w+)\s*(.*)", decl)
            base, rest = m.group(1), m.group(2)
            for item in rest.split(","):
                item = item.strip()
                arr = re.search(r"\[(\d+)\]", item)
                is_ptr = "*" in item or base.replace("const ", "") in HANDLES
                cls = "ptr" if is_ptr else SCALARS[base.replace("const ", "").replace("struct ", "")]
                if arr:
                    fields.extend([cls] * int(arr.group(1)))
                else:
                    fields.append(cls)
        out[name] = fields
    return out


def jl_structs():
    with open(GLUE) as f:
        src = re.sub(r"#[^\n]*", "", f.read())
    out = {}
    for name, body in re.findall(r"(?:mutable\s+)?struct\s+(\w+)(?:\s*<:\s*\w+)?\s*\n(.*?)\nend", src, flags=re.S):
        fields = []
        for part in re.split(r"[;\n]", body):
            part = part.strip()
            m = re.match(r"^(\w+)::(.+)$", part)
            if not m:
                continue
            t = m.group(2).strip()
            nt = re.match(r"NTuple\{(\d+),\s*(.+)\}$", t)
            if nt:
                fields.extend([jl_type_class(nt.group(2))] * int(nt.group(1)))
            else:
                fields.append(jl_type_class(t))
        out[name] = fields
    return out


@pytest.mark.parametrize("jl,c", [("CartPoleCfg", "rlhip_cartpole_cfg"), ("PendulumCfg", "rlhip_pendulum_cfg"),
                                  ("MountainCarCfg", "rlhip_mountaincar_cfg"), ("AcrobotCfg", "rlhip_acrobot_cfg"),
                                  ("EnvState", "rlhip_env_state"), ("Ring", "rlhip_ring"), ("PPOCfg", "rlhip_ppo_cfg"),
                                  ("PPOTraj", "rlhip_ppo_traj"), ("DqnStepArgs", "rlhip_dqn_step_args"),
                                  ("CommDesc", "rlhip_comm_desc")])
def test_struct_mirrors_have_the_c_layout(jl, c):
    cs, js = c_structs(), jl_structs()
    assert c in cs and jl in js
    assert js[jl] == cs[c], f"{jl} vs {c}:\n julia {js[jl]}\n c     {cs[c]}"


def test_julia_dqn_update_gates_consult_the_sample_ratio_controller():
    """both DQN loops of the Julia host (optimise!(::HipDQNLearner) and the fused _run) must gate an update on
    InsertSampleRatioController.on_sample! like rlhip/dqn.py should_update_ (and the reference's `for batch in trajectory`)"""
    src = open(GLUE).read()
    opt = src[src.index("function optimise!(L::HipDQNLearner"):]
    opt = opt[:opt.index("\nend\n")]
    assert "on_sample!(t.controller)" in opt
    fused = src[src.index("a.do_update = ("):]
    assert "on_sample!(t.controller)" in fused[:300]
    py = open(os.path.join(ROOT, "reinforcementlearning.jl_amd", "rlhip", "dqn.py")).read()
    assert "trajectory.controller.on_sample_()" in py


def test_f_rows_are_methods_on_the_reference_types_and_match_integration_md():
    """INTEGRATION.md promises methods on HipPrioritizedTraces / on the reference's explorer types / a
    StackFrames-shaped sampler.  The Julia module must define exactly those, each forwarding to a free function whose ccall
    the signature tests above already pinned against include/rlhip.h, and INTEGRATION.md must name them."""
    src = re.sub(r"#[^\n]*", "", open(GLUE).read())
    # prioritized traces: type, push! on both NamedTuple shapes, sample, setindex!(:priority)
    assert re.search(r"mutable struct HipPrioritizedTraces\b", src)
    assert re.search(r"Base\.push!\(p::HipPrioritizedTraces, x::NamedTuple\{\(:state,\)\}\)", src)
    body = src[src.index("function Base.push!(p::HipPrioritizedTraces, x::NamedTuple{(:state, :action, :reward, :terminal)})"):]
    body = body[:body.index("\nend")]
    assert "push!(p.traces, x)" in body and "push_priority!(p.traces, p.tree, p.default_priority)" in body
    smp = src[src.index("function sample(p::HipPrioritizedTraces, batchsize::Integer)"):]
    smp = smp[:smp.index("\nend")]
    assert "sample_prioritized!(idx, key, prio, p.traces, p.tree, batchsize, p.seed, p.draw_ctr)" in smp and "p.draw_ctr +=" in smp
    seti = src[src.index("function Base.setindex!(p::HipPrioritizedTraces, v::DevBuf{Float32}, name::Symbol, keys::DevBuf{Int64})"):]
    assert "set_priority!(p.tree, p.n_leaves, keys, v, keys.n)" in seti[:seti.index("\nend")]
    # StackFrames at sample time
    assert re.search(r"struct HipStackFrames\b", src)
    sf = src[src.index("function sample(sf::HipStackFrames, t::HipTrajectory, inds::DevBuf{Int64})"):]
    assert "gather_stacked!(t, inds, b, sf.n_stack, s, a, r, term, sn)" in sf[:sf.index("\nend")]
    # explorers: plan! methods on the reference's own types (kinds 0 / 1 / 2 of rlhip_explorer_select_f32)
    for typ, kind in (("WeightedExplorer{N}", 0), ("WeightedSoftmaxExplorer", 1), ("GumbelSoftmaxExplorer", 2)):
        m = re.search(r"plan!\(s::ReinforcementLearningCore\." + re.escape(typ) + r", v::DevValues, mask = nothing; kw\.\.\.\)[^=]*=\s*_plan_dev\((\d)", src)
        assert m and int(m.group(1)) == kind, typ
    assert "function plan!(s::EpsilonGreedyExplorer{<:Any,TIE}, v::DevValues, mask = nothing" in src
    # prob(explorer, values[, mask]): a method of RLBase.prob on the reference's explorer type, break-tie flag
    # from the type parameter, forwarding to rlhip_eps_greedy_prob_f32
    pr = src[src.index("function RLBase.prob(s::EpsilonGreedyExplorer{<:Any,TIE}, v::DevValues, mask = nothing) where {TIE}"):]
    pr = pr[:pr.index("\nend")]
    assert ":rlhip_eps_greedy_prob_f32" in pr and "get_ϵ(s), TIE" in pr
    assert "function plan!(s::ReinforcementLearningCore.UCBExplorer, v::DevValues, counts::DevBuf{Float64}" in src
    assert "plan!(x::ReinforcementLearningCore.BatchExplorer, v::DevValues, args...; kw...) = plan!(x.explorer, v, args...; kw...)" in src
    # the reference's explorer types really have the fields the methods read
    ref = "/root/reference/src/ReinforcementLearningCore/src/policies/explorers"
    if os.path.isdir(ref):  # (absent on the GPU box; this is a CPU test)
        ucb = open(os.path.join(ref, "UCB_explorer.jl")).read()
        assert re.search(r"\bc::Float64", ucb) and re.search(r"\bstep::Int", ucb)
        assert "struct WeightedExplorer{T,R<:AbstractRNG}" in open(os.path.join(ref, "weighted_explorer.jl")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for needle in ("HipPrioritizedTraces(traces; default_priority)", "HipStackFrames(n_stack)", "v::DevValues[, mask]"):
        assert needle in doc, needle
    for name in ("HipPrioritizedTraces", "HipStackFrames", "DevValues"):
        assert re.search(r"export[^#]*\b" + name + r"\b", open(GLUE).read(), flags=re.S), name
    # `sample` extends StatsBase.sample (the function RLCore's explorers and RLTrajectories' samplers share) and is NOT a new
    # exported generic: imported from RLCore's namespace, where `using StatsBase: sample` put it
    assert re.search(r"^import ReinforcementLearningCore: sample$", open(GLUE).read(), flags=re.M)
    export_stmt = re.search(r"^export .*?(?=\n\S)", src, flags=re.S | re.M).group(0)
    assert "HipVecEnv" in export_stmt and not re.search(r"\bsample\b", export_stmt)
    assert "using StatsBase: sample, Weights" in open("/root/reference/src/ReinforcementLearningCore/src/policies/explorers/weighted_explorer.jl").read() \
        if os.path.isdir(ref) else True


def _strip_julia_strings_and_comments(s):
    out, i, n = [], 0, len(s)
    while i < n:
        if s.startswith('"""', i):
            j = s.find('"""', i + 3)
            j = n if j < 0 else j + 3
            out.append('""' + "\n" * s[i:j].count("\n"))
            i = j
        elif s[i] == '"':
            j = i + 1
            while j < n and s[j] != '"':
                j += 2 if s[j] == "\\" else 1
            out.append('""')
            i = j + 1
        elif s.startswith("#=", i):
            j = s.find("=#", i)
            j = n if j < 0 else j + 2
            out.append("\n" * s[i:j].count("\n"))
            i = j
        elif s[i] == "#":
            j = s.find("\n", i)
            i = n if j < 0 else j
        elif s[i] == "'" and i + 2 < n and (s[i + 2] == "'" or (s[i + 1] == "\\" and s[i + 3:i + 4] == "'")):
            out.append("' '")
            i = i + 3 if s[i + 2] == "'" else i + 4
        else:
            out.append(s[i])
            i += 1
    return "".join(out)


def test_julia_module_blocks_and_brackets_balance():
    """No Julia parser exists in this image ("1128 lines that no Julia parser has ever seen").  The cheapest class of
    error an edit can introduce is structural: a missing `end`, an unclosed bracket.  This is a token-level check of exactly that --
    every block opener (function / struct / if / for / while / let / begin / module / try / macro / quote / do) is closed by an
    `end`, no `end` is left over, and (), [], {} nest properly -- with strings, comments and `a[end]` indexing stripped.  Not a
    parser: `for` / `if` count as openers only in statement position (generators and comprehensions have no `end`)."""
    t = _strip_julia_strings_and_comments(open(GLUE).read())
    stack, bad = [], []
    for ln, line in enumerate(t.split("\n"), 1):
        l = re.sub(r"\[[^\[\]]*\]", lambda m: m.group(0).replace("end", "END"), line)
        for tok in re.findall(r"\b(mutable struct|struct|function|if|for|while|let|begin|module|try|macro|quote|do|end)\b", l):
            if tok == "end":
                if not stack:
                    bad.append((ln, "`end` without an opener", line.strip()[:80]))
                else:
                    stack.pop()
            elif tok in ("for", "if"):
                if (re.match(r"\s*%s\b" % tok, l) or re.search(r";\s*%s\b" % tok, l) or re.search(r"=\s*%s\b" % tok, l)
                        or re.search(r"\belse\s+%s\b" % tok, l)):
                    stack.append((tok, ln))
            else:
                stack.append((tok, ln))
    assert not bad, bad[:5]
    assert not stack, f"unclosed blocks (opener, line): {stack[-5:]}"
    pairs, opens = {")": "(", "]": "[", "}": "{"}, []
    for ln, line in enumerate(t.split("\n"), 1):
        for ch in line:
            if ch in "([{":
                opens.append((ch, ln))
            elif ch in ")]}":
                assert opens and opens[-1][0] == pairs[ch], f"line {ln}: unexpected {ch!r} (open: {opens[-1:]})"
                opens.pop()
    assert not opens, f"unclosed brackets: {opens[-5:]}"
