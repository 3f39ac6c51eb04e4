# RLHip.jl -- Julia glue for librlhip.so: the reference-side binding a maintainer adds (ONE file, no edits to
# ReinforcementLearning.jl).  New types subtype the reference's abstract types, so `AbstractEnv` / `AbstractPolicy` /
# `Trajectory` and `run(policy, env, stop_condition, hook)` are unchanged; every method below is a thin `ccall` into
# include/rlhip.h (0-based ABI; the glue shifts actions 1..na <-> 0..na-1).
#
# STATUS: the build image has no `julia` binary (SURVEY.md section 0), so this module cannot be executed here.  What IS
# checked: (1) tests/test_julia_glue_signatures.py parses every `ccall` below and compares its name, arity and
# argument types with the prototypes of include/rlhip.h, and every `struct` with the C struct it mirrors (field count,
# types); (2) tests/abi_host/abi_host.c makes the SAME call sequence from a process without PyTorch (rlhip_malloc,
# rlhip_stream_create, rlhip_memcpy_*, env -> ring push -> rlhip_dqn_vec_step_f32, rlhip_ppo_rollout/update,
# rlhip_comm_*), on the GPU, against the oracle -- tests/test_gpu_abi_host.py.  The tested host mirror with the same
# structure is reinforcementlearning.jl_amd/rlhip/ (Python over ctypes).
#
# Reference call sites each method replaces are cited as file:line under /root/reference/src/
#   RLBase = ReinforcementLearningBase/src, RLCore = ReinforcementLearningCore/src,
#   RLEnvs = ReinforcementLearningEnvironments/src/environments
module RLHip

using ReinforcementLearningBase, ReinforcementLearningCore
import ReinforcementLearningBase: state, reward, is_terminated, action_space, state_space, act!, reset!, plan!, optimise!
import ReinforcementLearningCore: _run, check!, forward, target, model, PreExperimentStage, PostExperimentStage,
    PreEpisodeStage, PostEpisodeStage, PreActStage, PostActStage, AbstractLearner, AbstractExplorer, Agent,
    EpsilonGreedyExplorer, get_ϵ
# `sample` is StatsBase.sample: RLCore's explorers bring it in (`using StatsBase: sample, Weights`, weighted_explorer.jl:4) and
# RLTrajectories extends the same function with `sample(sampler, traces)`.  The methods below are added to THAT function and the
# name is not exported from here, so `using RLHip` next to StatsBase / ReinforcementLearning cannot produce two `sample`s (ADVICE r4)
import ReinforcementLearningCore: sample
using Random, DomainSets

export HipVecEnv, HipCartPoleEnv, HipPendulumEnv, HipMountainCarEnv, HipAcrobotRK4Env, HipTrajectory, HipApproximator,
    HipTargetNetwork, HipDQNLearner, HipQBasedPolicy, HipPPOPolicy, HipComm, HipEpisodeStats, DevBuf, to_host, to_dev!,
    HipPrioritizedTraces, HipStackFrames, DevValues

const LIB = get(ENV, "RLHIP_LIB", "librlhip.so")

struct RLHipError <: Exception
    code::Int32
    msg::String
end
Base.showerror(io::IO, e::RLHipError) = print(io, "RLHipError($(e.code)): $(e.msg)")

"status -> exception: RLHIP_EINVAL is what the reference throws as ArgumentError / AssertionError / MethodError"
function chk(rc::Int32)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:rlhip_last_error, LIB), Cstring, ()))
    rc == -1 ? throw(ArgumentError(msg)) : throw(RLHipError(rc, msg))
end

# ------------------------------------------------------------------------------------------------------------------
# runtime: device, stream, buffers -- all through the ABI (no AMDGPU.jl / HIP.jl dependency)
# ------------------------------------------------------------------------------------------------------------------
const STREAM = Ref{Ptr{Cvoid}}(C_NULL)   # the compute stream every call is enqueued on

function __init__()
    n = Ref{Int32}(0)
    chk(ccall((:rlhip_device_count, LIB), Int32, (Ref{Int32},), n))
    n[] >= 1 || error("rlhip: no gfx950 device visible -- there is no CPU fallback for this path")
    ccall((:rlhip_abi_version, LIB), Int32, ()) == 2 || error("rlhip: ABI version mismatch")
    chk(ccall((:rlhip_set_device, LIB), Int32, (Int32,), parse(Int32, get(ENV, "RLHIP_DEVICE", "0"))))
    chk(ccall((:rlhip_stream_create, LIB), Int32, (Ref{Ptr{Cvoid}},), STREAM))
end
stream() = STREAM[]
synchronize() = chk(ccall((:rlhip_stream_sync, LIB), Int32, (Ptr{Cvoid},), stream()))

"a typed device buffer owned through rlhip_malloc / rlhip_free (zero-initialised: several ABI workspaces require it)"
mutable struct DevBuf{T}
    ptr::Ptr{Cvoid}
    n::Int
    function DevBuf{T}(n::Integer) where {T}
        p = Ref{Ptr{Cvoid}}(C_NULL)
        bytes = max(Int(n), 1) * sizeof(T)
        chk(ccall((:rlhip_malloc, LIB), Int32, (Ref{Ptr{Cvoid}}, Csize_t), p, bytes))
        chk(ccall((:rlhip_memset, LIB), Int32, (Ptr{Cvoid}, Int32, Csize_t, Ptr{Cvoid}), p[], 0, bytes, stream()))
        b = new{T}(p[], Int(n))
        finalizer(x -> ccall((:rlhip_free, LIB), Int32, (Ptr{Cvoid},), x.ptr), b)
        b
    end
end
Base.length(b::DevBuf) = b.n
Base.pointer(b::DevBuf) = b.ptr
offset(b::DevBuf{T}, elems::Integer) where {T} = b.ptr + elems * sizeof(T)

function to_host(b::DevBuf{T}) where {T}
    h = Vector{T}(undef, b.n)
    chk(ccall((:rlhip_memcpy_d2h, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), h, b.ptr, sizeof(h), stream()))
    h
end
function to_dev!(b::DevBuf{T}, h::AbstractArray{T}) where {T}
    @assert length(h) == b.n
    v = vec(collect(h))
    chk(ccall((:rlhip_memcpy_h2d, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), b.ptr, v, sizeof(v), stream()))
    b
end
function Base.copyto!(dst::DevBuf{T}, src::DevBuf{T}) where {T}
    chk(ccall((:rlhip_memcpy_d2d, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), dst.ptr, src.ptr,
              min(dst.n, src.n) * sizeof(T), stream()))
    dst
end
Base.copy(b::DevBuf{T}) where {T} = copyto!(DevBuf{T}(b.n), b)

"device timing of a section: the TimerOutputs labels of run.jl:46-72 can be fed from these (RLCore TimePerStep)"
function elapsed_ms(f)
    e0, e1 = Ref{Ptr{Cvoid}}(C_NULL), Ref{Ptr{Cvoid}}(C_NULL)
    chk(ccall((:rlhip_event_create, LIB), Int32, (Ref{Ptr{Cvoid}},), e0))
    chk(ccall((:rlhip_event_create, LIB), Int32, (Ref{Ptr{Cvoid}},), e1))
    chk(ccall((:rlhip_event_record, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), e0[], stream()))
    f()
    chk(ccall((:rlhip_event_record, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), e1[], stream()))
    ms = Ref{Float32}(0)
    chk(ccall((:rlhip_event_elapsed_ms, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float32}), e0[], e1[], ms))
    chk(ccall((:rlhip_event_destroy, LIB), Int32, (Ptr{Cvoid},), e0[]))
    chk(ccall((:rlhip_event_destroy, LIB), Int32, (Ptr{Cvoid},), e1[]))
    ms[]
end

# ------------------------------------------------------------------------------------------------------------------
# POD structs of include/rlhip.h (isbits, C layout)
# ------------------------------------------------------------------------------------------------------------------
struct CartPoleCfg      # rlhip_cartpole_cfg  <- CartPoleEnv(; kwargs...)  RLEnvs/examples/CartPoleEnv.jl:22-46
    gravity::Float64; masscart::Float64; masspole::Float64; halflength::Float64; forcemag::Float64
    dt::Float64; thetathreshold_deg::Float64; xthreshold::Float64; max_steps::Int64; continuous::Int32
end
struct PendulumCfg      # rlhip_pendulum_cfg  <- PendulumEnv(; kwargs...)  RLEnvs/examples/PendulumEnv.jl:41-66
    max_speed::Float64; max_torque::Float64; g::Float64; m::Float64; l::Float64; dt::Float64
    max_steps::Int64; continuous::Int32; n_actions::Int32
end
struct MountainCarCfg   # rlhip_mountaincar_cfg <- MountainCarEnv(; kwargs...)  RLEnvs/examples/MountainCarEnv.jl:19-40,67-81
    min_pos::Float64; max_pos::Float64; max_speed::Float64; goal_pos::Float64; goal_velocity::Float64
    power::Float64; gravity::Float64; max_steps::Int64; continuous::Int32
end
struct AcrobotCfg       # rlhip_acrobot_cfg   <- AcrobotEnv(; kwargs...)  RLEnvs/3rd_party/AcrobotEnv.jl:22-40 (parity unpinned)
    link_length_a::Float64; link_length_b::Float64; link_mass_a::Float64; link_mass_b::Float64
    link_com_pos_a::Float64; link_com_pos_b::Float64; link_moi::Float64; max_torque_noise::Float64
    max_vel_a::Float64; max_vel_b::Float64; g::Float64; dt::Float64; max_steps::Int64; nips::Int32
end
struct EnvState         # rlhip_env_state
    s::NTuple{4,Ptr{Cvoid}}; t::Ptr{Cvoid}; done::Ptr{Cvoid}; reward::Ptr{Cvoid}; episode::Ptr{Cvoid}
end
mutable struct Ring     # rlhip_ring (mutable: the push calls advance its host-side counters through the pointer)
    capacity::Int64; n_env::Int64; obs_dim::Int64
    head_sa::Int64; len_sa::Int64; head_rt::Int64; len_rt::Int64
    elem_bytes::Int32
    layout::Int32       # RING_FRAMES | RING_RECORDS (set by rlhip_ring_init)
    state::Ptr{Cvoid}; action::Ptr{Cvoid}; reward::Ptr{Cvoid}; terminal::Ptr{Cvoid}
    Ring() = new(0, 0, 0, 0, 0, 0, 0, 0, 0, C_NULL, C_NULL, C_NULL, C_NULL)
end
struct PPOCfg           # rlhip_ppo_cfg (blog a_practical_introduction_to_RL.jl/index.html:15257-15278)
    gamma::Float32; lambda::Float32; clip_range::Float32; max_grad_norm::Float32
    actor_loss_weight::Float32; critic_loss_weight::Float32; entropy_loss_weight::Float32
    lr::Float32; beta1::Float32; beta2::Float32; adam_eps::Float32
    n_epochs::Int32; n_microbatches::Int32; hidden::Int32; act::Int32; continuous::Int32
    normalize_advantage::Int32; layers::Int32
end
struct PPOTraj          # rlhip_ppo_traj: device pointers of the time-major traces
    obs::Ptr{Cvoid}; logp::Ptr{Cvoid}; value::Ptr{Cvoid}; reward::Ptr{Cvoid}; adv::Ptr{Cvoid}; ret::Ptr{Cvoid}
    action_f::Ptr{Cvoid}; action_i::Ptr{Cvoid}; terminal::Ptr{Cvoid}
end
mutable struct DqnStepArgs   # rlhip_dqn_step_args
    kind::Int32; env_cfg::Ptr{Cvoid}; st::Ptr{Cvoid}; n::Int64; env_seed::UInt64; env_id_base::UInt32
    obs::Ptr{Cvoid}; last_obs::Ptr{Cvoid}; ring::Ptr{Cvoid}; layers::Int32; h::Int64; na::Int64; act::Int32
    params::Ptr{Cvoid}; packed::Ptr{Cvoid}; target::Ptr{Cvoid}; target_packed::Ptr{Cvoid}
    m::Ptr{Cvoid}; v::Ptr{Cvoid}; beta_pow::Ptr{Cvoid}
    lr::Float32; beta1::Float32; beta2::Float32; adam_eps::Float32; max_grad_norm::Float32; grad_scale::Float32
    eps::Float64; explorer_seed::UInt64; explorer_step::UInt32; batch::Int64; gamma::Float32; huber_delta::Float32
    sampler_seed::UInt64; draw_ctr::UInt32; do_update::Int32; do_sync::Int32; rho::Float32
    workspace::Ptr{Cvoid}; grad::Ptr{Cvoid}; loss::Ptr{Cvoid}; gn::Ptr{Cvoid}; actions::Ptr{Cvoid}; q::Ptr{Cvoid}
    DqnStepArgs() = new()
end
struct CommDesc         # rlhip_comm_desc
    rank::Int32; world::Int32; device::Int32; p2p_active::Int32; rccl_active::Int32
    seq::UInt32; cap::Int64; timeout_polls::Int64; status::Ptr{Cvoid}
    bufs::NTuple{16,Ptr{Cvoid}}; why::NTuple{256,UInt8}; rccl_path::NTuple{256,UInt8}
end

"rebuild an immutable cfg struct with keyword overrides (thetathreshold is given in degrees like the reference)"
function with_kwargs(c::C; kwargs...) where {C}
    names = fieldnames(C)
    vals = Any[getfield(c, f) for f in names]
    for (k, v) in kwargs
        k = k === :thetathreshold ? :thetathreshold_deg : k
        i = findfirst(==(k), names)
        i === nothing && throw(MethodError(C, (k,)))      # unknown keyword, as the reference's constructor would
        vals[i] = convert(fieldtype(C, i), v)
    end
    C(vals...)
end

# ------------------------------------------------------------------------------------------------------------------
# HipVecEnv <: AbstractEnv: N independent instances of a classic-control env, SoA in HBM, one lane per instance
#   replaces N x {CartPoleEnv, PendulumEnv, MountainCarEnv, AcrobotEnv} behind the historical MultiThreadEnv protocol
#   (docs/homepage/blog/an_introduction_to_reinforcement_learning_jl_design_implementations_thoughts/index.md:347-376)
# ------------------------------------------------------------------------------------------------------------------
const KIND = (cartpole = Int32(0), pendulum = Int32(1), mountaincar = Int32(2), acrobot = Int32(3))

mutable struct HipVecEnv{K,T,C} <: AbstractEnv
    kind::Int32
    cfg::Base.RefValue{C}
    n::Int
    seed::UInt64
    env_id_base::UInt32
    s::Vector{DevBuf{T}}
    t::DevBuf{Int32}
    done::DevBuf{UInt8}
    rew::DevBuf{T}
    episode::DevBuf{UInt32}
    obs::DevBuf{T}          # (obs_dim, n) component-major: state(env) after the last act! / reset!
    last_obs::DevBuf{T}     # observation of the last act! BEFORE the auto-reset (the terminal observation)
    st::Base.RefValue{EnvState}
    obs_valid::Bool
end

obs_dim(kind) = Int(ccall((:rlhip_env_obs_dim, LIB), Int32, (Int32,), kind))
state_dim(kind) = Int(ccall((:rlhip_env_state_dim, LIB), Int32, (Int32,), kind))

function make_env(K::Symbol, cfg::C, n::Integer, ::Type{T}, seed, env_id_base) where {C,T}
    kind = KIND[K]
    sd, od = state_dim(kind), obs_dim(kind)
    s = [DevBuf{T}(n) for _ in 1:sd]
    sp = ntuple(k -> k <= sd ? s[k].ptr : C_NULL, 4)
    t, done, rew, ep = DevBuf{Int32}(n), DevBuf{UInt8}(n), DevBuf{T}(n), DevBuf{UInt32}(n)
    env = HipVecEnv{K,T,C}(kind, Ref(cfg), Int(n), UInt64(seed), UInt32(env_id_base), s, t, done, rew, ep,
                           DevBuf{T}(od * n), DevBuf{T}(od * n), Ref(EnvState(sp, t.ptr, done.ptr, rew.ptr, ep.ptr)), false)
    reset!(env)                                     # the constructors call reset! once (CartPoleEnv.jl:77)
    env
end

"CartPoleEnv(; T, continuous, gravity, ..., thetathreshold, xthreshold)  x n   RLEnvs/examples/CartPoleEnv.jl:57-79"
function HipCartPoleEnv(n::Integer; T = Float32, continuous = false, seed = 0, env_id_base = 0, kwargs...)
    c = Ref{CartPoleCfg}()
    chk(ccall((:rlhip_cartpole_default, LIB), Int32, (Ref{CartPoleCfg},), c))
    make_env(:cartpole, with_kwargs(c[]; continuous = continuous, kwargs...), n, T, seed, env_id_base)
end
"PendulumEnv(; T, max_speed, ..., continuous = true, n_actions = 3)  x n   RLEnvs/examples/PendulumEnv.jl:24-66"
function HipPendulumEnv(n::Integer; T = Float32, continuous = true, seed = 0, env_id_base = 0, kwargs...)
    c = Ref{PendulumCfg}()
    chk(ccall((:rlhip_pendulum_default, LIB), Int32, (Ref{PendulumCfg},), c))
    make_env(:pendulum, with_kwargs(c[]; continuous = continuous, kwargs...), n, T, seed, env_id_base)
end
"MountainCarEnv(; T, continuous, ...)  x n   RLEnvs/examples/MountainCarEnv.jl:51-81"
function HipMountainCarEnv(n::Integer; T = Float32, continuous = false, seed = 0, env_id_base = 0, kwargs...)
    c = Ref{MountainCarCfg}()
    chk(ccall((:rlhip_mountaincar_default, LIB), Int32, (Ref{MountainCarCfg}, Int32), c, continuous))
    make_env(:mountaincar, with_kwargs(c[]; kwargs...), n, T, seed, env_id_base)
end
"AcrobotEnv(; T, ...)  x n   RLEnvs/3rd_party/AcrobotEnv.jl:22-70 -- ONE classic RK4 step per act!: parity unpinned"
function HipAcrobotRK4Env(n::Integer; T = Float32, seed = 0, env_id_base = 0, kwargs...)
    c = Ref{AcrobotCfg}()
    chk(ccall((:rlhip_acrobot_default, LIB), Int32, (Ref{AcrobotCfg},), c))
    make_env(:acrobot, with_kwargs(c[]; kwargs...), n, T, seed, env_id_base)
end

Base.length(env::HipVecEnv) = env.n
is_continuous(env::HipVecEnv{:acrobot}) = false
is_continuous(env::HipVecEnv) = env.cfg[].continuous != 0

"reset!(env): all instances (is_force, run.jl:46) or only the terminated ones (MultiThreadEnv.reset!)"
function reset!(env::HipVecEnv{K,T}; is_force = true) where {K,T}
    chk(ccall((:rlhip_env_reset, LIB), Int32,
              (Int32, Int32, Ptr{Cvoid}, Ref{EnvState}, Int64, UInt64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}),
              env.kind, T === Float64, env.cfg, env.st, env.n, env.seed, env.env_id_base,
              is_force ? C_NULL : env.done.ptr, stream()))
    env.obs_valid = false
    nothing
end

"act!(env, actions) on a DEVICE vector of 0-based Int32 (discrete) or T (continuous) actions: what the policy kernels
produce -- replaces N x `act!` + `_step!` (CartPoleEnv.jl:106-140, PendulumEnv.jl:94-122, MountainCarEnv.jl:107-135)"
function act!(env::HipVecEnv{K,T}, actions::DevBuf) where {K,T}
    chk(ccall((:rlhip_env_step, LIB), Int32,
              (Int32, Int32, Ptr{Cvoid}, Ref{EnvState}, Int64, Ptr{Cvoid}, Int32, UInt64, UInt32, Ptr{Cvoid}, Ptr{Cvoid},
               Ptr{Cvoid}),
              env.kind, T === Float64, env.cfg, env.st, env.n, actions.ptr, 1, env.seed, env.env_id_base,
              env.last_obs.ptr, env.obs.ptr, stream()))
    env.obs_valid = true
    nothing
end
"host-side actions (1-based like the reference): `@assert a in action_space(env)` (CartPoleEnv.jl:113), then shifted"
function act!(env::HipVecEnv{K,T}, actions::AbstractVector) where {K,T}
    length(actions) == env.n || throw(ArgumentError("expected $(env.n) actions"))
    if is_continuous(env)
        act!(env, to_dev!(DevBuf{T}(env.n), T.(actions)))
    else
        @assert all(a -> a in action_space(env), actions)
        act!(env, to_dev!(DevBuf{Int32}(env.n), Int32.(actions .- 1)))
    end
end

"device observation buffer (obs_dim, n), refreshed lazily after reset! (ADVICE r1: it was stale before the first act!)"
function device_state(env::HipVecEnv{K,T}) where {K,T}
    if !env.obs_valid
        chk(ccall((:rlhip_env_obs, LIB), Int32, (Int32, Int32, Ref{EnvState}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                  env.kind, T === Float64, env.st, env.n, env.obs.ptr, stream()))
        env.obs_valid = true
    end
    env.obs
end
"state(env): the reference's batched (ns, N) matrix -- may be reused and mutated at each step (RLBase/interface.jl:515-517)"
state(env::HipVecEnv, ::Observation{Any}, ::DefaultPlayer) = permutedims(reshape(to_host(device_state(env)), env.n, :))
state(env::HipVecEnv) = state(env, Observation{Any}(), DefaultPlayer())
reward(env::HipVecEnv) = to_host(env.rew)                        # CartPoleEnv.jl:84
is_terminated(env::HipVecEnv) = Bool.(to_host(env.done))          # a vector, as BatchStepsPerEpisode expects (hooks.jl:219-231)
action_space(env::HipVecEnv{:cartpole}) = is_continuous(env) ? (-1.0 .. 1.0) : Base.OneTo(2)
action_space(env::HipVecEnv{:pendulum}) = is_continuous(env) ? (-env.cfg[].max_torque .. env.cfg[].max_torque) : Base.OneTo(Int(env.cfg[].n_actions))
action_space(env::HipVecEnv{:mountaincar}) = is_continuous(env) ? (-1.0 .. 1.0) : Base.OneTo(3)
action_space(env::HipVecEnv{:acrobot}) = Base.OneTo(3)
function state_space(env::HipVecEnv{:cartpole})                   # CartPoleEnv.jl:88-93
    c = env.cfg[]; th = c.thetathreshold_deg * π / 180
    ArrayProductDomain([-2c.xthreshold .. 2c.xthreshold, -Inf .. Inf, -2th .. 2th, -Inf .. Inf])
end
state_space(env::HipVecEnv{:pendulum}) =                          # PendulumEnv.jl:75-79
    ArrayProductDomain([-1.0 .. 1.0, -1.0 .. 1.0, -env.cfg[].max_speed .. env.cfg[].max_speed])
state_space(env::HipVecEnv{:mountaincar}) =                       # MountainCarEnv.jl:83-86
    ArrayProductDomain([env.cfg[].min_pos .. env.cfg[].max_pos, -env.cfg[].max_speed .. env.cfg[].max_speed])
state_space(env::HipVecEnv{:acrobot}) =                           # AcrobotEnv.jl:77-86
    ArrayProductDomain([-1.0 .. 1.0, -1.0 .. 1.0, -1.0 .. 1.0, -1.0 .. 1.0, -env.cfg[].max_vel_a .. env.cfg[].max_vel_a,
                        -env.cfg[].max_vel_b .. env.cfg[].max_vel_b])
"Random.seed!(env, seed): re-keys the Philox streams and restarts the episode counters"
function Random.seed!(env::HipVecEnv, seed)
    env.seed = UInt64(seed)
    to_dev!(env.episode, zeros(UInt32, env.n))
    env
end
"copy(env): deep copy, same seed and counters -> identical future under identical actions (RLBase/base.jl:77-130)"
function Base.copy(env::HipVecEnv{K,T,C}) where {K,T,C}
    s = [copy(b) for b in env.s]
    sp = ntuple(k -> k <= length(s) ? s[k].ptr : C_NULL, 4)
    t, done, rew, ep = copy(env.t), copy(env.done), copy(env.rew), copy(env.episode)
    HipVecEnv{K,T,C}(env.kind, Ref(env.cfg[]), env.n, env.seed, env.env_id_base, s, t, done, rew, ep, copy(env.obs),
                     copy(env.last_obs), Ref(EnvState(sp, t.ptr, done.ptr, rew.ptr, ep.ptr)), env.obs_valid)
end

# ------------------------------------------------------------------------------------------------------------------
# HipTrajectory: Trajectory(CircularArraySARTSTraces(; capacity, state = Float32 => (ns, n_env), action = Int32 => (n_env,),
#   reward, terminal), BatchSampler(batchsize), InsertSampleRatioController(...)) resident in HBM
#   (un-vendored ReinforcementLearningTrajectories 0.4; call sites RLCore/policies/agent/agent_base.jl:25,45-59,
#    RLCore/test/policies/q_based_policy.jl:41-47).  One frame = one vec-step.
# ------------------------------------------------------------------------------------------------------------------
mutable struct InsertSampleRatioController     # docs/src/How_to_implement_a_new_algorithm.md:108
    ratio::Float64
    threshold::Int
    n_inserted::Int
    n_sampled::Int
end
InsertSampleRatioController(; ratio = 1.0, threshold = 1, n_inserted = 0, n_sampled = 0) =
    InsertSampleRatioController(ratio, threshold, n_inserted, n_sampled)
on_insert!(c::InsertSampleRatioController, n = 1) = (c.n_inserted += n)
function on_sample!(c::InsertSampleRatioController)
    if c.n_inserted >= c.threshold && c.n_sampled <= (c.n_inserted - c.threshold) * c.ratio
        c.n_sampled += 1
        return true
    end
    false
end

const RING_FRAMES = Int32(0)     # include/rlhip.h RLHIP_RING_FRAMES: every trace as pushed
const RING_RECORDS = Int32(2)    # RLHIP_RING_RECORDS: Float32 observations with <= 4 components, one 64-byte record {s[4],
                                 # action::Int32, reward::Float32, terminal::UInt32, spare, s_next[4], pad[4]} per (state slot, env)
mutable struct HipTrajectory{E}
    rb::Ring
    # RING_FRAMES: the four traces; RING_RECORDS: `state` is the record buffer (16 Float32 words per record, (capacity + 1) *
    # n_env records: unsafe_wrap it as a (16, n_env, capacity + 1) array -- rows 1:obs_dim are the reference's state trace,
    # rows 5 / 6 / 7 reinterpret as the action / reward / terminal of the transition that LEAVES that state, rows
    # 9:8+obs_dim its next state) and the other three are empty
    state::DevBuf{E}; action::DevBuf{Int32}; reward::DevBuf{Float32}; terminal::DevBuf{UInt8}
    batchsize::Int
    sampler_seed::UInt64
    draw_ctr::UInt32
    controller::InsertSampleRatioController
    # the sampled batch (reused): s, s' (obs_dim, batch) SoA; a 0-based
    idx::DevBuf{Int64}; bs::DevBuf{E}; bs_next::DevBuf{E}; ba::DevBuf{Int32}; br::DevBuf{Float32}; bt::DevBuf{UInt8}
end
function HipTrajectory(; capacity, n_env, obs_dim, batchsize = 32, E = Float32, seed = 0,
                       controller = InsertSampleRatioController())
    rb = Ring()
    bytes = ccall((:rlhip_ring_state_bytes, LIB), Int64, (Int64, Int64, Int64, Int32), capacity, n_env, obs_dim, sizeof(E))
    records = sizeof(E) == 4 && obs_dim <= 4   # the C side's rule (csrc/ring_device.h ring_records: elem_bytes == 4 && obs_dim <= 4)
    st = DevBuf{E}(bytes ÷ sizeof(E))
    a, r, t = records ? (DevBuf{Int32}(0), DevBuf{Float32}(0), DevBuf{UInt8}(0)) :
              (DevBuf{Int32}(capacity * n_env), DevBuf{Float32}(capacity * n_env), DevBuf{UInt8}(capacity * n_env))
    chk(ccall((:rlhip_ring_init, LIB), Int32,
              (Ref{Ring}, Int64, Int64, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
              rb, capacity, n_env, obs_dim, sizeof(E), st.ptr, records ? C_NULL : a.ptr, records ? C_NULL : r.ptr,
              records ? C_NULL : t.ptr))
    ccall((:rlhip_ring_layout, LIB), Int32, (Ref{Ring},), rb) == (records ? RING_RECORDS : RING_FRAMES) || error("rlhip: unexpected ring layout")
    HipTrajectory{E}(rb, st, a, r, t, batchsize, UInt64(seed), UInt32(0), controller, DevBuf{Int64}(batchsize),
                     DevBuf{E}(obs_dim * batchsize), DevBuf{E}(obs_dim * batchsize), DevBuf{Int32}(batchsize),
                     DevBuf{Float32}(batchsize), DevBuf{UInt8}(batchsize))
end
"length(trajectory.container): stored frames (agent_base.jl:58; test RLCore/test/policies/agent.jl:27-34)"
Base.length(t::HipTrajectory) = Int(ccall((:rlhip_ring_length, LIB), Int64, (Ref{Ring},), t.rb))
capacity(t::HipTrajectory) = Int(t.rb.capacity)
Base.haskey(t::HipTrajectory, k::Symbol) = k in (:state, :next_state, :action, :reward, :terminal)   # no :next_action
"push!(trajectory, (state = s,))  -- Agent PreEpisodeStage, agent_base.jl:45-47.  `s`: device (obs_dim, n_env) buffer"
function Base.push!(t::HipTrajectory, x::NamedTuple{(:state,)})
    chk(ccall((:rlhip_ring_push_state, LIB), Int32, (Ref{Ring}, Ptr{Cvoid}, Ptr{Cvoid}), t.rb, x.state.ptr, stream()))
    t
end
"push!(trajectory, (state = s', action = a, reward = r, terminal = t))  -- PostActStage, agent_base.jl:56-59"
function Base.push!(t::HipTrajectory, x::NamedTuple{(:state, :action, :reward, :terminal)})
    chk(ccall((:rlhip_ring_push_transition, LIB), Int32,
              (Ref{Ring}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
              t.rb, x.state.ptr, x.action.ptr, x.reward.ptr, x.terminal.ptr, stream()))
    on_insert!(t.controller, 1)
    t
end
"`for batch in trajectory` (td_learner.jl:85-92): BatchSampler draw + LDS-staged gather; ends when the controller says so"
function Base.iterate(t::HipTrajectory, _ = nothing)
    (length(t) > 0 && on_sample!(t.controller)) || return nothing
    chk(ccall((:rlhip_ring_sample_indices, LIB), Int32, (Ref{Ring}, Int64, UInt64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}),
              t.rb, t.batchsize, t.sampler_seed, t.draw_ctr, t.idx.ptr, stream()))
    t.draw_ctr += 1
    chk(ccall((:rlhip_ring_gather, LIB), Int32,
              (Ref{Ring}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
              t.rb, t.idx.ptr, t.batchsize, t.bs.ptr, t.ba.ptr, t.br.ptr, t.bt.ptr, t.bs_next.ptr, stream()))
    ((state = t.bs, action = t.ba, reward = t.br, terminal = t.bt, next_state = t.bs_next), nothing)
end

"""
    sample_nstep!(t::HipTrajectory{Float32}, folded::HipTrajectory{Float32}, iota::DevBuf{Int64}, n, γ) -> γⁿ

`NStepBatchSampler(n, γ, batchsize)` of RLTrajectories 0.4 on the device: draws `t.batchsize` start indices that have `n` transitions
ahead of them and folds each window (cut at its first terminal step) into ONE transition `(s_i, a_i, R, any(terminal), s_{i+ns})`
with `R = discount_rewards_reduced(r_i … r_{i+ns-1}, γ)` (`RLCore/src/utils/basic.jl:237-319`), written as the records of `folded` --
a `HipTrajectory(capacity = 1, n_env = t.batchsize, obs_dim = …)`.  The DQN gradient `ccall`s then take `folded.rb`, `iota` (0-based
`0:batchsize-1`) and the returned `γⁿ` instead of `t.rb`, the sampled indices and `γ`: `R + γⁿ·(1−t)·max Qₜ(s_{i+n})` (SURVEY row L2).
"""
function sample_nstep!(t::HipTrajectory{Float32}, folded::HipTrajectory{Float32}, iota::DevBuf{Int64}, n::Integer, γ::Float32)
    chk(ccall((:rlhip_ring_sample_indices_nstep, LIB), Int32, (Ref{Ring}, Int64, Int32, UInt64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}),
              t.rb, t.batchsize, n, t.sampler_seed, t.draw_ctr, t.idx.ptr, stream()))
    t.draw_ctr += 1
    chk(ccall((:rlhip_ring_fold_nstep, LIB), Int32,
              (Ref{Ring}, Ptr{Cvoid}, Int64, Int32, Float32, Ref{Ring}, Ptr{Cvoid}, Ptr{Cvoid}),
              t.rb, t.idx.ptr, t.batchsize, n, γ, folded.rb, iota.ptr, stream()))
    ccall((:rlhip_gamma_pow, LIB), Float32, (Float32, Int32), γ, n)
end

"""
    check_indices(t::HipTrajectory, idx::DevBuf{Int64}, n = length(idx)) -> (n_bad, first_bad)

How many of the flat logical indices lie outside `1:length(t) * n_env` (0-based on the device), and the position of the first one
(-1 if none): the debugging aid behind the bounds-checked build (`lib/librlhip_bounds.so`, `bounds_checked_build()`), where the
reference's `traces[inds]` would throw a `BoundsError`.  One launch and a stream synchronisation.
"""
function check_indices(t::HipTrajectory, idx::DevBuf{Int64}, n::Integer = idx.n)
    n_bad = Ref{Int64}(0)
    first_bad = Ref{Int64}(-1)
    chk(ccall((:rlhip_ring_check_indices, LIB), Int32, (Ref{Ring}, Ptr{Cvoid}, Int64, Ref{Int64}, Ref{Int64}, Ptr{Cvoid}),
              t.rb, idx.ptr, n, n_bad, first_bad, stream()))
    (n_bad[], first_bad[])
end
"true when `LIB` is the build that validates caller-supplied gather indices inside every call (-DRLHIP_BOUNDS_CHECK)"
bounds_checked_build() = ccall((:rlhip_ring_bounds_checked_build, LIB), Int32, ()) != 0

# the Agent push protocol on device buffers (no host round trip): agent_base.jl:45-59
Base.push!(agent::Agent{P,<:HipTrajectory}, ::PreExperimentStage, env::HipVecEnv) where {P} =
    push!(agent.trajectory, (state = device_state(env),))          # the vector env has no episode stages: pushed once
Base.push!(agent::Agent{P,<:HipTrajectory}, ::PreEpisodeStage, env::HipVecEnv) where {P} = nothing
Base.push!(agent::Agent{P,<:HipTrajectory}, ::PostEpisodeStage, env::HipVecEnv) where {P} = nothing
Base.push!(agent::Agent{P,<:HipTrajectory}, ::PreActStage, env::HipVecEnv) where {P} = nothing
Base.push!(agent::Agent{P,<:HipTrajectory}, ::PostActStage, env::HipVecEnv, action::DevBuf{Int32}) where {P} =
    push!(agent.trajectory, (state = device_state(env), action = action, reward = env.rew, terminal = env.done))

# ------------------------------------------------------------------------------------------------------------------
# HipApproximator / HipTargetNetwork <: AbstractLearner
#   FluxApproximator(model = Chain(Dense(ns, h, act), [Dense(h, h, act),] Dense(h, nout)), optimiser = Adam(lr))
#   RLCore/policies/learners/flux_approximator.jl:11-46; TargetNetwork target_network.jl:27-88
# ------------------------------------------------------------------------------------------------------------------
mutable struct HipApproximator <: AbstractLearner
    n_in::Int; hidden::Int; n_out::Int; layers::Int; act::Int32
    params::DevBuf{Float32}; m::DevBuf{Float32}; v::DevBuf{Float32}; beta_pow::DevBuf{Float32}; gn::DevBuf{Float32}
    packed::Union{Nothing,DevBuf{UInt16}}     # layers == 3: bf16 MFMA fragments of the hidden x hidden layer
    lr::Float32; beta1::Float32; beta2::Float32; eps::Float32
end
function HipApproximator(n_in, hidden, n_out; layers = 2, act = 0, lr = 1f-3, beta1 = 0.9f0, beta2 = 0.999f0, eps = 1f-8,
                         seed = 0, net_id = 0)
    np = layers == 2 ? ccall((:rlhip_mlp2_nparams, LIB), Int64, (Int64, Int64, Int64), n_in, hidden, n_out) :
                       ccall((:rlhip_mlp3_nparams, LIB), Int64, (Int64, Int64, Int64), n_in, hidden, n_out)
    p = DevBuf{Float32}(np)
    if layers == 2     # glorot_uniform stand-in on the Philox INIT stream
        chk(ccall((:rlhip_mlp2_init_f32, LIB), Int32, (Ptr{Cvoid}, Int64, Int64, Int64, UInt64, UInt32, Ptr{Cvoid}),
                  p.ptr, n_in, hidden, n_out, seed, net_id, stream()))
    else
        chk(ccall((:rlhip_mlp3_init_f32, LIB), Int32, (Ptr{Cvoid}, Int64, Int64, Int64, UInt64, UInt32, Ptr{Cvoid}),
                  p.ptr, n_in, hidden, n_out, seed, net_id, stream()))
    end
    A = HipApproximator(n_in, hidden, n_out, layers, act, p, DevBuf{Float32}(np), DevBuf{Float32}(np),
                        to_dev!(DevBuf{Float32}(2), Float32[beta1, beta2]), DevBuf{Float32}(1),
                        layers == 3 ? DevBuf{UInt16}(ccall((:rlhip_mlp3_packed_elems, LIB), Int64, (Int64,), hidden)) : nothing,
                        lr, beta1, beta2, eps)
    repack!(A)
    A
end
"bf16 copies of W2 in both MFMA operand orders: refresh after every parameter update (layers == 3)"
function repack!(A::HipApproximator, params = A.params, packed = A.packed)
    A.layers == 3 || return nothing
    chk(ccall((:rlhip_mlp3_pack_bf16, LIB), Int32, (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
              params.ptr, A.n_in, A.hidden, A.n_out, packed.ptr, stream()))
end
"forward(A, x) = A.model(x)  flux_approximator.jl:43: x (n_in, batch) SoA device buffer -> (n_out, batch)"
function forward(A::HipApproximator, x::DevBuf{Float32}, batch::Integer)
    out = DevBuf{Float32}(A.n_out * batch)
    if A.layers == 2
        chk(ccall((:rlhip_mlp2_forward_f32, LIB), Int32,
                  (Ptr{Cvoid}, Int64, Int64, Int64, Int32, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                  A.params.ptr, A.n_in, A.hidden, A.n_out, A.act, x.ptr, batch, out.ptr, stream()))
    else     # pure forward of the MFMA network: actions = NULL
        chk(ccall((:rlhip_dqn3_plan_f32, LIB), Int32,
                  (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Int32, Ptr{Cvoid}, Int64, Float64, UInt64, UInt32, UInt32,
                   Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                  A.params.ptr, A.packed.ptr, A.n_in, A.hidden, A.n_out, A.act, x.ptr, batch, 0.0, 0, 0, 0, C_NULL,
                  out.ptr, stream()))
    end
    out
end
"optimise!(A, grad) = Flux.Optimise.update!(A.optimiser_state, A.model, grad)  flux_approximator.jl:46 (+ optional
clip_by_global_norm!, RLCore/utils/basic.jl:19-29, and the 1 / world scale after a gradient all-reduce)"
function optimise!(A::HipApproximator, grad::DevBuf{Float32}; clip_norm = 0f0, grad_scale = 1f0)
    chk(ccall((:rlhip_clip_adam_f32, LIB), Int32,
              (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float32, Float32, Float32, Float32, Float32,
               Float32, Ptr{Cvoid}, Ptr{Cvoid}),
              A.params.ptr, grad.ptr, A.m.ptr, A.v.ptr, A.beta_pow.ptr, A.params.n, grad_scale, clip_norm, A.lr, A.beta1,
              A.beta2, A.eps, A.gn.ptr, stream()))
    repack!(A)
end

mutable struct HipTargetNetwork <: AbstractLearner      # TargetNetwork(network; sync_freq = 1, ρ = 0f0)  target_network.jl:27-60
    network::HipApproximator
    target::DevBuf{Float32}
    target_packed::Union{Nothing,DevBuf{UInt16}}
    sync_freq::Int
    ρ::Float32
    n_optimise::Int
end
function HipTargetNetwork(network::HipApproximator; sync_freq = 1, ρ = 0f0)
    @assert 0 <= ρ <= 1 "ρ must in [0,1]"                # target_network.jl:50
    HipTargetNetwork(network, copy(network.params), network.packed === nothing ? nothing : copy(network.packed),
                     sync_freq, ρ, 0)
end
model(tn::HipTargetNetwork) = tn.network
target(tn::HipTargetNetwork) = tn.target
forward(tn::HipTargetNetwork, x, batch) = forward(tn.network, x, batch)
"optimise!(tn, grad)  target_network.jl:70-88: update the network, every sync_freq calls dest = ρ dest + (1 - ρ) src"
function optimise!(tn::HipTargetNetwork, grad::DevBuf{Float32}; kw...)
    optimise!(tn.network, grad; kw...)
    tn.n_optimise += 1
    if tn.n_optimise % tn.sync_freq == 0
        chk(ccall((:rlhip_polyak_f32, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float32, Ptr{Cvoid}),
                  tn.target.ptr, tn.network.params.ptr, tn.target.n, tn.ρ, stream()))
        tn.n_optimise = 0
        repack!(tn.network, tn.target, tn.target_packed)
    end
end

# ------------------------------------------------------------------------------------------------------------------
# HipDQNLearner + HipQBasedPolicy  (QBasedPolicy q_based_policy.jl:13-49; BasicDQN/DQN learner: removed Zoo, spec
#   docs/src/rlcore.md:28 and blog a_practical_introduction_to_RL.jl/index.html:15121-15147)
# ------------------------------------------------------------------------------------------------------------------
mutable struct HipDQNLearner <: AbstractLearner
    approximator::HipTargetNetwork
    batchsize::Int; γ::Float32; δ::Float32; min_replay_history::Int; update_freq::Int; max_grad_norm::Float32
    seed::UInt64; draw_ctr::UInt32; n_updates::Int; vec_steps::Int
    grad::DevBuf{Float32}; loss::DevBuf{Float32}; workspace::DevBuf{UInt8}
end
function HipDQNLearner(tn::HipTargetNetwork; batchsize = 32, γ = 0.99f0, huber_delta = 1f0, min_replay_history = 100,
                       update_freq = 1, max_grad_norm = 0f0, seed = 0)
    net = tn.network
    ws = net.layers == 2 ?
        ccall((:rlhip_dqn_workspace_bytes, LIB), Int64, (Int64, Int64, Int64, Int64), net.n_in, net.hidden, net.n_out, batchsize) :
        ccall((:rlhip_dqn3_workspace_bytes, LIB), Int64, (Int64, Int64, Int64, Int64), net.n_in, net.hidden, net.n_out, batchsize)
    HipDQNLearner(tn, batchsize, γ, huber_delta, min_replay_history, update_freq, max_grad_norm, UInt64(seed), UInt32(0), 0, 0,
                  DevBuf{Float32}(net.params.n), DevBuf{Float32}(1), DevBuf{UInt8}(ws))     # zeroed workspace: ABI contract
end
forward(L::HipDQNLearner, x, batch) = forward(L.approximator, x, batch)

"optimise!(learner, stage, trajectory)  (abstract_learner.jl; q_based_policy.jl:49): every `update_freq` vec-steps once
`min_replay_history` transitions are stored -- sample, TD target with the target network, Huber, gradient, clip, Adam,
target sync.  Sampling + gather are fused into the gradient launch (same draws as `for batch in trajectory`)."
function optimise!(L::HipDQNLearner, ::PostActStage, t::HipTrajectory)
    L.vec_steps += 1
    # the gate of rlhip/dqn.py should_update_: warm-up, every update_freq-th vec-step, then the trajectory's
    # InsertSampleRatioController (the reference's `for batch in trajectory` draws a batch only when it allows one)
    (length(t) * t.rb.n_env >= L.min_replay_history && L.vec_steps % L.update_freq == 0 && on_sample!(t.controller)) || return false
    tn, net = L.approximator, L.approximator.network
    if net.layers == 2
        chk(ccall((:rlhip_dqn_update_f32, LIB), Int32,
                  (Ref{Ring}, Int64, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float32, Float32, UInt64, UInt32, Ptr{Cvoid},
                   Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Float32, Float32, Float32,
                   Float32, Ptr{Cvoid}, Ptr{Cvoid}),
                  t.rb, net.hidden, net.n_out, net.act, net.params.ptr, tn.target.ptr, L.batchsize, L.γ, L.δ, L.seed,
                  L.draw_ctr, L.workspace.ptr, L.grad.ptr, L.loss.ptr, net.m.ptr, net.v.ptr, net.beta_pow.ptr, 1f0,
                  L.max_grad_norm, net.lr, net.beta1, net.beta2, net.eps, net.gn.ptr, stream()))
    else
        chk(ccall((:rlhip_dqn3_update_f32, LIB), Int32,
                  (Ref{Ring}, Int64, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float32, Float32,
                   UInt64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32,
                   Float32, Float32, Float32, Float32, Ptr{Cvoid}, Ptr{Cvoid}),
                  t.rb, net.hidden, net.n_out, net.act, net.params.ptr, net.packed.ptr, tn.target.ptr, tn.target_packed.ptr,
                  L.batchsize, L.γ, L.δ, L.seed, L.draw_ctr, L.workspace.ptr, L.grad.ptr, L.loss.ptr, net.m.ptr, net.v.ptr,
                  net.beta_pow.ptr, 1f0, L.max_grad_norm, net.lr, net.beta1, net.beta2, net.eps, net.gn.ptr, stream()))
    end
    L.draw_ctr += 1
    L.n_updates += 1
    tn.n_optimise += 1                       # the network update happened inside the fused call; the sync stays here
    if tn.n_optimise % tn.sync_freq == 0
        chk(ccall((:rlhip_polyak_f32, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float32, Ptr{Cvoid}),
                  tn.target.ptr, net.params.ptr, tn.target.n, tn.ρ, stream()))
        tn.n_optimise = 0
        repack!(net, tn.target, tn.target_packed)
    end
    true
end

mutable struct HipQBasedPolicy{E<:EpsilonGreedyExplorer} <: AbstractPolicy
    learner::HipDQNLearner
    explorer::E                     # the reference's own explorer struct: its schedule and `step` counter are used as is
    explorer_seed::UInt64
    actions::Union{Nothing,DevBuf{Int32}}
    q::Union{Nothing,DevBuf{Float32}}
end
HipQBasedPolicy(; learner, explorer, seed = 0) = HipQBasedPolicy(learner, explorer, UInt64(seed), nothing, nothing)

"plan!(p::QBasedPolicy, env) = plan!(explorer, forward(learner, env))  q_based_policy.jl:30-32 -- one launch: Q forward +
EpsilonGreedyExplorer (no tie-break; epsilon_greedy_explorer.jl:108-112).  Returns the DEVICE vector of 0-based actions"
function plan!(p::HipQBasedPolicy, env::HipVecEnv)
    net = p.learner.approximator.network
    p.actions === nothing && (p.actions = DevBuf{Int32}(env.n); p.q = DevBuf{Float32}(net.n_out * env.n))
    ϵ = get_ϵ(p.explorer)                     # the reference's schedule (epsilon_greedy_explorer.jl:69-90)
    step = UInt32(p.explorer.step)
    p.explorer.step += 1                      # :104,:110 -- incremented on every call, before the draw
    obs = device_state(env)
    if net.layers == 2
        chk(ccall((:rlhip_dqn_plan_f32, LIB), Int32,
                  (Ptr{Cvoid}, Int64, Int64, Int64, Int32, Ptr{Cvoid}, Int64, Float64, UInt64, UInt32, UInt32, Ptr{Cvoid},
                   Ptr{Cvoid}, Ptr{Cvoid}),
                  net.params.ptr, net.n_in, net.hidden, net.n_out, net.act, obs.ptr, env.n, ϵ, p.explorer_seed,
                  env.env_id_base, step, p.actions.ptr, p.q.ptr, stream()))
    else
        chk(ccall((:rlhip_dqn3_plan_f32, LIB), Int32,
                  (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Int32, Ptr{Cvoid}, Int64, Float64, UInt64, UInt32, UInt32,
                   Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                  net.params.ptr, net.packed.ptr, net.n_in, net.hidden, net.n_out, net.act, obs.ptr, env.n, ϵ,
                  p.explorer_seed, env.env_id_base, step, p.actions.ptr, p.q.ptr, stream()))
    end
    p.actions
end
optimise!(p::HipQBasedPolicy, s::PostActStage, t::HipTrajectory) = optimise!(p.learner, s, t)   # q_based_policy.jl:49
optimise!(agent::Agent{<:HipQBasedPolicy,<:HipTrajectory}, s::PostActStage) = optimise!(agent.policy, s, agent.trajectory)
optimise!(::Agent{<:HipQBasedPolicy,<:HipTrajectory}, ::Any) = nothing

# ------------------------------------------------------------------------------------------------------------------
# HipPPOPolicy (removed Zoo PPOPolicy; hyper-parameters blog index.html:15257-15287): ActorCritic(actor ns->h->na_out,
#   critic ns->h->1), trajectory of `update_freq` vec-steps in HBM, GAE, clipped surrogate, Adam
# ------------------------------------------------------------------------------------------------------------------
mutable struct HipPPOPolicy <: AbstractPolicy
    kind::Int32
    cfg::Base.RefValue{PPOCfg}
    n::Int; T::Int; ns::Int; na::Int
    seed::UInt64
    params::DevBuf{Float32}; m::DevBuf{Float32}; v::DevBuf{Float32}; beta_pow::DevBuf{Float32}
    grad::DevBuf{Float32}; losses::DevBuf{Float32}; workspace::DevBuf{UInt8}
    obs::DevBuf{Float32}; logp::DevBuf{Float32}; value::DevBuf{Float32}; rew::DevBuf{Float32}; adv::DevBuf{Float32}
    ret::DevBuf{Float32}; action_f::DevBuf{Float32}; action_i::DevBuf{Int32}; terminal::DevBuf{UInt8}
    traj::Base.RefValue{PPOTraj}
    a_i::DevBuf{Int32}; a_f::DevBuf{Float32}; lp::DevBuf{Float32}; val::DevBuf{Float32}    # per-step plan! outputs
    vec_step::UInt32; update_ctr::UInt32; n_pushed::Int
    comm::Any        # nothing or a HipComm: the sharded learner (SURVEY 8e)
    fused::Bool      # `run` drives one launch per update period instead of the per-step stages (see `_run` below)
end
function HipPPOPolicy(env::HipVecEnv; update_freq = 32, seed = env.seed, comm = nothing, fused = false, kwargs...)
    c = Ref{PPOCfg}()
    chk(ccall((:rlhip_ppo_default, LIB), Int32, (Ref{PPOCfg},), c))
    cfg = Ref(with_kwargs(c[]; continuous = is_continuous(env), kwargs...))
    np = ccall((:rlhip_ppo_nparams, LIB), Int64, (Int32, Ref{PPOCfg}), env.kind, cfg)
    np > 0 || chk(Int32(-1))
    n, T, ns = env.n, Int(update_freq), obs_dim(env.kind)
    na = is_continuous(env) ? 1 : length(action_space(env))
    nout = is_continuous(env) ? 2na : na
    h = Int(cfg[].hidden)
    params = DevBuf{Float32}(np)
    np_actor = cfg[].layers == 3 ? ccall((:rlhip_mlp3_nparams, LIB), Int64, (Int64, Int64, Int64), ns, h, nout) :
                                   ccall((:rlhip_mlp2_nparams, LIB), Int64, (Int64, Int64, Int64), ns, h, nout)
    for (net_id, off, no) in ((0, 0, nout), (1, np_actor, 1))      # actor net_id 0, critic net_id 1 (Philox INIT stream)
        if cfg[].layers == 3
            chk(ccall((:rlhip_mlp3_init_f32, LIB), Int32, (Ptr{Cvoid}, Int64, Int64, Int64, UInt64, UInt32, Ptr{Cvoid}),
                      offset(params, off), ns, h, no, seed, net_id, stream()))
        else
            chk(ccall((:rlhip_mlp2_init_f32, LIB), Int32, (Ptr{Cvoid}, Int64, Int64, Int64, UInt64, UInt32, Ptr{Cvoid}),
                      offset(params, off), ns, h, no, seed, net_id, stream()))
        end
    end
    ws = ccall((:rlhip_ppo_workspace_bytes, LIB), Int64, (Int32, Ref{PPOCfg}, Int64, Int64), env.kind, cfg, n, T)
    wsbuf = DevBuf{UInt8}(ws)
    # registers the size: a later call whose (n, T) needs more than `ws` bytes is an ArgumentError, not a write past the buffer
    chk(ccall((:rlhip_ppo_workspace_init, LIB), Int32, (Ptr{Cvoid}, Int64, Ptr{Cvoid}), wsbuf.ptr, ws, stream()))
    finalizer(b -> ccall((:rlhip_ppo_workspace_release, LIB), Int32, (Ptr{Cvoid},), b.ptr), wsbuf)
    obs, logp, value = DevBuf{Float32}((T + 1) * ns * n), DevBuf{Float32}(T * n), DevBuf{Float32}((T + 1) * n)
    rew, adv, ret = DevBuf{Float32}(T * n), DevBuf{Float32}(T * n), DevBuf{Float32}(T * n)
    af, ai, term = DevBuf{Float32}(T * na * n), DevBuf{Int32}(T * n), DevBuf{UInt8}(T * n)
    HipPPOPolicy(env.kind, cfg, n, T, ns, na, UInt64(seed), params, DevBuf{Float32}(np), DevBuf{Float32}(np),
                 to_dev!(DevBuf{Float32}(2), Float32[cfg[].beta1, cfg[].beta2]), DevBuf{Float32}(np), DevBuf{Float32}(4),
                 wsbuf, obs, logp, value, rew, adv, ret, af, ai, term,
                 Ref(PPOTraj(obs.ptr, logp.ptr, value.ptr, rew.ptr, adv.ptr, ret.ptr, af.ptr, ai.ptr, term.ptr)),
                 DevBuf{Int32}(n), DevBuf{Float32}(n), DevBuf{Float32}(n), DevBuf{Float32}(n), UInt32(0), UInt32(0), 0, comm, fused)
end

"plan!(policy, env): actor + critic forward, Gumbel-max / Gaussian sampling, log-prob -- one launch
(sample_categorical RLCore/utils/networks.jl:425-432; normlogpdf RLCore/utils/distributions.jl:18-21)"
function plan!(p::HipPPOPolicy, env::HipVecEnv)
    chk(ccall((:rlhip_ppo_plan_f32, LIB), Int32,
              (Int32, Ref{PPOCfg}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, UInt64, UInt32, UInt32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
               Ptr{Cvoid}, Ptr{Cvoid}),
              p.kind, p.cfg, p.params.ptr, device_state(env).ptr, env.n, p.seed, env.env_id_base, p.vec_step, p.a_i.ptr,
              p.a_f.ptr, p.lp.ptr, p.val.ptr, stream()))
    is_continuous(env) ? p.a_f : p.a_i
end
d2d!(dst::Ptr{Cvoid}, src::Ptr{Cvoid}, bytes) =
    chk(ccall((:rlhip_memcpy_d2d, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), dst, src, bytes, stream()))
"PreActStage push: (state, action, action_log_prob) + value of the step about to be taken (blog index.html:15280-15286)"
function Base.push!(p::HipPPOPolicy, ::PreActStage, env::HipVecEnv)
    cont = is_continuous(env)
    chk(ccall((:rlhip_ppo_push_preact_f32, LIB), Int32,
              (Ref{PPOTraj}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
              p.traj, p.n_pushed, p.ns, p.n, device_state(env).ptr, p.val.ptr, p.lp.ptr, cont ? C_NULL : p.a_i.ptr,
              cont ? p.a_f.ptr : C_NULL, stream()))
    nothing
end
"PostActStage push: reward, terminal"
function Base.push!(p::HipPPOPolicy, ::PostActStage, env::HipVecEnv, action = nothing)
    chk(ccall((:rlhip_ppo_push_postact_f32, LIB), Int32, (Ref{PPOTraj}, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
              p.traj, p.n_pushed, p.n, env.rew.ptr, env.done.ptr, stream()))
    p.n_pushed += 1
    p.vec_step += 1
    nothing
end
Base.push!(::HipPPOPolicy, ::Union{PreExperimentStage,PostExperimentStage,PreEpisodeStage,PostEpisodeStage}, ::HipVecEnv) = nothing

"T vec-steps of plan!/push!/act!/push! in ONE launch (+ the GAE scan of every env): the fused form of the loop body"
function rollout!(p::HipPPOPolicy, env::HipVecEnv)
    chk(ccall((:rlhip_ppo_rollout_f32, LIB), Int32,
              (Int32, Ptr{Cvoid}, Ref{EnvState}, Int64, Int64, Ref{PPOCfg}, Ptr{Cvoid}, UInt64, UInt32, UInt32, Ref{PPOTraj},
               Ptr{Cvoid}),
              p.kind, env.cfg, env.st, env.n, p.T, p.cfg, p.params.ptr, p.seed, env.env_id_base, p.vec_step, p.traj,
              stream()))
    env.obs_valid = false
    p.vec_step += p.T
    p.n_pushed = p.T
    nothing
end

"optimise!(policy, PostActStage): when `update_freq` vec-steps are stored -- bootstrap value, GAE
(generalized_advantage_estimation RLCore/utils/basic.jl:334-417), then n_epochs x n_microbatches of gradient ->
[gradient exchange] -> clip_by_global_norm! -> Adam.  ONE ccall enqueues the whole update."
function optimise!(p::HipPPOPolicy, ::PostActStage, env::HipVecEnv; fused_rollout = false)
    p.n_pushed == p.T || return false
    if !fused_rollout           # per-step protocol: bootstrap state / value of step T + 1, then the scan
        plan!(p, env)
        chk(ccall((:rlhip_ppo_push_preact_f32, LIB), Int32,
                  (Ref{PPOTraj}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                  p.traj, p.T, p.ns, p.n, device_state(env).ptr, p.val.ptr, C_NULL, C_NULL, C_NULL, stream()))
        chk(ccall((:rlhip_ppo_gae_f32, LIB), Int32, (Ref{PPOCfg}, Int64, Int64, Ref{PPOTraj}, Ptr{Cvoid}),
                  p.cfg, p.n, p.T, p.traj, stream()))
    end
    if p.comm === nothing
        chk(ccall((:rlhip_ppo_update_f32, LIB), Int32,
                  (Int32, Ref{PPOCfg}, Int64, Int64, Ref{PPOTraj}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt64,
                   UInt32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                  p.kind, p.cfg, p.n, p.T, p.traj, p.params.ptr, p.m.ptr, p.v.ptr, p.beta_pow.ptr, p.seed, p.update_ctr,
                  p.workspace.ptr, p.grad.ptr, p.losses.ptr, stream()))
    else                        # sharded learner: the exchange (peer-to-peer kernel or ncclAllReduce) is inside the call
        chk(ccall((:rlhip_ppo_update_comm_f32, LIB), Int32,
                  (Int32, Ref{PPOCfg}, Int64, Int64, Ref{PPOTraj}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt64,
                   UInt32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                  p.kind, p.cfg, p.n, p.T, p.traj, p.params.ptr, p.m.ptr, p.v.ptr, p.beta_pow.ptr, p.seed, p.update_ctr,
                  p.workspace.ptr, p.grad.ptr, p.losses.ptr, p.comm.h, stream()))
        check(p.comm)
    end
    p.update_ctr += 1
    p.n_pushed = 0
    true
end

# ------------------------------------------------------------------------------------------------------------------
# HipComm: the sharded learner's collective (csrc/comm.hip).  `transport` is ANY function that all-gathers a byte vector
# over the ranks (MPI.Allgather, Distributed.jl remote calls, a shared directory ...): Vector{UInt8} -> Vector{Vector{UInt8}}
# ------------------------------------------------------------------------------------------------------------------
mutable struct HipComm
    h::Ptr{Cvoid}
    rank::Int
    world::Int
end
function HipComm(rank::Integer, world::Integer, cap::Integer, allgather::Function; use_rccl = true)
    uid = zeros(UInt8, 128)
    if use_rccl
        rank == 0 && chk(ccall((:rlhip_comm_unique_id, LIB), Int32, (Ptr{UInt8},), uid))
        uid = allgather(uid)[1]                         # rank 0's id on every rank
    end
    h = Ref{Ptr{Cvoid}}(C_NULL)
    chk(ccall((:rlhip_comm_init, LIB), Int32, (Int32, Int32, Ptr{UInt8}, Int64, Ref{Ptr{Cvoid}}),
              rank, world, use_rccl ? pointer(uid) : Ptr{UInt8}(C_NULL), cap, h))
    handle, dev = zeros(UInt8, 64), Ref{Int32}(0)
    chk(ccall((:rlhip_comm_export, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Ref{Int32}), h[], handle, dev))
    recs = allgather(vcat(handle, reinterpret(UInt8, Int32[dev[]])))
    handles = reduce(vcat, (r[1:64] for r in recs))
    devices = Int32[reinterpret(Int32, r[65:68])[1] for r in recs]
    active = Ref{Int32}(0)
    chk(ccall((:rlhip_p2p_setup, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Ptr{Int32}, Ref{Int32}), h[], handles, devices, active))
    c = HipComm(h[], rank, world)
    active[] == 0 && world > 1 && @warn "rlhip: peer-to-peer gradient exchange not active: $(why(c)) -> ncclAllReduce"
    c
end
function info(c::HipComm)
    d = Ref{CommDesc}()
    chk(ccall((:rlhip_comm_info, LIB), Int32, (Ptr{Cvoid}, Ref{CommDesc}), c.h, d))
    d[]
end
why(c::HipComm) = unsafe_string(pointer(collect(info(c).why)))
"in-place SUM of the flat gradient over the ranks, on the compute stream -- before clip_by_global_norm! (basic.jl:19-29)"
allreduce_grads!(c::HipComm, g::DevBuf{Float32}) =
    chk(ccall((:rlhip_allreduce_grads, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}), c.h, g.ptr, g.n, stream()))
"throws RLHipError(-4) if a peer never arrived at an exchange (the result was NaN-poisoned): no synchronisation"
check(c::HipComm) = chk(ccall((:rlhip_comm_check, LIB), Int32, (Ptr{Cvoid},), c.h))
"collective tear-down in two phases: barrier (the host's own) -> unmap!(c) on every rank -> barrier -> destroy!(c)"
unmap!(c::HipComm) = (synchronize(); chk(ccall((:rlhip_comm_unmap, LIB), Int32, (Ptr{Cvoid},), c.h)))
destroy!(c::HipComm) = (synchronize(); chk(ccall((:rlhip_comm_destroy, LIB), Int32, (Ptr{Cvoid},), c.h)); c.h = C_NULL)

# ------------------------------------------------------------------------------------------------------------------
# scans and updates on device matrices: the pure functions of RLCore/utils/basic.jl
# ------------------------------------------------------------------------------------------------------------------
"generalized_advantage_estimation(rewards, values, γ, λ; dims, terminal)  basic.jl:334-417 (column-major n1 x n2)"
function gae!(adv::DevBuf{Float32}, r::DevBuf{Float32}, v::DevBuf{Float32}, n1, n2, γ::Float32, λ::Float32;
              terminal::Union{Nothing,DevBuf{UInt8}} = nothing, dims = 2)
    chk(ccall((:rlhip_gae_f32, LIB), Int32,
              (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Float32, Float32, Ptr{Cvoid}, Int32, Ptr{Cvoid}),
              adv.ptr, r.ptr, v.ptr, n1, n2, γ, λ, terminal === nothing ? C_NULL : terminal.ptr, dims, stream()))
    adv
end
"discount_rewards(rewards, γ; dims, terminal, init)  basic.jl:138-235"
function discount_rewards!(out::DevBuf{Float32}, r::DevBuf{Float32}, n1, n2, γ::Float32; terminal = nothing, init = nothing, dims = 0)
    chk(ccall((:rlhip_discount_rewards_f32, LIB), Int32,
              (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Float32, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}),
              out.ptr, r.ptr, n1, n2, γ, terminal === nothing ? C_NULL : terminal.ptr, init === nothing ? C_NULL : init.ptr,
              dims, stream()))
    out
end
"clip_by_global_norm!(gs, ps, clip_norm)  basic.jl:19-29 over the flat gradient; returns the norm"
function clip_by_global_norm!(g::DevBuf{Float32}, clip_norm::Float32)
    gn = DevBuf{Float32}(1)
    chk(ccall((:rlhip_clip_by_global_norm_f32, LIB), Int32, (Ptr{Cvoid}, Int64, Float32, Ptr{Cvoid}, Ptr{Cvoid}),
              g.ptr, g.n, clip_norm, gn.ptr, stream()))
    to_host(gn)[1]
end

# ------------------------------------------------------------------------------------------------------------------
# SURVEY 8f rows: prioritized replay, frame stacking at sample time, explorers, stochastic heads (device buffers in / out)
# ------------------------------------------------------------------------------------------------------------------
# CircularPrioritizedTraces + prioritized BatchSampler (RLTrajectories 0.4): `tree` = zeroed DevBuf{Float32}(sumtree_nodes(n))
sumtree_nodes(n_leaves) = ccall((:rlhip_sumtree_nodes, LIB), Int64, (Int64,), n_leaves)
push_priority!(t::HipTrajectory, tree::DevBuf{Float32}, p::Float32) = chk(ccall((:rlhip_ring_push_priority, LIB), Int32,
    (Ref{Ring}, Ptr{Cvoid}, Float32, Ptr{Cvoid}), t.rb, tree.ptr, p, stream()))
"`inds, priorities = rand(rng, sumtree, batchsize)`: logical indices for the gather, physical keys for the write-back"
sample_prioritized!(idx::DevBuf{Int64}, key::DevBuf{Int64}, prio::DevBuf{Float32}, t::HipTrajectory, tree, batch, seed, ctr) =
    chk(ccall((:rlhip_ring_sample_prioritized, LIB), Int32,
              (Ref{Ring}, Ptr{Cvoid}, Int64, UInt64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
              t.rb, tree.ptr, batch, seed, ctr, idx.ptr, key.ptr, prio.ptr, stream()))
"the same draw AND the gather of its batch in one launch (outputs as `sample_prioritized!` + the ring gather)"
sample_gather_prioritized!(idx::DevBuf{Int64}, key::DevBuf{Int64}, prio::DevBuf{Float32}, t::HipTrajectory, tree, batch, seed,
                           ctr, s, a, r, term, s_next) =
    chk(ccall((:rlhip_ring_sample_gather_prioritized, LIB), Int32,
              (Ref{Ring}, Ptr{Cvoid}, Int64, UInt64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
               Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
              t.rb, tree.ptr, batch, seed, ctr, idx.ptr, key.ptr, prio.ptr, s.ptr, a.ptr, r.ptr, term.ptr, s_next.ptr, stream()))
"priority write-back value (|td| + eps)^alpha of PrioritizedDQN"
per_priority!(out::DevBuf{Float32}, td::DevBuf{Float32}, eps::Float32, alpha::Float32) =
    chk(ccall((:rlhip_per_priority_f32, LIB), Int32, (Ptr{Cvoid}, Int64, Float32, Float32, Ptr{Cvoid}, Ptr{Cvoid}),
              td.ptr, td.n, eps, alpha, out.ptr, stream()))
"importance-sampling weights of the sampled batch: w = 1 ./ ((priority .+ 1f-10) .^ β); w ./= maximum(w)  (PrioritizedDQN)"
is_weights!(w::DevBuf{Float32}, prio::DevBuf{Float32}, β::Float32) =
    chk(ccall((:rlhip_per_is_weights_f32, LIB), Int32, (Ptr{Cvoid}, Int64, Float32, Ptr{Cvoid}, Ptr{Cvoid}),
              prio.ptr, prio.n, β, w.ptr, stream()))
"the DQN gradient on explicit indices with importance-sampling weights: loss = mean(w .* huber(td)) (2-layer Q-network)"
dqn_grad_weighted!(t::HipTrajectory, h, na, act, params::DevBuf{Float32}, target::DevBuf{Float32}, batch, idx::DevBuf{Int64},
                   w::DevBuf{Float32}, γ::Float32, δ::Float32, workspace, grad::DevBuf{Float32}, loss::DevBuf{Float32},
                   td::DevBuf{Float32}) =
    chk(ccall((:rlhip_dqn_grad_idx_w_f32, LIB), Int32,
              (Ref{Ring}, Int64, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Ptr{Cvoid},
               Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
              t.rb, h, na, act, params.ptr, target.ptr, batch, idx.ptr, w.ptr, γ, δ, workspace.ptr, grad.ptr, loss.ptr, td.ptr,
              stream()))
"the same for the 3-layer (bf16 MFMA) Q-network"
dqn3_grad_weighted!(t::HipTrajectory, h, na, act, params::DevBuf{Float32}, packed::DevBuf{UInt16}, target::DevBuf{Float32},
                    tpacked::DevBuf{UInt16}, batch, idx::DevBuf{Int64}, w::DevBuf{Float32}, γ::Float32, δ::Float32, workspace,
                    grad::DevBuf{Float32}, loss::DevBuf{Float32}, td::DevBuf{Float32}) =
    chk(ccall((:rlhip_dqn3_grad_w_f32, LIB), Int32,
              (Ref{Ring}, Int64, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Float32,
               Float32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
              t.rb, h, na, act, params.ptr, packed.ptr, target.ptr, tpacked.ptr, batch, idx.ptr, w.ptr, γ, δ, workspace.ptr, grad.ptr,
              loss.ptr, td.ptr, stream()))
"trajectory[:priority, keys] = p   (0-based physical leaf keys from the sampler)"
set_priority!(tree::DevBuf{Float32}, n_leaves, key::DevBuf{Int64}, p::DevBuf{Float32}, n) =
    chk(ccall((:rlhip_sumtree_update, LIB), Int32, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
              tree.ptr, n_leaves, key.ptr, p.ptr, n, stream()))
# StackFrames (RLCore/utils/stack_frames.jl:11-44) applied at sample time + AtariEnv's 2-frame max-pool (atari.jl:104-107)
gather_stacked!(t::HipTrajectory, idx::DevBuf{Int64}, batch, n_stack, s, a, r, term, s_next) =
    chk(ccall((:rlhip_ring_gather_stacked, LIB), Int32,
              (Ref{Ring}, Ptr{Cvoid}, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
              t.rb, idx.ptr, batch, n_stack, s.ptr, a.ptr, r.ptr, term.ptr, s_next.ptr, stream()))
push_maxpool!(t::HipTrajectory, screen1::DevBuf{UInt8}, screen2::DevBuf{UInt8}, a::DevBuf{Int32}, r::DevBuf{Float32},
              term::DevBuf{UInt8}) =
    chk(ccall((:rlhip_ring_push_transition_maxpool, LIB), Int32,
              (Ref{Ring}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
              t.rb, screen1.ptr, screen2.ptr, a.ptr, r.ptr, term.ptr, stream()))
# plan!(explorer, values[, mask]) on a device (na, n) SoA matrix: EpsilonGreedy (epsilon_greedy_explorer.jl:102-131),
# Weighted / WeightedSoftmax / GumbelSoftmax (kind 0 / 1 / 2), UCB; BatchExplorer semantics (one column per env)
plan_eps_greedy!(actions::DevBuf{Int32}, values::DevBuf{Float32}, na, n, ϵ; mask = nothing, is_break_tie = false, seed = 0,
                 env_id_base = 0, step = 1) =
    chk(ccall((:rlhip_eps_greedy_select_f32, LIB), Int32,
              (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Ptr{Cvoid}, Float64, Int32, UInt64, UInt32, UInt32, Ptr{Cvoid}, Ptr{Cvoid}),
              values.ptr, na, n, n, 1, mask === nothing ? C_NULL : mask.ptr, ϵ, is_break_tie, seed, env_id_base, step,
              actions.ptr, stream()))
plan_explorer!(kind::Integer, actions::DevBuf{Int32}, values::DevBuf{Float32}, na, n; mask = nothing, is_normalized = false,
               seed = 0, env_id_base = 0, step = 1) =
    chk(ccall((:rlhip_explorer_select_f32, LIB), Int32,
              (Int32, Ptr{Cvoid}, Int64, Int64, Int64, Int64, Ptr{Cvoid}, Int32, UInt64, UInt32, UInt32, Ptr{Cvoid}, Ptr{Cvoid}),
              kind, values.ptr, na, n, n, 1, mask === nothing ? C_NULL : mask.ptr, is_normalized, seed, env_id_base, step,
              actions.ptr, stream()))
plan_ucb!(actions::DevBuf{Int32}, values::DevBuf{Float32}, counts::DevBuf{Float64}, na, n, c, step; seed = 0, env_id_base = 0) =
    chk(ccall((:rlhip_ucb_select_f32, LIB), Int32,
              (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Float64, Ptr{Cvoid}, Int64, UInt64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}),
              values.ptr, na, n, n, 1, c, counts.ptr, step, seed, env_id_base, actions.ptr, stream()))
"sample_categorical (Gumbel-max, RLCore/utils/networks.jl:425-432) with optional mask (:466-468)"
sample_categorical!(actions::DevBuf{Int32}, logp::DevBuf{Float32}, logits::DevBuf{Float32}, na, n; mask = nothing, seed = 0,
                    env_id_base = 0, step = 0) =
    chk(ccall((:rlhip_categorical_sample_f32, LIB), Int32,
              (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Ptr{Cvoid}, UInt64, UInt32, UInt32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
              logits.ptr, na, n, n, 1, mask === nothing ? C_NULL : mask.ptr, seed, env_id_base, step, actions.ptr, logp.ptr,
              stream()))
"(gn::GaussianNetwork)(rng, s; is_sampling = true, is_return_log_prob = true) on the mu / sigma heads' outputs (d x n),
K samples per state; squash = tanh, soft = SoftGaussianNetwork  (RLCore/utils/networks.jl:64-116,147-198)"
gaussian_sample!(action::DevBuf{Float32}, logp::DevBuf{Float32}, mu::DevBuf{Float32}, raw_sigma::DevBuf{Float32}, d, n, K;
                 min_sigma = 0f0, max_sigma = Inf32, squash = false, soft = false, seed = 0, env_id_base = 0, step = 0) =
    chk(ccall((:rlhip_gaussian_head_sample_f32, LIB), Int32,
              (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Float32, Float32, Int32, Int32, UInt64, UInt32, UInt32, Ptr{Cvoid},
               Ptr{Cvoid}, Ptr{Cvoid}),
              mu.ptr, raw_sigma.ptr, d, n, K, min_sigma, max_sigma, squash, soft, seed, env_id_base, step, action.ptr, logp.ptr,
              stream()))
"(gn::GaussianNetwork)(s, action): log-probability of given (squashed) actions"
gaussian_logp!(logp::DevBuf{Float32}, mu::DevBuf{Float32}, raw_sigma::DevBuf{Float32}, action::DevBuf{Float32}, d, n, K;
               min_sigma = 0f0, max_sigma = Inf32, squash = false, soft = false) =
    chk(ccall((:rlhip_gaussian_head_logp_f32, LIB), Int32,
              (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Float32, Float32, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}),
              mu.ptr, raw_sigma.ptr, action.ptr, d, n, K, min_sigma, max_sigma, squash, soft, logp.ptr, stream()))

# ---- the same rows on the REFERENCE's types (round 4): what a user of the reference writes keeps working when the values /
# traces live on the device.  These are methods, not new entry points: each one forwards to a free function above.

"""
    HipPrioritizedTraces(traces::HipTrajectory; default_priority = 100f0)

`CircularPrioritizedTraces(CircularArraySARTSTraces(...); default_priority)` of RLTrajectories 0.4 with the priorities in a
device sum-tree: `push!` gives every new transition `default_priority`, `sample(traces, batchsize)` is the prioritized
`BatchSampler` draw (`inds, priorities = rand(rng, sumtree, batchsize)`) and `traces[:priority, keys] = p` the write-back.
"""
mutable struct HipPrioritizedTraces
    traces::HipTrajectory
    tree::DevBuf{Float32}
    n_leaves::Int
    default_priority::Float32
    seed::UInt64
    draw_ctr::UInt32
end
function HipPrioritizedTraces(t::HipTrajectory; default_priority = 100f0, seed = 0)
    default_priority > 0 || throw(ArgumentError("default_priority must be > 0"))
    n = Int(t.rb.capacity * t.rb.n_env)
    HipPrioritizedTraces(t, DevBuf{Float32}(sumtree_nodes(n)), n, Float32(default_priority), UInt64(seed), UInt32(0))
end
Base.length(p::HipPrioritizedTraces) = length(p.traces)
Base.push!(p::HipPrioritizedTraces, x::NamedTuple{(:state,)}) = (push!(p.traces, x); p)
"push!(traces, (state = s', action, reward, terminal)): the trajectory's push, then the new leaves := default_priority"
function Base.push!(p::HipPrioritizedTraces, x::NamedTuple{(:state, :action, :reward, :terminal)})
    push!(p.traces, x)
    push_priority!(p.traces, p.tree, p.default_priority)
    p
end
"the prioritized BatchSampler: (inds for the gather, keys for the write-back, priorities) as device buffers"
function sample(p::HipPrioritizedTraces, batchsize::Integer)
    idx, key, prio = DevBuf{Int64}(batchsize), DevBuf{Int64}(batchsize), DevBuf{Float32}(batchsize)
    sample_prioritized!(idx, key, prio, p.traces, p.tree, batchsize, p.seed, p.draw_ctr)
    p.draw_ctr += UInt32(1)
    (inds = idx, key = key, priority = prio)
end
"trajectory[:priority, keys] = p  (sequential semantics: the last duplicate key wins)"
function Base.setindex!(p::HipPrioritizedTraces, v::DevBuf{Float32}, name::Symbol, keys::DevBuf{Int64})
    name === :priority || throw(ArgumentError("only the :priority trace can be assigned"))
    keys.n == v.n || throw(DimensionMismatch("keys and priorities differ in length"))
    set_priority!(p.tree, p.n_leaves, keys, v, keys.n)
    v
end

"""
    HipStackFrames(n_stack)

`StackFrames(T, d..., n_stack)` (RLCore/src/utils/stack_frames.jl:11-44) moved from the way INTO the trajectory to the way
OUT: the ring stores single frames (4x less HBM for n_stack = 4); `sample(sf, traces, inds)` returns what the reference's
agent would have stored -- the last n_stack frames ending at each index, zero frames before an episode's first observation
(`reset!(::StackFrames)` fills the buffer with zeros, :36-39).
"""
struct HipStackFrames
    n_stack::Int
end
function sample(sf::HipStackFrames, t::HipTrajectory, inds::DevBuf{Int64})
    b, fb = inds.n, Int(t.rb.obs_dim)
    T = t.rb.elem_bytes == 1 ? UInt8 : Float32
    s, sn = DevBuf{T}(b * sf.n_stack * fb), DevBuf{T}(b * sf.n_stack * fb)
    a, r, term = DevBuf{Int32}(b), DevBuf{Float32}(b), DevBuf{UInt8}(b)
    gather_stacked!(t, inds, b, sf.n_stack, s, a, r, term, sn)
    (state = s, action = a, reward = r, terminal = term, next_state = sn)
end

# plan!(explorer, values[, mask]) with `values` a device (na, n) SoA matrix = one column per env: the BatchExplorer form
# (`[x.explorer(v) for v in eachcol(values)]`, batch_explorer.jl:15-18).  The explorer object keeps its hyper-parameters;
# its `rng` field is replaced by the shared Philox streams (seed / step keywords), like every other draw of this path.
struct DevValues          # a device value matrix and its shape (na actions x n envs), 1-based actions come back
    values::DevBuf{Float32}
    na::Int
    n::Int
end
_plan_dev(kind, v::DevValues, mask; is_normalized = false, seed = 0, env_id_base = 0, step = 1) = begin
    a = DevBuf{Int32}(v.n)
    plan_explorer!(kind, a, v.values, v.na, v.n; mask = mask, is_normalized = is_normalized, seed = seed,
                   env_id_base = env_id_base, step = step)
    a   # 0-based on the device; `to_host(a) .+ 1` are the reference's action indices
end
plan!(s::ReinforcementLearningCore.WeightedExplorer{N}, v::DevValues, mask = nothing; kw...) where {N} =
    _plan_dev(0, v, mask; is_normalized = N, kw...)                       # weighted_explorer.jl:25-34
plan!(s::ReinforcementLearningCore.WeightedSoftmaxExplorer, v::DevValues, mask = nothing; kw...) =
    _plan_dev(1, v, mask; kw...)                                          # weighted_softmax_explorer.jl:20-26
plan!(s::ReinforcementLearningCore.GumbelSoftmaxExplorer, v::DevValues, mask = nothing; kw...) =
    _plan_dev(2, v, mask; kw...)                                          # gumbel_softmax_explorer.jl:12-22
function plan!(s::EpsilonGreedyExplorer{<:Any,TIE}, v::DevValues, mask = nothing; seed = 0, env_id_base = 0) where {TIE}
    ϵ = get_ϵ(s)                                                          # epsilon_greedy_explorer.jl:69-90
    s.step += 1                                                           # :104, :119: one explorer step per vec-step
    a = DevBuf{Int32}(v.n)
    plan_eps_greedy!(a, v.values, v.na, v.n, ϵ; mask = mask, is_break_tie = TIE, seed = seed, env_id_base = env_id_base,
                     step = s.step)
    a
end
"""
    prob(s::EpsilonGreedyExplorer, v::DevValues[, mask]) -> DevBuf{Float64}

`RLBase.prob(s, values[, mask])` (epsilon_greedy_explorer.jl:141-194) for every column of a device `(na, n)` matrix: the
Float64 probability vectors the reference wraps in `Categorical(probs; check_args = false)`, as a device `(na, n)` SoA matrix
(`Categorical(to_host(p)[k:n:end]; check_args = false)` is env k's distribution).  `mask`: device `(na, n)` UInt8 matrix.
"""
function RLBase.prob(s::EpsilonGreedyExplorer{<:Any,TIE}, v::DevValues, mask = nothing) where {TIE}
    p = DevBuf{Float64}(v.na * v.n)
    chk(ccall((:rlhip_eps_greedy_prob_f32, LIB), Int32,
              (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Ptr{Cvoid}, Float64, Int32, Ptr{Cvoid}, Ptr{Cvoid}),
              v.values.ptr, v.na, v.n, v.n, 1, mask === nothing ? C_NULL : mask.ptr, get_ϵ(s), TIE, p.ptr, stream()))
    p
end
"UCBExplorer (UCB_explorer.jl:24-28): `counts` is the explorer's per-env action counter on the device (na, n), Float64"
function plan!(s::ReinforcementLearningCore.UCBExplorer, v::DevValues, counts::DevBuf{Float64}; seed = 0, env_id_base = 0)
    a = DevBuf{Int32}(v.n)
    plan_ucb!(a, v.values, counts, v.na, v.n, s.c, s.step; seed = seed, env_id_base = env_id_base)
    s.step += 1
    a
end
plan!(x::ReinforcementLearningCore.BatchExplorer, v::DevValues, args...; kw...) = plan!(x.explorer, v, args...; kw...)

# ------------------------------------------------------------------------------------------------------------------
# hooks: TotalRewardPerEpisode + BatchStepsPerEpisode (RLCore/core/hooks.jl:146-231) with device accumulators
# ------------------------------------------------------------------------------------------------------------------
mutable struct HipEpisodeStats <: AbstractHook
    n::Int; cap::Int; vec_step::UInt32
    steps_acc::DevBuf{Int32}; ret_acc::DevBuf{Float64}; log::DevBuf{UInt8}; count::DevBuf{UInt32}
end
HipEpisodeStats(n; log_capacity = 1 << 20) = HipEpisodeStats(n, log_capacity, UInt32(0), DevBuf{Int32}(n), DevBuf{Float64}(n),
                                                             DevBuf{UInt8}(24 * log_capacity), DevBuf{UInt32}(1))
function Base.push!(h::HipEpisodeStats, ::PostActStage, policy, env::HipVecEnv)
    chk(ccall((:rlhip_hook_episode_stats, LIB), Int32,
              (Ptr{Cvoid}, Ptr{Cvoid}, Int64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Ptr{Cvoid}, Ptr{Cvoid}),
              env.rew.ptr, env.done.ptr, h.n, h.vec_step, h.steps_acc.ptr, h.ret_acc.ptr, h.log.ptr, h.cap, h.count.ptr,
              stream()))
    h.vec_step += 1
    nothing
end

# ------------------------------------------------------------------------------------------------------------------
# run(policy, env, stop_condition, hook): ONE `_run` method for the vector env; RLCore/src/core/run.jl is untouched.
# The MultiThreadEnv specialisation (blog index.md:351-374): no episode stages -- instances auto-reset in the kernel.
# ------------------------------------------------------------------------------------------------------------------
function _run(policy::AbstractPolicy, env::HipVecEnv, stop_condition, hook, reset_condition)
    push!(hook, PreExperimentStage(), policy, env)
    push!(policy, PreExperimentStage(), env)
    while true
        action = plan!(policy, env)                        # run.jl:57
        push!(policy, PreActStage(), env)
        push!(hook, PreActStage(), policy, env)
        act!(env, action)                                  # run.jl:58
        push!(policy, PostActStage(), env, action)         # run.jl:60
        optimise!(policy, PostActStage())                  # run.jl:61
        push!(hook, PostActStage(), policy, env)
        check!(stop_condition, policy, env) && break       # run.jl:64
    end
    push!(policy, PostExperimentStage(), env)
    push!(hook, PostExperimentStage(), policy, env)
    hook
end
optimise!(p::HipPPOPolicy, s::PostActStage) = nothing      # (the env is needed: the PPO agent below passes it)

"fast path: the whole loop body for Agent{HipQBasedPolicy} as ONE ccall per vec-step (rlhip_dqn_vec_step_f32) -- same
kernels, same order, same counters as the generic loop above (bit-identical: tests/test_gpu_run.py, test_gpu_abi_host.py)"
function _run(agent::Agent{<:HipQBasedPolicy,<:HipTrajectory}, env::HipVecEnv{K,Float32}, stop_condition, hook,
              reset_condition) where {K}
    p, t = agent.policy, agent.trajectory
    L, tn = p.learner, p.learner.approximator
    net = tn.network
    push!(hook, PreExperimentStage(), agent, env)
    push!(agent, PreExperimentStage(), env)
    p.actions === nothing && (p.actions = DevBuf{Int32}(env.n); p.q = DevBuf{Float32}(net.n_out * env.n))
    a = DqnStepArgs()
    a.kind = env.kind; a.env_cfg = Base.unsafe_convert(Ptr{Cvoid}, env.cfg); a.st = Base.unsafe_convert(Ptr{Cvoid}, env.st)
    a.n = env.n; a.env_seed = env.seed; a.env_id_base = env.env_id_base
    a.obs = device_state(env).ptr; a.last_obs = env.last_obs.ptr; a.ring = pointer_from_objref(t.rb)
    a.layers = net.layers; a.h = net.hidden; a.na = net.n_out; a.act = net.act
    a.params = net.params.ptr; a.target = tn.target.ptr
    a.packed = net.packed === nothing ? C_NULL : net.packed.ptr
    a.target_packed = tn.target_packed === nothing ? C_NULL : tn.target_packed.ptr
    a.m = net.m.ptr; a.v = net.v.ptr; a.beta_pow = net.beta_pow.ptr
    a.lr = net.lr; a.beta1 = net.beta1; a.beta2 = net.beta2; a.adam_eps = net.eps
    a.max_grad_norm = L.max_grad_norm; a.grad_scale = 1f0
    a.explorer_seed = p.explorer_seed; a.batch = L.batchsize; a.gamma = L.γ; a.huber_delta = L.δ
    a.sampler_seed = L.seed; a.rho = tn.ρ
    a.workspace = L.workspace.ptr; a.grad = L.grad.ptr; a.loss = L.loss.ptr; a.gn = net.gn.ptr
    a.actions = p.actions.ptr; a.q = p.q.ptr
    GC.@preserve a t env begin
        while true
            a.eps = get_ϵ(p.explorer); a.explorer_step = UInt32(p.explorer.step); p.explorer.step += 1
            on_insert!(t.controller, 1)
            L.vec_steps += 1
            frames = min(length(t) + 1, capacity(t))
            a.do_update = (frames * env.n >= L.min_replay_history && L.vec_steps % L.update_freq == 0 &&
                           on_sample!(t.controller)) ? 1 : 0
            a.draw_ctr = L.draw_ctr
            a.do_sync = (a.do_update == 1 && (tn.n_optimise + 1) % tn.sync_freq == 0) ? 1 : 0
            chk(ccall((:rlhip_dqn_vec_step_f32, LIB), Int32, (Ref{DqnStepArgs}, Ptr{Cvoid}), a, stream()))
            if a.do_update == 1
                L.draw_ctr += 1; L.n_updates += 1
                tn.n_optimise = a.do_sync == 1 ? 0 : tn.n_optimise + 1
            end
            push!(hook, PostActStage(), agent, env)
            check!(stop_condition, agent, env) && break
        end
    end
    env.obs_valid = true
    push!(agent, PostExperimentStage(), env)
    push!(hook, PostExperimentStage(), agent, env)
    hook
end

"""
The PPO loop.  Default (`policy.fused == false`): the per-step protocol through the generic stages -- exactly `run.jl:52-67`:
one `check!(stop_condition, …)` and one `PostActStage` hook push per vec-step, so `StopAfterNSteps(n)` runs n vec-steps
(`stop_conditions.jl:65-69`) and per-step hooks see every step.

`HipPPOPolicy(env; fused = true)` (or the keyword here): ONE launch per update period of T = `update_freq` vec-steps
(`rlhip_ppo_rollout_f32`), then the update.  The loop still speaks in vec-steps: after each period the stop condition is
checked T times and the hook gets T `PostActStage` pushes -- `HipEpisodeStats` reads step t's rewards / terminal flags from row t
of the policy's trajectory (so its episode log is identical to the per-step run's); any other hook is pushed with the env as it
stands after the period's last step.  The run ends at the first period boundary at or after the step the stop condition fired
on: `StopAfterNSteps(n)` runs `ceil(n / T) * T` vec-steps (n itself when T divides n).
"""
function _run(p::HipPPOPolicy, env::HipVecEnv, stop_condition, hook, reset_condition; fused = p.fused)
    push!(hook, PreExperimentStage(), p, env)
    while true
        if fused
            rollout!(p, env)
            optimise!(p, PostActStage(), env; fused_rollout = true)
            stop = false
            for t in 1:p.T
                push_period_step!(hook, p, env, t)
                stop |= check!(stop_condition, p, env)      # run.jl:64, once per vec-step of the period
            end
            stop && break
        else
            action = plan!(p, env)
            push!(p, PreActStage(), env)
            act!(env, action)
            push!(p, PostActStage(), env, action)
            optimise!(p, PostActStage(), env)
            push!(hook, PostActStage(), p, env)
            check!(stop_condition, p, env) && break
        end
    end
    push!(hook, PostExperimentStage(), p, env)
    hook
end
"the `PostActStage` hook push of step t (1-based) of a fused update period"
push_period_step!(hook, p::HipPPOPolicy, env::HipVecEnv, t::Int) = push!(hook, PostActStage(), p, env)
function push_period_step!(h::HipEpisodeStats, p::HipPPOPolicy, env::HipVecEnv, t::Int)
    # rewards / terminal flags of step t: row t of the time-major (T, n) traces the rollout launch wrote
    chk(ccall((:rlhip_hook_episode_stats, LIB), Int32,
              (Ptr{Cvoid}, Ptr{Cvoid}, Int64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Ptr{Cvoid}, Ptr{Cvoid}),
              offset(p.rew, (t - 1) * p.n), offset(p.terminal, (t - 1) * p.n), h.n, h.vec_step, h.steps_acc.ptr, h.ret_acc.ptr,
              h.log.ptr, h.cap, h.count.ptr, stream()))
    h.vec_step += 1
    nothing
end

end # module
