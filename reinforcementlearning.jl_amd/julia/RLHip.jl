# RLHip.jl -- Julia glue for librlhip.so (the reference-side binding a maintainer would add).
#
# NOT EXECUTED in this repository: the build image has no `julia` binary (SURVEY.md section 0), so this
# file documents the binding; the tested host mirror is reinforcementlearning.jl_amd/rlhip/ (Python, same
# structure, same C ABI calls).  Every method is a thin `ccall` into include/rlhip.h.  The plugin surface
# (AbstractEnv / AbstractPolicy / Trajectory, `run(policy, env, stop, hook)`) is unchanged: new types
# subtype the reference's abstract types and one `_run` method is added for the vector env, exactly as the
# historical MultiThreadEnv did (docs/homepage/blog/an_introduction_..._thoughts/index.md:351-374).
module RLHip

using ReinforcementLearningBase, ReinforcementLearningCore
import ReinforcementLearningBase: state, reward, is_terminated, action_space, state_space, act!, reset!, plan!, optimise!
import ReinforcementLearningCore: _run, check!, PreActStage, PostActStage, PreExperimentStage, PostExperimentStage
using Random, DomainSets

const LIB = get(ENV, "RLHIP_LIB", "librlhip.so")

struct RLHipError <: Exception
    code::Int32
    msg::String
end
function chk(rc::Int32)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:rlhip_last_error, LIB), Cstring, ()))
    rc == -1 ? throw(ArgumentError(msg)) : throw(RLHipError(rc, msg))
end

# ---- device buffers owned through the ABI (no AMDGPU.jl needed) -------------------------------------
mutable struct DevBuf{T}
    ptr::Ptr{Cvoid}
    n::Int
    function DevBuf{T}(n::Integer) where {T}
        p = Ref{Ptr{Cvoid}}()
        chk(ccall((:rlhip_malloc, LIB), Int32, (Ref{Ptr{Cvoid}}, Csize_t), p, n * sizeof(T)))
        chk(ccall((:rlhip_memset, LIB), Int32, (Ptr{Cvoid}, Int32, Csize_t, Ptr{Cvoid}), p[], 0, n * sizeof(T), C_NULL))
        b = new{T}(p[], n)
        finalizer(x -> ccall((:rlhip_free, LIB), Int32, (Ptr{Cvoid},), x.ptr), b)
    end
end
to_host(b::DevBuf{T}) where {T} = (h = Vector{T}(undef, b.n);
    chk(ccall((:rlhip_memcpy_d2h, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), h, b.ptr, sizeof(h), C_NULL)); h)
to_dev!(b::DevBuf{T}, h::Vector{T}) where {T} =
    chk(ccall((:rlhip_memcpy_h2d, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), b.ptr, h, sizeof(h), C_NULL))

# ---- POD structs of include/rlhip.h ----------------------------------------------------------------
struct CartPoleCfg  # rlhip_cartpole_cfg <- CartPoleEnv(; kwargs...) CartPoleEnv.jl:22-32
    gravity::Float64; masscart::Float64; masspole::Float64; halflength::Float64; forcemag::Float64
    dt::Float64; thetathreshold_deg::Float64; xthreshold::Float64; max_steps::Int64; continuous::Int32
end
struct AcrobotCfg  # rlhip_acrobot_cfg <- AcrobotEnv(; kwargs...) 3rd_party/AcrobotEnv.jl:22-40 (kind = 3)
    link_length_a::Float64; link_length_b::Float64; link_mass_a::Float64; link_mass_b::Float64
    link_com_pos_a::Float64; link_com_pos_b::Float64; link_moi::Float64; max_torque_noise::Float64
    max_vel_a::Float64; max_vel_b::Float64; g::Float64; dt::Float64; max_steps::Int64; nips::Int32
end
struct EnvState  # rlhip_env_state
    s::NTuple{4,Ptr{Cvoid}}; t::Ptr{Cvoid}; done::Ptr{Cvoid}; reward::Ptr{Cvoid}; episode::Ptr{Cvoid}
end

# ---- HipVecEnv <: AbstractEnv ------------------------------------------------------------------------
mutable struct HipVecEnv{K,T} <: AbstractEnv     # K in (:cartpole, :pendulum, :mountaincar, :acrobot)
    kind::Int32
    cfg::Ref                                       # CartPoleCfg / PendulumCfg / MountainCarCfg / AcrobotCfg
    n::Int
    seed::UInt64
    env_id_base::UInt32
    s::Vector{DevBuf{T}}; t::DevBuf{Int32}; done::DevBuf{UInt8}; rew::DevBuf{T}; episode::DevBuf{UInt32}
    obs::DevBuf{T}
    st::Ref{EnvState}
end

function HipCartPoleEnv(n::Integer; T = Float32, seed = 0, env_id_base = 0, kwargs...)
    cfg = Ref{CartPoleCfg}()
    chk(ccall((:rlhip_cartpole_default, LIB), Int32, (Ref{CartPoleCfg},), cfg))
    # kwargs (gravity = ..., max_steps = ...) overwrite fields of cfg[] here
    s = [DevBuf{T}(n) for _ in 1:4]
    env = HipVecEnv{:cartpole,T}(0, cfg, n, seed, env_id_base, s, DevBuf{Int32}(n), DevBuf{UInt8}(n), DevBuf{T}(n),
                                 DevBuf{UInt32}(n), DevBuf{T}(4n), Ref{EnvState}())
    env.st[] = EnvState((s[1].ptr, s[2].ptr, s[3].ptr, s[4].ptr), env.t.ptr, env.done.ptr, env.rew.ptr, env.episode.ptr)
    reset!(env)                                    # the constructor resets once, CartPoleEnv.jl:77
    env
end

# reset!(env): all instances (is_force) or only the terminated ones (MultiThreadEnv semantics)
function reset!(env::HipVecEnv{K,T}; is_force = true) where {K,T}
    mask = is_force ? C_NULL : env.done.ptr
    chk(ccall((:rlhip_env_reset, LIB), Int32,
              (Int32, Int32, Ptr{Cvoid}, Ref{EnvState}, Int64, UInt64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}),
              env.kind, T === Float64, env.cfg, env.st, env.n, env.seed, env.env_id_base, mask, C_NULL))
end

# act!(env, actions::DevBuf): Julia actions are 1-based; the ABI is 0-based -> the policy kernels already
# produce 0-based device actions, host-provided vectors are shifted here.
function act!(env::HipVecEnv{K,T}, actions::DevBuf) where {K,T}
    chk(ccall((:rlhip_env_step, LIB), Int32,
              (Int32, Int32, Ptr{Cvoid}, Ref{EnvState}, Int64, Ptr{Cvoid}, Int32, UInt64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
              env.kind, T === Float64, env.cfg, env.st, env.n, actions.ptr, 1, env.seed, env.env_id_base,
              C_NULL, env.obs.ptr, C_NULL))
end
act!(env::HipVecEnv, actions::AbstractVector{<:Integer}) = (@assert all(a -> a in action_space(env), actions);
    d = DevBuf{Int32}(length(actions)); to_dev!(d, Int32.(actions .- 1)); act!(env, d))

state(env::HipVecEnv{K,T}, ::Observation, ::DefaultPlayer) where {K,T} =
    permutedims(reshape(to_host(env.obs), env.n, :))          # (ns, N) like the reference's batched state
reward(env::HipVecEnv) = to_host(env.rew)
is_terminated(env::HipVecEnv) = Bool.(to_host(env.done))       # iterable, as BatchStepsPerEpisode expects (hooks.jl:219-231)
action_space(env::HipVecEnv{:cartpole}) = Base.OneTo(2)
Random.seed!(env::HipVecEnv, seed) = (env.seed = seed)

# ---- scans: generalized_advantage_estimation on device matrices ---------------------------------------
function gae!(adv::DevBuf{Float32}, r::DevBuf{Float32}, v::DevBuf{Float32}, n1, n2, γ::Float32, λ::Float32;
              terminal::Union{Nothing,DevBuf{UInt8}} = nothing, dims = 2)
    chk(ccall((:rlhip_gae_f32, LIB), Int32,
              (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Float32, Float32, Ptr{Cvoid}, Int32, Ptr{Cvoid}),
              adv.ptr, r.ptr, v.ptr, n1, n2, γ, λ, terminal === nothing ? C_NULL : terminal.ptr, dims, C_NULL))
end

# ---- PPO: one ccall per update period -------------------------------------------------------------------
# rlhip_ppo_rollout_f32 / rlhip_ppo_gae_f32 / rlhip_ppo_update_f32 take the POD structs rlhip_ppo_cfg and
# rlhip_ppo_traj (device pointers of the PPOTrajectory traces); see INTEGRATION.md for the full stubs.

# ---- prioritized replay: CircularPrioritizedTraces + prioritized BatchSampler (RLTrajectories 0.4) -------
# `ring` is the POD rlhip_ring mirror (Ref{Ring}); `tree` a zero-initialised DevBuf{Float32}(rlhip_sumtree_nodes(n)).
sumtree_nodes(n_leaves) = ccall((:rlhip_sumtree_nodes, LIB), Int64, (Int64,), n_leaves)
push_priority!(ring, tree::DevBuf{Float32}, p::Float32) = chk(ccall((:rlhip_ring_push_priority, LIB), Int32,
    (Ptr{Cvoid}, Ptr{Cvoid}, Float32, Ptr{Cvoid}), ring, tree.ptr, p, C_NULL))
sample_prioritized!(idx::DevBuf{Int64}, key::DevBuf{Int64}, prio::DevBuf{Float32}, ring, tree, batch, seed, ctr) =
    chk(ccall((:rlhip_ring_sample_prioritized, LIB), Int32,
              (Ptr{Cvoid}, Ptr{Cvoid}, Int64, UInt64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
              ring, tree.ptr, batch, seed, ctr, idx.ptr, key.ptr, prio.ptr, C_NULL))
# trajectory[:priority, keys] = p      (keys are the 0-based physical leaf keys returned by the sampler)
set_priority!(tree::DevBuf{Float32}, n_leaves, key::DevBuf{Int64}, p::DevBuf{Float32}, n) =
    chk(ccall((:rlhip_sumtree_update, LIB), Int32, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
              tree.ptr, n_leaves, key.ptr, p.ptr, n, C_NULL))

# ---- the blog's 3-layer Q-network on the MFMA (Chain(Dense(ns,128,relu), Dense(128,128,relu), Dense(128,na))) ---
# forward(learner, x) / plan!(QBasedPolicy, env): params = Flux.destructure(model)[1] on the device, `packed` =
# DevBuf{UInt16}(rlhip_mlp3_packed_elems(128)) refreshed by rlhip_mlp3_pack_bf16 after every optimise!.
mlp3_pack!(packed::DevBuf{UInt16}, params::DevBuf{Float32}, ns, na) = chk(ccall((:rlhip_mlp3_pack_bf16, LIB), Int32,
    (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}), params.ptr, ns, 128, na, packed.ptr, C_NULL))
dqn3_plan!(actions::DevBuf{Int32}, q, params, packed, ns, na, act, obs, n, ϵ, seed, env_id_base, step) =
    chk(ccall((:rlhip_dqn3_plan_f32, LIB), Int32,
              (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Int32, Ptr{Cvoid}, Int64, Float64, UInt64, UInt32, UInt32,
               Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
              params.ptr, packed.ptr, ns, 128, na, act, obs.ptr, n, ϵ, seed, env_id_base, step, actions.ptr, q.ptr, C_NULL))
# optimise!(learner, batch) up to the gradient: rlhip_dqn3_grad_f32(ring, 128, na, act, params, packed, target,
# target_packed, batch, idx_or_NULL, γ, δ, seed, draw_ctr, workspace, grad, loss, td_or_NULL, stream); then
# rlhip_clip_adam_f32 + rlhip_mlp3_pack_bf16 (+ rlhip_polyak_f32 and a re-pack of the target every sync_freq).

# ---- the vector-env run loop (one method added; RLCore/src/core/run.jl is untouched) -------------------
function _run(policy::AbstractPolicy, env::HipVecEnv, stop_condition, hook, reset_condition)
    push!(hook, PreExperimentStage(), policy, env)
    push!(policy, PreExperimentStage(), env)
    while true
        action = plan!(policy, env)
        push!(policy, PreActStage(), env)
        push!(hook, PreActStage(), policy, env)
        act!(env, action)
        push!(policy, PostActStage(), env, action)
        optimise!(policy, PostActStage())
        push!(hook, PostActStage(), policy, env)
        check!(stop_condition, policy, env) && break
    end
    push!(policy, PostExperimentStage(), env)
    push!(hook, PostExperimentStage(), policy, env)
    hook
end

end # module
