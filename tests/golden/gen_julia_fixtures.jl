# gen_julia_fixtures.jl -- turns the env-physics rows E1-E3 of SURVEY.md section 8 from "parity unpinned" into "pinned".
#
# The reference's own tests of CartPoleEnv / PendulumEnv / MountainCarEnv are interface-level (RLBase.test_interfaces!), and
# there is no `julia` binary in the build image, so the oracle (oracle/rlo_envs_impl.h) restates the physics from the source
# text alone.  This script runs the REAL reference environments and writes teacher-forced one-step vectors:
#
#     julia --project=<a project with ReinforcementLearningEnvironments> tests/golden/gen_julia_fixtures.jl
#         -> tests/golden/julia_env_steps.json
#
# tests/test_oracle_julia_fixtures.py then loads the file (it is skipped while the file is absent), feeds every recorded
# (state, t, action) to the oracle's step and demands the recorded next state / reward / terminal flag -- Float64 cases bit
# for bit, Float32 cases within 1 ulp (sin / cos of the two libms).  Anyone with Julia can generate the fixture; nothing in
# the test-suite needs Julia afterwards.  UNTESTED HERE (no julia in the image): written against
#   src/ReinforcementLearningEnvironments/src/environments/examples/CartPoleEnv.jl:48-140
#   .../PendulumEnv.jl:13-122      .../MountainCarEnv.jl:42-135
#
# Per case the script sets the env's fields directly (teacher forcing: no dependence on the RNG stream), calls
# act!(env, action) once and records state(env), reward(env), is_terminated(env) and the step counter.  reset!(env) is
# covered separately: it is run with a scripted RNG whose uniforms are recorded next to the resulting state, so that the
# loader can check `state = 0.1 u - 0.05` (CartPole), `theta = 2 pi (u - 1)` ... in the recorded element type.
using ReinforcementLearningBase, ReinforcementLearningEnvironments
using Random

# ---------------------------------------------------------------- a scripted RNG: uniforms come from a list, integers from Xoshiro
mutable struct ScriptedRNG <: AbstractRNG
    u::Vector{Float64}
    i::Int
    ints::Xoshiro
    drawn::Vector{Float64}
end
ScriptedRNG(u) = ScriptedRNG(u, 0, Xoshiro(1), Float64[])
function next_u!(r::ScriptedRNG)
    r.i += 1
    v = r.u[mod1(r.i, length(r.u))]
    push!(r.drawn, v)
    v
end
Random.rand(r::ScriptedRNG, ::Random.SamplerTrivial{Random.CloseOpen01{Float64}}) = next_u!(r)
Random.rand(r::ScriptedRNG, ::Random.SamplerTrivial{Random.CloseOpen01{Float32}}) = Float32(next_u!(r))
Random.rand(r::ScriptedRNG, ::Random.SamplerTrivial{Random.CloseOpen01{Float16}}) = Float16(next_u!(r))
Random.rand(r::ScriptedRNG, s::Random.SamplerType{T}) where {T<:Union{UInt8,UInt16,UInt32,UInt64,UInt128,Int32,Int64}} = rand(r.ints, s)
Random.rng_native_52(::ScriptedRNG) = UInt64

# ---------------------------------------------------------------- tiny JSON writer (no JSON.jl dependency)
jnum(x::Bool) = x ? "true" : "false"
jnum(x::Integer) = string(x)
jnum(x::AbstractFloat) = isfinite(x) ? string(Float64(x)) : "null"   # exact: every Float32 is a Float64; the loader casts back
jarr(v) = "[" * join((jnum(x) for x in v), ", ") * "]"
# bit patterns as well: the loader compares integers, no decimal parsing in the loop
bits(x::Float32) = reinterpret(UInt32, x)
bits(x::Float64) = reinterpret(UInt64, x)
jbits(v) = "[" * join((string(bits(x)) for x in v), ", ") * "]"

cases = String[]
function record!(env_name, T, cfg, state_in, t_in, action, env)
    s_out = collect(T, env.state)
    push!(cases, """{"env": "$env_name", "T": "$T", "cfg": $cfg, "state": $(jarr(state_in)), "state_bits": $(jbits(state_in)), """ *
                 """"t": $t_in, "action": $(action isa AbstractFloat ? jnum(action) : string(action)), """ *
                 """"action_bits": $(action isa AbstractFloat ? string(bits(T(action))) : string(action)), """ *
                 """"next_state": $(jarr(s_out)), "next_state_bits": $(jbits(s_out)), "reward": $(jnum(T(reward(env)))), """ *
                 """"reward_bits": $(string(bits(T(reward(env))))), "terminated": $(jnum(is_terminated(env))), "t_out": $(env.t), """ *
                 """"obs": $(jarr(collect(T, state(env)))), "obs_bits": $(jbits(collect(T, state(env))))}""")
end

rng = Xoshiro(20260925)
for T in (Float32, Float64)
    # ---- CartPole: default config and the reference test-suite's own `thetathreshold = 90` configuration
    for (cfgname, kw) in (("{}", (;)), ("{\"thetathreshold\": 90}", (; thetathreshold = 90)))
        env = CartPoleEnv(; T = T, rng = Xoshiro(1), kw...)
        for k in 1:400
            reset!(env)
            th_max = cfgname == "{}" ? 0.25 : 3.0
            s = T[4.8 * rand(rng) - 2.4, 4 * rand(rng) - 2, 2 * th_max * rand(rng) - th_max, 6 * rand(rng) - 3]
            if k <= 40      # edges: exactly at / next to the thresholds (strict `>`), last allowed step
                s[1] = T((k % 2 == 0 ? 1 : -1) * 2.4) + (k % 3 == 0 ? eps(T(2.4)) : T(0))
            end
            t_in = k <= 80 ? 199 + (k % 3) : rand(rng, 0:150)
            env.state .= s
            env.t = t_in
            env.done = false
            a = rand(rng, 1:2)
            act!(env, a)
            record!("cartpole", T, cfgname, s, t_in, a, env)
        end
    end
    # ---- Pendulum: continuous torque and the discrete variant
    for continuous in (true, false)
        env = PendulumEnv(; T = T, continuous = continuous, rng = Xoshiro(1))
        cfgname = continuous ? "{\"continuous\": true}" : "{\"continuous\": false}"
        for k in 1:400
            reset!(env)
            s = T[(k <= 60 ? 40 : 2pi) * (2 * rand(rng) - 1), 16 * rand(rng) - 8]      # large angles exercise angle_normalize
            t_in = k <= 80 ? 198 + (k % 3) : rand(rng, 0:150)
            env.state .= s
            env.t = t_in
            env.done = false
            a = continuous ? T(6 * rand(rng) - 3) : rand(rng, 1:3)                     # |torque| > 2 exercises the clamp
            act!(env, a)
            record!("pendulum", T, cfgname, s, t_in, a, env)
        end
    end
    # ---- MountainCar: discrete and continuous
    for continuous in (false, true)
        env = MountainCarEnv(; T = T, continuous = continuous, rng = Xoshiro(1))
        cfgname = continuous ? "{\"continuous\": true}" : "{\"continuous\": false}"
        for k in 1:400
            reset!(env)
            s = T[1.8 * rand(rng) - 1.2, 0.14 * rand(rng) - 0.07]
            if k <= 40
                s[1] = k % 2 == 0 ? T(-1.2) : T(0.5)                                   # the wall (v := 0) and the goal edge
            end
            t_in = k <= 80 ? 198 + (k % 3) : rand(rng, 0:150)
            env.state .= s
            env.t = t_in
            env.done = false
            a = continuous ? T(2.4 * rand(rng) - 1.2) : rand(rng, 1:3)
            act!(env, a)
            record!("mountaincar", T, cfgname, s, t_in, a, env)
        end
    end
end

# ---------------------------------------------------------------- reset!(env) with scripted uniforms
resets = String[]
for T in (Float32, Float64), k in 1:50
    u = [rand(rng) for _ in 1:8]
    for (name, mk) in (("cartpole", r -> CartPoleEnv(; T = T, rng = r)), ("pendulum", r -> PendulumEnv(; T = T, rng = r)),
                       ("mountaincar", r -> MountainCarEnv(; T = T, rng = r)))
        r = ScriptedRNG(T == Float32 ? Float64.(Float32.(u)) : u)
        env = mk(r)          # the constructors call reset! themselves
        empty!(r.drawn)
        r.i = 0
        reset!(env)
        push!(resets, """{"env": "$name", "T": "$T", "uniforms": $(jarr(r.drawn)), "state": $(jarr(collect(T, env.state))), """ *
                      """"state_bits": $(jbits(collect(T, env.state))), "t": $(env.t)}""")
    end
end

open(joinpath(@__DIR__, "julia_env_steps.json"), "w") do io
    println(io, "{\"generator\": \"tests/golden/gen_julia_fixtures.jl\", \"julia\": \"$(VERSION)\",")
    println(io, " \"steps\": [")
    println(io, join(cases, ",\n"))
    println(io, " ],\n \"resets\": [")
    println(io, join(resets, ",\n"))
    println(io, " ]}")
end
println("wrote $(length(cases)) step cases and $(length(resets)) reset cases")
