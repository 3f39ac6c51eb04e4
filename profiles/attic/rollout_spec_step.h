// profiles/attic/rollout_spec_step.h -- NOT part of the library.  The two roles of rollout_split_kernel (csrc/ppo.hip) with the env
// step of categorical heads SPECULATED by the critic wave (one candidate action per lane of the env's group; the actor wave
// picks the candidate its selection names), as built and measured in round 4: bit-identical trajectories (87 GPU tests of the
// learner / run / abi-host suites passed), and SLOWER -- 54.5 us against 49.0 us per 32-step rollout of 4096 CartPole envs on the
// same box, 0.4042 / 0.4026 against 0.3968 / 0.4005 ms per PPO iteration (profiles/r04_rollout.md section 2b).  The body below sat
// between the LDS declarations and the `role == 0` branch of the kernel, under a `bool SPEC` template parameter.
#if 0
    // SPEC (categorical heads: the action is one of <= 3 values): the critic wave steps the env for EVERY action ahead of the
    // selection, one candidate per lane of the env's group -- the same env_step1 / env_reset1 / env_obs1 on the same operands,
    // so the chosen candidate IS the sequential result -- and the actor wave picks: the env step leaves the dependent chain
    // (obs -> actor -> select), the two waves carry ~285 / ~265 instructions per vec-step instead of ~380 / ~155.
    constexpr int NCAND = SPEC ? 3 : 1;
    __shared__ float4 l_cand[2][EPB][NCAND][3];  // per action: {s0..s3}, {t, episode, reward, terminal}, {x0..x3} after the step
    __shared__ float2 l_msg[2][EPB];             // the actor wave's answer for step t: {logp, action bits}
    static_assert(!SPEC || HEAD == 2 || HEAD == 3, "speculation needs a small action set known at compile time");
    static_assert(!SPEC || L >= 4, "one candidate per lane of the env's group");

    if (SPEC && role == 0) {
        __builtin_amdgcn_s_setprio(3);
        NetRegs<NS, HPL> A;
        load_net<NS, HPL>(A, params, H, pd.nout_a, sub, L);
        LaneState<float> e;
#pragma unroll
        for (int k = 0; k < 4; ++k) e.s[k] = 0.0f;
#pragma unroll
        for (int k = 0; k < P::SDIM; ++k) e.s[k] = st.s[k][env];
        e.t = st.t[env];
        e.episode = st.episode[env];
        float last_r = 0.0f;
        bool last_d = false;
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        env_obs1(p, e, x);
        // the noise of chunk 0 (the critic wave fills chunk 1 meanwhile): one step per lane of the env's group
        for (int i = sub; i < NOISE_CH && i < T; i += L) {
            double nz[MAXO] = {0.0, 0.0, 0.0, 0.0};
            policy_noise(cont, na, seed, id, vec_step0 + (uint32_t)i, nz);
#pragma unroll
            for (int k = 0; k < MAXO; ++k) l_noise[0][eg][i][k] = nz[k];
        }
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            double nz[MAXO];
#pragma unroll
            for (int k = 0; k < MAXO; ++k) nz[k] = (k < na) ? l_noise[(t / NOISE_CH) & 1][eg][t & (NOISE_CH - 1)][k] : 0.0;
            float oa[MAXO];
            net_forward<NS, HPL, L, ACT, NOA>(A, x, oa);
            int32_t ai;
            float af, lp;
            policy_select(cont, na, oa, nz, ai, af, lp);
            if (sub == 0) l_msg[t & 1][eg] = make_float2(lp, __int_as_float(ai));
            __syncthreads();  // the candidates of step t are in LDS, the critic wave reads the answer
            const float4 c0 = l_cand[t & 1][eg][ai][0], c1 = l_cand[t & 1][eg][ai][1], c2 = l_cand[t & 1][eg][ai][2];
            e.s[0] = c0.x, e.s[1] = c0.y, e.s[2] = c0.z, e.s[3] = c0.w;
            e.t = __float_as_int(c1.x);
            e.episode = (uint32_t)__float_as_int(c1.y);
            last_r = c1.z;
            last_d = c1.w != 0.0f;
            x[0] = c2.x, x[1] = c2.y, x[2] = c2.z, x[3] = c2.w;
        }
        if (writer && store_state) {
#pragma unroll
            for (int k = 0; k < P::SDIM; ++k) st.s[k][env] = e.s[k];
            st.t[env] = e.t;
            st.episode[env] = e.episode;
            if (T > 0) {
                st.reward[env] = last_r;
                st.done[env] = (uint8_t)last_d;
            }
        }
    } else if (SPEC) {
        NetRegs<NS, HPL> C;
        load_net<NS, HPL>(C, params + pd.np_a, H, 1, sub, L);
        LaneState<float> e;
#pragma unroll
        for (int k = 0; k < 4; ++k) e.s[k] = 0.0f;
#pragma unroll
        for (int k = 0; k < P::SDIM; ++k) e.s[k] = st.s[k][env];
        e.t = st.t[env];
        e.episode = st.episode[env];
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        env_obs1(p, e, x);
        auto fill_noise = [&](int c0) {
            for (int i = sub; i < NOISE_CH && c0 + i < T; i += L) {
                double nz[MAXO] = {0.0, 0.0, 0.0, 0.0};
                policy_noise(cont, na, seed, id, vec_step0 + (uint32_t)(c0 + i), nz);
#pragma unroll
                for (int k = 0; k < MAXO; ++k) l_noise[(c0 / NOISE_CH) & 1][eg][i][k] = nz[k];
            }
        };
        if (NOISE_CH < T) fill_noise(NOISE_CH);  // chunk 1; chunk 0 comes from the actor wave
        __syncthreads();
        const int cand = sub % HEAD;  // this lane's candidate action (lanes >= HEAD recompute one of them and do not store)
        for (int t = 0; t <= T; ++t) {
            if (t > 0) {  // what the actor wave chose in step t - 1, and the candidate it thereby made real
                const float2 msg = l_msg[(t - 1) & 1][eg];
                const int a_prev = __float_as_int(msg.y);
                const float4 c0 = l_cand[(t - 1) & 1][eg][a_prev][0], c1 = l_cand[(t - 1) & 1][eg][a_prev][1],
                             c2 = l_cand[(t - 1) & 1][eg][a_prev][2];
                e.s[0] = c0.x, e.s[1] = c0.y, e.s[2] = c0.z, e.s[3] = c0.w;
                e.t = __float_as_int(c1.x);
                e.episode = (uint32_t)__float_as_int(c1.y);
                x[0] = c2.x, x[1] = c2.y, x[2] = c2.z, x[3] = c2.w;
                if (writer) {
                    tr.logp[(int64_t)(t - 1) * n + env] = msg.x;
                    tr.action_i[(int64_t)(t - 1) * n + env] = a_prev;
                    tr.reward[(int64_t)(t - 1) * n + env] = c1.z;
                    tr.terminal[(int64_t)(t - 1) * n + env] = (uint8_t)(c1.w != 0.0f);
                }
            }
            if (t < T) {
                // a later noise chunk (rollouts longer than 32 steps): into the buffer the actor wave left in step t - 1
                if (t > 0 && (t & (NOISE_CH - 1)) == 0 && t + NOISE_CH < T) fill_noise(t + NOISE_CH);
                LaneState<float> ec = e;
                float rc;
                bool dc;
                env_step1(p, ec, cand, 0.0f, rc, dc);
                if (dc) env_reset1(p, ec, seed, id);  // MultiThreadEnv auto-reset
                float xc[4] = {0.f, 0.f, 0.f, 0.f};
                env_obs1(p, ec, xc);
                if (sub < HEAD) {
                    l_cand[t & 1][eg][sub][0] = make_float4(ec.s[0], ec.s[1], ec.s[2], ec.s[3]);
                    l_cand[t & 1][eg][sub][1] = make_float4(__int_as_float(ec.t), __int_as_float((int)ec.episode), rc, dc ? 1.0f : 0.0f);
                    l_cand[t & 1][eg][sub][2] = make_float4(xc[0], xc[1], xc[2], xc[3]);
                }
            }
            float oc[MAXO];
            net_forward<NS, HPL, L, ACT, 1>(C, x, oc);  // V(s_t); t = T: the bootstrap value
            if (writer) {
#pragma unroll
                for (int k = 0; k < NS; ++k) tr.obs[((int64_t)t * NS + k) * n + env] = x[k];
                tr.value[(int64_t)t * n + env] = oc[0];
            }
            if (t < T) __syncthreads();
        }
        if (writer && T > 0 && tr.adv && tr.ret)
            gae_scan_lane(tr.adv, tr.ret, tr.reward, tr.value, tr.terminal, n, T, env, pd.gamma, pd.lambda);
    }
#endif
