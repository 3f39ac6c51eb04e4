"""ctypes/numpy binding of oracle/_build/librl_oracle.so (TEST INFRASTRUCTURE ONLY).

`build()` compiles the C restatement with gcc (make -C oracle); `lib()` loads it.  All array
arguments are numpy arrays; matrices are passed column-major (Fortran order) like Julia.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RLO_ORACLE_SO: an alternative build of the same sources (the sanitizer build of `make asan`, tests/test_asan.py)
_SO = os.environ.get("RLO_ORACLE_SO") or os.path.join(_HERE, "_build", "librl_oracle.so")
_SO_OMP = os.path.join(_HERE, "_build", "librl_oracle_omp.so")
_lib = None
_lib_serial = None
_lib_omp = None

KIND = {"cartpole": 0, "pendulum": 1, "mountaincar": 2, "acrobot": 3}
TAG = dict(RESET=0, EXPLORE=1, GUMBEL=2, NORMAL=3, SAMPLER=4, SHUFFLE=5, INIT=6, SYNTH=7, ENVNOISE=8)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if (not force and os.path.exists(_SO) and os.path.exists(_SO_OMP)
            and all(min(os.path.getmtime(_SO), os.path.getmtime(_SO_OMP)) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return _SO


class CartPoleCfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("gravity", "masscart", "masspole", "halflength", "forcemag",
                                          "dt", "thetathreshold_deg", "xthreshold")] + \
               [("max_steps", C.c_int64), ("continuous", C.c_int32)]


class PendulumCfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("max_speed", "max_torque", "g", "m", "l", "dt")] + \
               [("max_steps", C.c_int64), ("continuous", C.c_int32), ("n_actions", C.c_int32)]


class MountainCarCfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("min_pos", "max_pos", "max_speed", "goal_pos",
                                          "goal_velocity", "power", "gravity")] + \
               [("max_steps", C.c_int64), ("continuous", C.c_int32)]


class AcrobotCfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("link_length_a", "link_length_b", "link_mass_a", "link_mass_b",
                                          "link_com_pos_a", "link_com_pos_b", "link_moi", "max_torque_noise",
                                          "max_vel_a", "max_vel_b", "g", "dt")] + \
               [("max_steps", C.c_int64), ("nips", C.c_int32)]


class EnvStateC(C.Structure):
    _fields_ = [("s", C.c_void_p * 4), ("t", C.c_void_p), ("done", C.c_void_p),
                ("reward", C.c_void_p), ("episode", C.c_void_p)]


class PPOCfg(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("gamma", "lam", "clip_range", "max_grad_norm",
                                         "actor_loss_weight", "critic_loss_weight",
                                         "entropy_loss_weight", "lr", "beta1", "beta2", "adam_eps")] + \
               [(n, C.c_int32) for n in ("n_epochs", "n_microbatches", "hidden", "act", "continuous",
                                         "normalize_advantage", "layers")]


class PPOTrajC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("obs", "logp", "value", "reward", "adv", "ret", "action_f",
                                          "action_i", "terminal")]


class RingC(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("capacity", "n_env", "obs_dim", "head_sa", "len_sa",
                                         "head_rt", "len_rt")] + \
               [(n, C.c_void_p) for n in ("state", "action", "reward", "terminal")]


def usable_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a container with
    `cpu.max = 1600000 100000` on a 256-thread host gets 16 CPUs' worth of time: more threads only get throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def use_all_cores(enable=True, threads=None):
    """Switch every binding call to the -fopenmp build of the same sources (bench.py's all-cores cpu_baseline)
    or back to the sequential parity oracle.  Returns the number of threads the selected build uses."""
    global _lib, _lib_serial, _lib_omp
    lib()
    if _lib_serial is None:
        _lib_serial = _lib
    if enable:
        if _lib_omp is None:
            _lib_omp = _load(_SO_OMP)
        _lib = _lib_omp
        gomp = C.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(int(threads or usable_cpus()))
        return int(gomp.omp_get_max_threads())
    _lib = _lib_serial
    return 1


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _load(_SO)
    return _lib


def _load(path):
    if True:
        _lib = C.CDLL(path)
        _lib.rlo_u01_f32.restype = C.c_float
        _lib.rlo_u01_f64.restype = C.c_double
        _lib.rlo_randint.restype = C.c_uint32
        _lib.rlo_permute.restype = C.c_uint32
        _lib.rlo_permute.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        _lib.rlo_get_eps.restype = C.c_double
        _lib.rlo_get_eps.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int64, C.c_int64, C.c_int64]
        _lib.rlo_find_all_max_f64.restype = C.c_int64
        _lib.rlo_findmax_f64.restype = C.c_int64
        _lib.rlo_findmax_f32.restype = C.c_int64
        _lib.rlo_clip_by_global_norm_f32.restype = C.c_float
        _lib.rlo_normlogpdf_f32.restype = C.c_float
        _lib.rlo_normlogpdf_f32.argtypes = [C.c_float] * 3
        _lib.rlo_huber_f32.restype = C.c_float
        _lib.rlo_mlp2_nparams.restype = C.c_int64
        _lib.rlo_mlp2_nparams.argtypes = [C.c_int64] * 3
        _lib.rlo_ppo_nparams.restype = C.c_int64
        _lib.rlo_ring_length.restype = C.c_int64
        _lib.rlo_sumtree_nodes.restype = C.c_int64
        _lib.rlo_sumtree_nodes.argtypes = [C.c_int64]
        _lib.rlo_dqn_loss_grad_f32.restype = C.c_float
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u8(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.uint8))


# ------------------------------------------------------------------------------------------ RNG
def philox(seed, idx, blk, t, tag):
    out = (C.c_uint32 * 4)()
    lib().rlo_philox4x32_10(C.c_uint64(seed), C.c_uint32(idx), C.c_uint32(blk), C.c_uint32(t),
                            C.c_uint32(tag), out)
    return [int(x) for x in out]


def u01_f32(w):
    return float(lib().rlo_u01_f32(C.c_uint32(w)))


def u01_f64(hi, lo):
    return float(lib().rlo_u01_f64(C.c_uint32(hi), C.c_uint32(lo)))


def permute(seed, epoch, n, i):
    return int(lib().rlo_permute(seed, epoch, n, i))


def permutation(seed, epoch, n):
    return np.array([permute(seed, epoch, n, i) for i in range(n)], dtype=np.int64)


def fill_uniform_f32(n, seed, t, tag):
    out = np.empty(n, np.float32)
    lib().rlo_fill_uniform_f32(_p(out), C.c_int64(n), C.c_uint64(seed), C.c_uint32(t), C.c_uint32(tag))
    return out


# ----------------------------------------------------------------------------------------- envs
def default_cfg(kind, continuous=None, **kw):
    k = KIND[kind] if isinstance(kind, str) else kind
    if k == 0:
        c = CartPoleCfg()
        lib().rlo_cartpole_default(C.byref(c))
        if continuous is not None:
            c.continuous = int(continuous)
    elif k == 1:
        c = PendulumCfg()
        lib().rlo_pendulum_default(C.byref(c))
        if continuous is not None:
            c.continuous = int(continuous)
    elif k == 2:
        c = MountainCarCfg()
        lib().rlo_mountaincar_default(C.byref(c), C.c_int(int(bool(continuous))))
    else:
        c = AcrobotCfg()
        lib().rlo_acrobot_default(C.byref(c))
    for key, val in kw.items():
        key = {"thetathreshold": "thetathreshold_deg"}.get(key, key)  # the reference's keyword (degrees, CartPoleEnv.jl:30)
        if not hasattr(c, key):
            raise TypeError(f"unknown env kwarg {key}")
        setattr(c, key, val)
    return c


class VecEnv:
    """Oracle vector env: n independent instances of one classic-control env (SoA numpy arrays)."""

    def __init__(self, kind, n, seed=0, env_id_base=0, dtype=np.float32, auto_reset=True, cfg=None,
                 **kw):
        self.kind = KIND[kind] if isinstance(kind, str) else kind
        self.n = n
        self.seed = seed
        self.env_id_base = env_id_base
        self.dtype = np.dtype(dtype)
        self.is_f64 = int(self.dtype == np.float64)
        self.auto_reset = auto_reset
        self.cfg = cfg if cfg is not None else default_cfg(self.kind, **kw)
        self.sdim = lib().rlo_env_state_dim(self.kind)
        self.odim = lib().rlo_env_obs_dim(self.kind)
        self.s = [np.zeros(n, self.dtype) for _ in range(self.sdim)]
        self.t = np.zeros(n, np.int32)
        self.done = np.zeros(n, np.uint8)
        self.reward = np.zeros(n, self.dtype)
        self.episode = np.zeros(n, np.uint32)
        self.last_obs = np.zeros((self.odim, n), self.dtype)
        self._st = EnvStateC()
        for k in range(self.sdim):
            self._st.s[k] = self.s[k].ctypes.data
        self._st.t = self.t.ctypes.data
        self._st.done = self.done.ctypes.data
        self._st.reward = self.reward.ctypes.data
        self._st.episode = self.episode.ctypes.data
        self.reset()

    def reset(self, mask=None):
        m = _u8(mask)
        lib().rlo_env_reset(self.kind, self.is_f64, C.byref(self.cfg), C.byref(self._st),
                            C.c_int64(self.n), C.c_uint64(self.seed), C.c_uint32(self.env_id_base), _p(m))

    def set_state(self, s, t=None):
        for k in range(self.sdim):
            self.s[k][:] = s[k]
        if t is not None:
            self.t[:] = t

    def step(self, actions):
        continuous = bool(getattr(self.cfg, "continuous", 0))
        a = np.ascontiguousarray(actions, dtype=self.dtype if continuous else np.int32)
        lib().rlo_env_step(self.kind, self.is_f64, C.byref(self.cfg), C.byref(self._st),
                           C.c_int64(self.n), _p(a), C.c_int(int(self.auto_reset)),
                           C.c_uint64(self.seed), C.c_uint32(self.env_id_base), _p(self.last_obs))

    def obs(self):
        o = np.zeros((self.odim, self.n), self.dtype)
        lib().rlo_env_obs(self.kind, self.is_f64, C.byref(self._st), C.c_int64(self.n), _p(o))
        return o


# ---------------------------------------------------------------------------------------- scans
def _colmajor(a, dtype):
    a = np.asarray(a, dtype=dtype)
    return np.asfortranarray(a)


def _scan_call(fn_base, dtype, out, r, extra, gamma_args, terminal, init, dims):
    sfx = "f64" if dtype == np.float64 else "f32"
    ct = C.c_double if dtype == np.float64 else C.c_float
    f = getattr(lib(), f"{fn_base}_{sfx}")
    n1, n2 = (r.shape[0], 1) if r.ndim == 1 else r.shape
    term = None if terminal is None else np.asfortranarray(np.asarray(terminal, dtype=np.uint8))
    ini = None if init is None else np.ascontiguousarray(np.atleast_1d(np.asarray(init, dtype=dtype)))
    args = [_p(out), _p(r)] + [_p(e) for e in extra] + [C.c_int64(n1), C.c_int64(n2)] + \
           [ct(g) for g in gamma_args] + [_p(term)]
    if init is not False:
        args.append(_p(ini))
    args.append(C.c_int(dims))
    rc = f(*args)
    if rc != 0:
        raise TypeError("MethodError: matrix input requires dims = 1 or 2")
    return out


def discount_rewards(rewards, gamma, terminal=None, init=None, dims=0, dtype=np.float64):
    r = _colmajor(rewards, dtype)
    out = np.empty_like(r, order="F")
    return _scan_call("rlo_discount_rewards", dtype, out, r, [], [gamma], terminal, init, dims)


def discount_rewards_reduced(rewards, gamma, terminal=None, init=None, dims=0, dtype=np.float64):
    r = _colmajor(rewards, dtype)
    if r.ndim == 1:
        out = np.empty(1, dtype)
    elif dims in (1, 2):
        out = np.empty(r.shape[1] if dims == 1 else r.shape[0], dtype)
    else:
        out = np.empty(1, dtype)
    return _scan_call("rlo_discount_rewards_reduced", dtype, out, r, [], [gamma], terminal, init, dims)


def generalized_advantage_estimation(rewards, values, gamma, lam, terminal=None, dims=0,
                                     dtype=np.float64):
    r = _colmajor(rewards, dtype)
    v = _colmajor(values, dtype)
    out = np.empty_like(r, order="F")
    return _scan_call("rlo_gae", dtype, out, r, [v], [gamma, lam], terminal, False, dims)


# ------------------------------------------------------------------------------------ selection
def find_all_max(x, mask=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    idx = np.empty(len(x), np.int64)
    vmax = C.c_double(0)
    m = _u8(mask)
    c = lib().rlo_find_all_max_f64(_p(x), C.c_int64(len(x)), _p(m), C.byref(vmax), _p(idx))
    return vmax.value, idx[:c].copy()


def findmax(x, mask=None, dtype=np.float64):
    x = np.ascontiguousarray(x, dtype=dtype)
    f = lib().rlo_findmax_f64 if dtype == np.float64 else lib().rlo_findmax_f32
    return int(f(_p(x), C.c_int64(len(x)), _p(_u8(mask))))


def get_eps(kind, eps_stable, eps_init, warmup_steps, decay_steps, step):
    return float(lib().rlo_get_eps({"linear": 0, "exp": 1}[kind], eps_stable, eps_init, warmup_steps,
                                   decay_steps, step))


def eps_greedy_select(values, eps, seed, step, env_id_base=0, mask=None, is_break_tie=False):
    """values: (na, n) array (one column per env). Returns 0-based int32 actions."""
    v = np.asfortranarray(np.asarray(values, dtype=np.float32))
    na, n = v.shape
    m = None if mask is None else np.asfortranarray(np.asarray(mask, dtype=np.uint8))
    out = np.empty(n, np.int32)
    lib().rlo_eps_greedy_select_f32(_p(v), C.c_int64(na), C.c_int64(n), _p(m), C.c_double(eps),
                                    C.c_int(int(is_break_tie)), C.c_uint64(seed),
                                    C.c_uint32(env_id_base), C.c_uint32(step), _p(out))
    return out


def explorer_select(kind, values, seed, step, env_id_base=0, mask=None, is_normalized=False):
    """kind: 0 / "weighted", 1 / "weighted_softmax", 2 / "gumbel_softmax".  values (na, n).  0-based actions."""
    kind = {"weighted": 0, "weighted_softmax": 1, "gumbel_softmax": 2}.get(kind, kind)
    v = np.asfortranarray(np.asarray(values, dtype=np.float32))
    na, n = v.shape
    m = None if mask is None else np.asfortranarray(np.asarray(mask, dtype=np.uint8))
    out = np.empty(n, np.int32)
    lib().rlo_explorer_select_f32(C.c_int(kind), _p(v), C.c_int64(na), C.c_int64(n), _p(m), C.c_int(int(is_normalized)),
                                  C.c_uint64(seed), C.c_uint32(env_id_base), C.c_uint32(step), _p(out))
    return out


def ucb_select(values, c, counts, step, seed, env_id_base=0):
    """values (na, n); counts (na, n) float64 C-contiguous, updated in place.  0-based actions."""
    v = np.asfortranarray(np.asarray(values, dtype=np.float32))
    na, n = v.shape
    assert counts.dtype == np.float64 and counts.shape == (na, n) and counts.flags.c_contiguous
    out = np.empty(n, np.int32)
    lib().rlo_ucb_select_f32(_p(v), C.c_int64(na), C.c_int64(n), C.c_double(c), _p(counts), C.c_int64(step),
                             C.c_uint64(seed), C.c_uint32(env_id_base), _p(out))
    return out


def eps_greedy_prob(values, eps, mask=None, is_break_tie=False):
    v = np.ascontiguousarray(values, dtype=np.float64)
    out = np.empty(len(v), np.float64)
    lib().rlo_eps_greedy_prob_f64(_p(v), C.c_int64(len(v)), _p(_u8(mask)), C.c_double(eps),
                                  C.c_int(int(is_break_tie)), _p(out))
    return out


def categorical_sample(logits, seed, step, env_id_base=0, mask=None):
    l = np.asfortranarray(np.asarray(logits, dtype=np.float32))
    na, n = l.shape
    m = None if mask is None else np.asfortranarray(np.asarray(mask, dtype=np.uint8))
    a = np.empty(n, np.int32)
    lp = np.empty(n, np.float32)
    lib().rlo_categorical_sample_f32(_p(l), C.c_int64(na), C.c_int64(n), _p(m), C.c_uint64(seed),
                                     C.c_uint32(env_id_base), C.c_uint32(step), _p(a), _p(lp))
    return a, lp


# -------------------------------------------------------------------------------------- updates
def polyak(dst, src, rho):
    lib().rlo_polyak_f32(_p(dst), _p(src), C.c_int64(dst.size), C.c_float(rho))
    return dst


def target_sync_due(n_optimise, sync_freq):
    c = C.c_int64(n_optimise)
    due = lib().rlo_target_sync_due(C.byref(c), C.c_int64(sync_freq))
    return bool(due), c.value


def clip_by_global_norm(g, clip_norm):
    return float(lib().rlo_clip_by_global_norm_f32(_p(g), C.c_int64(g.size), C.c_float(clip_norm)))


def adam(p, g, m, v, lr, beta1, beta2, eps, t):
    lib().rlo_adam_f32(_p(p), _p(g), _p(m), _p(v), C.c_int64(p.size), C.c_float(lr), C.c_float(beta1),
                       C.c_float(beta2), C.c_float(eps), C.c_int64(t))


def normlogpdf(mu, sigma, x):
    return float(lib().rlo_normlogpdf_f32(mu, sigma, x))


def diagnormlogpdf(mu, sigma, x):
    mu, sigma, x = (np.asfortranarray(np.asarray(a, np.float32)) for a in (mu, sigma, x))
    d, n = mu.shape
    out = np.empty(n, np.float32)
    lib().rlo_diagnormlogpdf_f32(_p(mu), _p(sigma), _p(x), C.c_int64(d), C.c_int64(n), _p(out))
    return out


def huber(q, target, delta=1.0):
    q = np.ascontiguousarray(q, np.float32)
    target = np.ascontiguousarray(target, np.float32)
    dq = np.empty_like(q)
    loss = lib().rlo_huber_f32(_p(q), _p(target), C.c_int64(q.size), C.c_float(delta), _p(dq))
    return float(loss), dq


def td_target(qt_next, r, terminal, gamma):
    q = np.asfortranarray(np.asarray(qt_next, np.float32))
    na, n = q.shape
    r = np.ascontiguousarray(r, np.float32)
    out = np.empty(n, np.float32)
    lib().rlo_td_target_f32(_p(q), C.c_int64(na), C.c_int64(n), _p(r), _p(_u8(terminal)),
                            C.c_float(gamma), _p(out))
    return out


# ----------------------------------------------------------------------------------------- ring
class Ring:
    def __init__(self, capacity, n_env, obs_dim):
        self.state = np.zeros(((capacity + 1), obs_dim, n_env), np.float32)
        self.action = np.zeros((capacity, n_env), np.int32)
        self.reward = np.zeros((capacity, n_env), np.float32)
        self.terminal = np.zeros((capacity, n_env), np.uint8)
        self.rb = RingC()
        lib().rlo_ring_init(C.byref(self.rb), C.c_int64(capacity), C.c_int64(n_env),
                            C.c_int64(obs_dim), _p(self.state), _p(self.action), _p(self.reward),
                            _p(self.terminal))

    def push_state(self, obs):
        o = np.ascontiguousarray(obs, np.float32)
        lib().rlo_ring_push_state(C.byref(self.rb), _p(o))

    def push_transition(self, next_obs, action, reward, terminal):
        lib().rlo_ring_push_transition(C.byref(self.rb), _p(np.ascontiguousarray(next_obs, np.float32)),
                                       _p(np.ascontiguousarray(action, np.int32)),
                                       _p(np.ascontiguousarray(reward, np.float32)), _p(_u8(terminal)))

    def __len__(self):
        return int(lib().rlo_ring_length(C.byref(self.rb)))

    def sample_indices(self, batch, seed, draw_ctr):
        idx = np.empty(batch, np.int64)
        lib().rlo_ring_sample_indices(C.byref(self.rb), C.c_int64(batch), C.c_uint64(seed),
                                      C.c_uint32(draw_ctr), _p(idx))
        return idx

    def gather(self, idx):
        b = len(idx)
        d = self.rb.obs_dim
        s = np.empty((d, b), np.float32)
        sn = np.empty((d, b), np.float32)
        a = np.empty(b, np.int32)
        r = np.empty(b, np.float32)
        t = np.empty(b, np.uint8)
        idx = np.ascontiguousarray(idx, np.int64)
        lib().rlo_ring_gather(C.byref(self.rb), _p(idx), C.c_int64(b), _p(s), _p(a), _p(r), _p(t), _p(sn))
        return s, a, r, t, sn


def ring_sample_indices_nstep(ring, batch, n_step, seed, draw_ctr):
    idx = np.empty(batch, np.int64)
    lib().rlo_ring_sample_indices_nstep(C.byref(ring.rb), C.c_int64(batch), C.c_int64(n_step), C.c_uint64(seed),
                                        C.c_uint32(draw_ctr), _p(idx))
    return idx


def ring_gather_nstep(ring, idx, n_step, gamma):
    """-> (s, a, R, t, s_n): the n-step transitions starting at the flat logical indices idx (NStepBatchSampler)"""
    b, d = len(idx), ring.rb.obs_dim
    s, sn = np.empty((d, b), np.float32), np.empty((d, b), np.float32)
    a, r, t = np.empty(b, np.int32), np.empty(b, np.float32), np.empty(b, np.uint8)
    idx = np.ascontiguousarray(idx, np.int64)
    lib().rlo_ring_gather_nstep(C.byref(ring.rb), _p(idx), C.c_int64(b), C.c_int64(n_step), C.c_float(gamma), _p(s), _p(a),
                                _p(r), _p(t), _p(sn))
    return s, a, r, t, sn


def gamma_pow(gamma, n):
    f = lib().rlo_gamma_pow
    f.restype = C.c_float
    return float(f(C.c_float(gamma), C.c_int64(n)))


class SumTree:
    """priority sum-tree (rlo_buffer.c); leaves keyed 0-based"""

    def __init__(self, n_leaves):
        self.n_leaves = int(n_leaves)
        self.tree = np.zeros(int(lib().rlo_sumtree_nodes(self.n_leaves)), np.float32)
        self.P = self.tree.size // 2

    def fill_range(self, start, count, value):
        lib().rlo_sumtree_fill_range(_p(self.tree), C.c_int64(self.n_leaves), C.c_int64(start), C.c_int64(count),
                                     C.c_float(value))

    def update(self, keys, prio):
        k = np.ascontiguousarray(keys, np.int64)
        p = np.ascontiguousarray(prio, np.float32)
        lib().rlo_sumtree_update(_p(self.tree), C.c_int64(self.n_leaves), _p(k), _p(p), C.c_int64(k.size))

    def sample(self, batch, seed, draw_ctr):
        leaf = np.empty(batch, np.int64)
        prio = np.empty(batch, np.float32)
        lib().rlo_sumtree_sample(_p(self.tree), C.c_int64(self.n_leaves), C.c_int64(batch), C.c_uint64(seed),
                                 C.c_uint32(draw_ctr), _p(leaf), _p(prio))
        return leaf, prio

    def leaves(self):
        return self.tree[self.P:self.P + self.n_leaves]


def per_priority(td, eps, alpha):
    td = np.ascontiguousarray(td, np.float32)
    out = np.empty_like(td)
    lib().rlo_per_priority_f32(_p(td), C.c_int64(td.size), C.c_float(eps), C.c_float(alpha), _p(out))
    return out


def per_is_weights(prio, beta):
    prio = np.ascontiguousarray(prio, np.float32)
    out = np.empty_like(prio)
    lib().rlo_per_is_weights_f32(_p(prio), C.c_int64(prio.size), C.c_float(beta), _p(out))
    return out


def ring_push_priority(ring, st, priority):
    lib().rlo_ring_push_priority(C.byref(ring.rb), _p(st.tree), C.c_float(priority))


def ring_sample_prioritized(ring, st, batch, seed, draw_ctr):
    idx = np.empty(batch, np.int64)
    key = np.empty(batch, np.int64)
    prio = np.empty(batch, np.float32)
    lib().rlo_ring_sample_prioritized(C.byref(ring.rb), _p(st.tree), C.c_int64(batch), C.c_uint64(seed),
                                      C.c_uint32(draw_ctr), _p(idx), _p(key), _p(prio))
    return idx, key, prio


def ring_gather_stacked(ring, idx, n_stack):
    """-> (s (batch, n_stack, obs_dim), a, r, t, s_next) for a single-env ring of single frames"""
    b, d = len(idx), ring.rb.obs_dim
    s = np.empty((b, n_stack, d), np.float32)
    sn = np.empty((b, n_stack, d), np.float32)
    a, r, t = np.empty(b, np.int32), np.empty(b, np.float32), np.empty(b, np.uint8)
    idx = np.ascontiguousarray(idx, np.int64)
    lib().rlo_ring_gather_stacked(C.byref(ring.rb), _p(idx), C.c_int64(b), C.c_int64(n_stack), _p(s), _p(a), _p(r),
                                  _p(t), _p(sn))
    return s, a, r, t, sn


# ------------------------------------------------------------------------------------------ MLP
def mlp2_nparams(n_in, h, n_out):
    return int(lib().rlo_mlp2_nparams(n_in, h, n_out))


def mlp2_init(n_in, h, n_out, seed, net_id):
    p = np.empty(mlp2_nparams(n_in, h, n_out), np.float32)
    lib().rlo_mlp2_init_f32(_p(p), C.c_int64(n_in), C.c_int64(h), C.c_int64(n_out), C.c_uint64(seed),
                            C.c_uint32(net_id))
    return p


def mlp2_forward(p, n_in, h, n_out, act, x):
    """x: (n_in, batch) SoA C-contiguous.  Returns (n_out, batch)."""
    x = np.ascontiguousarray(x, np.float32)
    batch = x.shape[1]
    out = np.empty((n_out, batch), np.float32)
    lib().rlo_mlp2_forward_f32(_p(p), C.c_int64(n_in), C.c_int64(h), C.c_int64(n_out), C.c_int(act),
                               _p(x), C.c_int64(batch), _p(out))
    return out


def mlp2_backward(p, n_in, h, n_out, act, x, dout):
    x = np.ascontiguousarray(x, np.float32)
    dout = np.ascontiguousarray(dout, np.float32)
    g = np.zeros_like(p)
    lib().rlo_mlp2_backward_f32(_p(p), C.c_int64(n_in), C.c_int64(h), C.c_int64(n_out), C.c_int(act),
                                _p(x), C.c_int64(x.shape[1]), _p(dout), _p(g))
    return g


def ppo_default(**kw):
    c = PPOCfg()
    lib().rlo_ppo_default(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise TypeError(f"unknown PPO kwarg {k}")
        setattr(c, k, v)
    return c


def ppo_nparams(kind, cfg):
    return int(lib().rlo_ppo_nparams(C.c_int(kind), C.byref(cfg)))


def ppo_loss_grad(cfg, ns, na, params, obs, action, logp_old, adv, ret):
    obs = np.ascontiguousarray(obs, np.float32)
    bm = obs.shape[1]
    grad = np.zeros_like(params)
    losses = np.zeros(4, np.float32)
    if cfg.continuous:
        af = np.ascontiguousarray(action, np.float32)
        ai = None
    else:
        ai = np.ascontiguousarray(action, np.int32)
        af = None
    lib().rlo_ppo_loss_grad_f32(C.byref(cfg), C.c_int64(ns), C.c_int64(na), _p(params), _p(obs), _p(ai),
                                _p(af), _p(np.ascontiguousarray(logp_old, np.float32)),
                                _p(np.ascontiguousarray(adv, np.float32)),
                                _p(np.ascontiguousarray(ret, np.float32)), C.c_int64(bm), _p(grad),
                                _p(losses))
    return grad, losses


def dqn_loss_grad(ns, h, na, act, params, target_params, s, a, r, term, s_next, gamma, delta=1.0, weights=None):
    s = np.ascontiguousarray(s, np.float32)
    s_next = np.ascontiguousarray(s_next, np.float32)
    grad = np.zeros_like(params)
    loss = lib().rlo_dqn_loss_grad_f32(C.c_int64(ns), C.c_int64(h), C.c_int64(na), C.c_int(act),
                                       _p(params), _p(target_params), _p(s),
                                       _p(np.ascontiguousarray(a, np.int32)),
                                       _p(np.ascontiguousarray(r, np.float32)), _p(_u8(term)),
                                       _p(s_next), C.c_int64(s.shape[1]), C.c_float(gamma),
                                       C.c_float(delta), _p(grad),
                                       _p(None if weights is None else np.ascontiguousarray(weights, np.float32)))
    return float(loss), grad


# ------------------------------------------------------------------------- 3-layer bf16 Q-net
def bf16_round(x):
    x = np.ascontiguousarray(x, np.float32)
    f = lib().rlo_bf16_round_f32
    f.restype, f.argtypes = C.c_float, [C.c_float]
    return np.array([f(float(v)) for v in x.ravel()], np.float32).reshape(x.shape)


def mlp3_nparams(ns, h, na):
    f = lib().rlo_mlp3_nparams
    f.restype, f.argtypes = C.c_int64, [C.c_int64] * 3
    return int(f(ns, h, na))


def mlp3_init(ns, h, na, seed, net_id):
    p = np.empty(mlp3_nparams(ns, h, na), np.float32)
    lib().rlo_mlp3_init_f32(_p(p), C.c_int64(ns), C.c_int64(h), C.c_int64(na), C.c_uint64(seed), C.c_uint32(net_id))
    return p


def mlp3_forward(p, ns, h, na, act, x):
    """x: (ns, batch) SoA.  Returns (na, batch)."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty((na, x.shape[1]), np.float32)
    lib().rlo_mlp3_forward_f32(_p(np.ascontiguousarray(p, np.float32)), C.c_int64(ns), C.c_int64(h), C.c_int64(na),
                               C.c_int(act), _p(x), C.c_int64(x.shape[1]), _p(out))
    return out


def dqn3_loss_grad(ns, h, na, act, params, target_params, s, a, r, term, s_next, gamma, delta=1.0, weights=None):
    s = np.ascontiguousarray(s, np.float32)
    s_next = np.ascontiguousarray(s_next, np.float32)
    params = np.ascontiguousarray(params, np.float32)
    target_params = np.ascontiguousarray(target_params, np.float32)
    grad = np.zeros_like(params)
    q = np.empty((na, s.shape[1]), np.float32)
    f = lib().rlo_dqn3_loss_grad_f32
    f.restype = C.c_float
    loss = f(C.c_int64(ns), C.c_int64(h), C.c_int64(na), C.c_int(act), _p(params), _p(target_params), _p(s),
             _p(np.ascontiguousarray(a, np.int32)), _p(np.ascontiguousarray(r, np.float32)), _p(_u8(term)),
             _p(s_next), C.c_int64(s.shape[1]), C.c_float(gamma), C.c_float(delta), _p(grad), _p(q),
             _p(None if weights is None else np.ascontiguousarray(weights, np.float32)))
    return float(loss), grad, q


class PPOTraj:
    """Trajectory buffers of one PPO iteration (time-major)."""

    def __init__(self, kind, n, T, na=1, continuous=False):
        ns = lib().rlo_env_obs_dim(kind)
        self.n, self.T, self.ns, self.na = n, T, ns, na
        self.obs = np.zeros((T + 1, ns, n), np.float32)
        self.logp = np.zeros((T, n), np.float32)
        self.value = np.zeros((T + 1, n), np.float32)
        self.reward = np.zeros((T, n), np.float32)
        self.adv = np.zeros((T, n), np.float32)
        self.ret = np.zeros((T, n), np.float32)
        self.action_f = np.zeros((T, na, n), np.float32)
        self.action_i = np.zeros((T, n), np.int32)
        self.terminal = np.zeros((T, n), np.uint8)
        self.c = PPOTrajC()
        for name in ("obs", "logp", "value", "reward", "adv", "ret", "action_f", "action_i", "terminal"):
            setattr(self.c, name, getattr(self, name).ctypes.data)


def ppo_rollout(env, T, cfg, params, traj, vec_step0):
    return lib().rlo_ppo_rollout_f32(C.c_int(env.kind), C.byref(env.cfg), C.byref(env._st),
                                     C.c_int64(env.n), C.c_int64(T), C.byref(cfg), _p(params),
                                     C.c_uint64(env.seed), C.c_uint32(env.env_id_base),
                                     C.c_uint32(vec_step0), C.byref(traj.c))


def ppo_gae(cfg, traj):
    lib().rlo_ppo_gae_f32(C.byref(cfg), C.c_int64(traj.n), C.c_int64(traj.T), C.byref(traj.c))


def ppo_update(kind, cfg, traj, params, m, v, opt_step, seed, update_ctr):
    st = C.c_int64(opt_step)
    losses = np.zeros(4, np.float32)
    lib().rlo_ppo_update_f32(C.c_int(kind), C.byref(cfg), C.c_int64(traj.n), C.c_int64(traj.T),
                             C.byref(traj.c), _p(params), _p(m), _p(v), C.byref(st), C.c_uint64(seed),
                             C.c_uint32(update_ctr), _p(losses))
    return st.value, losses


# ------------------------------------------------------------------------------- stochastic heads
def gaussian_head_sample(mu, raw_sigma, K=1, min_sigma=0.0, max_sigma=np.inf, squash=0, soft=0, seed=0, env_id_base=0,
                         step=0, want_logp=True):
    """mu, raw_sigma (d, n) Julia-shaped -> actions (d, K, n), logp (K, n)  (rlo_heads.c)"""
    d, n = mu.shape
    m = np.ascontiguousarray(mu.T, np.float32)
    s = np.ascontiguousarray(raw_sigma.T, np.float32)
    act = np.zeros((n, K, d), np.float32)
    lp = np.zeros((n, K), np.float32) if want_logp else None
    rc = lib().rlo_gaussian_head_sample_f32(_p(m), _p(s), C.c_int64(d), C.c_int64(n), C.c_int64(K), C.c_float(min_sigma),
                                            C.c_float(max_sigma), C.c_int(squash), C.c_int(soft), C.c_uint64(seed),
                                            C.c_uint32(env_id_base), C.c_uint32(step), _p(act), _p(lp))
    assert rc == 0
    return act.transpose(2, 1, 0), (lp.T if want_logp else None)


def gaussian_head_logp(mu, raw_sigma, action, min_sigma=0.0, max_sigma=np.inf, squash=0, soft=0):
    """action (d, K, n) -> logp (K, n)"""
    d, n = mu.shape
    K = action.shape[1]
    m = np.ascontiguousarray(mu.T, np.float32)
    s = np.ascontiguousarray(raw_sigma.T, np.float32)
    a = np.ascontiguousarray(action.transpose(2, 1, 0), np.float32)
    lp = np.zeros((n, K), np.float32)
    rc = lib().rlo_gaussian_head_logp_f32(_p(m), _p(s), _p(a), C.c_int64(d), C.c_int64(n), C.c_int64(K),
                                          C.c_float(min_sigma), C.c_float(max_sigma), C.c_int(squash), C.c_int(soft), _p(lp))
    assert rc == 0
    return lp.T
