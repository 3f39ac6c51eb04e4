// comm.hip -- the collective of the sharded learner behind the C ABI: rlhip_comm_* / rlhip_allreduce_grads.
//
// independent env shards per GPU, replicated parameters, ONE exchange per optimiser step -- the SUM
// of the small flat gradient before clip_by_global_norm! (RLCore/src/utils/basic.jl:19-29) and the Adam step of
// `optimise!(::FluxApproximator, grad)` (RLCore/src/policies/learners/flux_approximator.jl:46).  The reference has no
// counterpart (no Distributed / MPI / NCCL anywhere).  A host in ANY language needs only a byte transport (a file, a
// socket, MPI, torch.distributed ...) to move two small blobs between its ranks:
//
//   rank 0:     rlhip_comm_unique_id(id[128])                       -> broadcast id
//   every rank: rlhip_comm_init(rank, world, id, cap, &comm)        RCCL communicator (ncclCommInitRank) + own exchange
//                                                                   buffer; id = NULL: no RCCL communicator
//               rlhip_comm_export(comm, handle[64], &device)        -> all-gather (handle, device)
//               rlhip_p2p_setup(comm, handles, devices, &active)    maps the peers, self-test, cross-rank agreement
//   per step:   rlhip_allreduce_grads(comm, grad, n, stream)        in-place SUM on the caller's stream
//               rlhip_comm_check(comm)                              RLHIP_ETIMEOUT if a peer never arrived
//
// Two transports, one entry point: the one-shot peer-to-peer kernel of p2p.hip (every GPU reads every peer's buffer
// over its direct xGMI link and sums in rank order: latency-optimal for a 13-140 KB gradient, bit-identical on all
// ranks) once rlhip_p2p_setup validated it on EVERY rank, RCCL's ncclAllReduce on the same stream otherwise or for
// vectors that do not fit the exchange buffer.  RCCL is bound at rlhip_comm_init time with dlopen("librccl.so.1"):
// a process that already loaded RCCL (PyTorch-ROCm does) shares that copy, a Julia / C host gets /opt/rocm's, and a
// single-GPU user of librlhip.so never pays for loading a 500 MB collective library.
//
// Failure semantics: a rank whose peer does not arrive within `timeout_polls` sets the communicator's
// host-visible status word AND overwrites its result with NaN, so that the optimiser step that follows cannot silently
// train on an unreduced gradient -- the replica poisons itself, the next exchange spreads the NaNs, and
// rlhip_comm_check() reports RLHIP_ETIMEOUT at the host's next look.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <vector>

extern "C" int32_t rlhip_p2p_allreduce_f32(float* data, int64_t n, int64_t cap, int32_t rank, int32_t world,
                                           void* const* comm_bufs_host, uint32_t seq, int64_t timeout_polls,
                                           int32_t* status_dev, rlhip_stream_t stream);
extern "C" int64_t rlhip_p2p_comm_bytes(int64_t cap);

namespace rlhip {

constexpr int COMM_MAX_WORLD = 16;
constexpr uint32_t COMM_MAGIC = 0x524C434Du;  // "RLCM"

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    char path[256] = "";
};

static RcclApi g_rccl;

static bool rccl_load() {
    if (g_rccl.lib) return true;
    const char* env = getenv("RLHIP_RCCL_PATH");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
        if (!nm || !*nm) continue;
        void* h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (!h) continue;
        RcclApi a;
        a.lib = h;
        a.get_unique_id = (decltype(a.get_unique_id))dlsym(h, "ncclGetUniqueId");
        a.comm_init_rank = (decltype(a.comm_init_rank))dlsym(h, "ncclCommInitRank");
        a.comm_destroy = (decltype(a.comm_destroy))dlsym(h, "ncclCommDestroy");
        a.all_reduce = (decltype(a.all_reduce))dlsym(h, "ncclAllReduce");
        a.error_string = (decltype(a.error_string))dlsym(h, "ncclGetErrorString");
        if (a.get_unique_id && a.comm_init_rank && a.comm_destroy && a.all_reduce && a.error_string) {
            strncpy(a.path, nm, sizeof(a.path) - 1);
            g_rccl = a;
            return true;
        }
        dlclose(h);
    }
    set_error("RCCL not found: dlopen(librccl.so.1) failed (%s); set RLHIP_RCCL_PATH", dlerror());
    return false;
}

#define RLHIP_CHECK_RCCL(expr)                                                                             \
    do {                                                                                                   \
        ncclResult_t _r = (expr);                                                                          \
        if (_r != ncclSuccess) {                                                                           \
            ::rlhip::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.error_string(_r));     \
            return RLHIP_ECOMM;                                                                            \
        }                                                                                                  \
    } while (0)

struct Comm {
    uint32_t magic = COMM_MAGIC;
    int rank = 0, world = 1, device = 0;
    int64_t cap = 0;
    ncclComm_t nccl = nullptr;          // NULL: no RCCL communicator
    void* own = nullptr;                // own exchange buffer (uncached device memory, exported over HIP IPC)
    void* peers[COMM_MAX_WORLD] = {};   // every rank's exchange buffer as mapped here (peers[rank] = own)
    bool imported[COMM_MAX_WORLD] = {};
    bool p2p_active = false;
    uint32_t seq = 0;                   // last sequence number used by the peer-to-peer protocol
    int64_t timeout_polls = 1ll << 26;  // steady state: minutes of polling before a rank gives up
    int32_t* status = nullptr;          // host-pinned, device-visible: [0] = a wait timed out
    float* scratch = nullptr;           // self-test vector (device)
    char why[256] = "rlhip_p2p_setup was not called";  // why the peer-to-peer path is not active
};

static Comm* as_comm(rlhip_comm_t c) {
    Comm* q = reinterpret_cast<Comm*>(c);
    return (q && q->magic == COMM_MAGIC) ? q : nullptr;
}

// the self-test vector of rank r in round k: small integers times 2^-10 -- every partial sum of up to 16 of them is
// exact in Float32, so the expected result does not depend on the summation order and a wrong or stale slot is certain
// to show
static inline float selftest_value(int r, int k, int64_t i) {
    uint32_t h = (uint32_t)(i * 2654435761u) ^ (uint32_t)(r * 40503u + k * 9176u + 12345u);
    h ^= h >> 13;
    h *= 0x5bd1e995u;
    h ^= h >> 15;
    return (float)((int)(h & 0xFFFF) - 32768) * (1.0f / 1024.0f);
}

}  // namespace rlhip

using namespace rlhip;

extern "C" {

int32_t rlhip_comm_unique_id(uint8_t id_out_host[128]) {
    RLHIP_REQUIRE(id_out_host != nullptr, "id_out_host is NULL");
    if (!rccl_load()) return RLHIP_ECOMM;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    RLHIP_CHECK_RCCL(g_rccl.get_unique_id(&id));
    memcpy(id_out_host, &id, 128);
    return RLHIP_OK;
}

int32_t rlhip_comm_init(int32_t rank, int32_t world, const uint8_t* unique_id_host, int64_t cap,
                        rlhip_comm_t* comm_out) {
    RLHIP_REQUIRE(comm_out != nullptr, "comm_out is NULL");
    *comm_out = nullptr;
    RLHIP_REQUIRE(world >= 1 && world <= COMM_MAX_WORLD && rank >= 0 && rank < world, "bad rank / world (world <= 16)");
    RLHIP_REQUIRE(cap >= 1 && cap <= (1 << 24), "cap (floats of the largest vector to exchange) out of range");
    Comm* c = new Comm();
    c->rank = rank;
    c->world = world;
    c->cap = cap;
    auto fail = [&](int32_t code) {  // every error path gives back what was allocated so far
        if (c->own) (void)hipFree(c->own);
        if (c->scratch) (void)hipFree(c->scratch);
        if (c->status) (void)hipHostFree(c->status);
        delete c;
        return code;
    };
    hipError_t e = hipGetDevice(&c->device);
    if (e != hipSuccess) {
        set_error("hipGetDevice: %s", hipGetErrorString(e));
        return fail(RLHIP_EHIP);
    }
    e = hipHostMalloc((void**)&c->status, 64, hipHostMallocMapped);
    if (e != hipSuccess) {
        c->status = nullptr;
        set_error("hipHostMalloc(status): %s", hipGetErrorString(e));
        return fail(RLHIP_EHIP);
    }
    memset(c->status, 0, 64);
    if (world > 1) {
        e = hipExtMallocWithFlags(&c->own, (size_t)rlhip_p2p_comm_bytes(cap), hipDeviceMallocUncached);
        if (e != hipSuccess) c->own = nullptr;
        if (e == hipSuccess) e = hipMemset(c->own, 0, (size_t)rlhip_p2p_comm_bytes(cap));
        if (e == hipSuccess) {
            e = hipMalloc((void**)&c->scratch, (size_t)cap * sizeof(float));
            if (e != hipSuccess) c->scratch = nullptr;
        }
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            set_error("exchange buffer allocation: %s", hipGetErrorString(e));
            return fail(RLHIP_EHIP);
        }
        c->peers[rank] = c->own;
    }
    if (unique_id_host != nullptr) {  // also for world = 1: a one-rank communicator exercises the whole RCCL side
        if (!rccl_load()) return fail(RLHIP_ECOMM);
        ncclUniqueId id;
        memcpy(&id, unique_id_host, 128);
        ncclResult_t r = g_rccl.comm_init_rank(&c->nccl, world, id, rank);  // collective: returns when all ranks joined
        if (r != ncclSuccess) {
            set_error("ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.error_string(r));
            c->nccl = nullptr;
            return fail(RLHIP_ECOMM);
        }
    }
    if (world == 1) {
        snprintf(c->why, sizeof(c->why), "world = 1: nothing to exchange");
        if (c->nccl && hipMalloc((void**)&c->scratch, 64) != hipSuccess) c->scratch = nullptr;
    }
    *comm_out = c;
    return RLHIP_OK;
}

int32_t rlhip_comm_export(rlhip_comm_t comm, uint8_t handle_out_host[64], int32_t* device_out) {
    Comm* c = as_comm(comm);
    RLHIP_REQUIRE(c && handle_out_host && device_out, "bad communicator / NULL argument");
    memset(handle_out_host, 0, 64);
    *device_out = c->device;
    if (c->world == 1) return RLHIP_OK;
    hipIpcMemHandle_t h;
    RLHIP_CHECK_HIP(hipIpcGetMemHandle(&h, c->own));
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    memcpy(handle_out_host, &h, 64);
    return RLHIP_OK;
}

// One agreement round: every rank contributes ok (1 / 0); returns 1 only if all did.  Over RCCL when there is a
// communicator (MIN all-reduce of one int), otherwise over the peer-to-peer path itself: a rank that cannot take part
// simply stays away and the others time out -- any failure anywhere ends with "not active" everywhere.
static int32_t comm_agree(Comm* c, bool ok, int round, bool* all_ok) {
    *all_ok = false;
    if (c->nccl) {
        int32_t* d = (int32_t*)c->scratch;
        int32_t v = ok ? 1 : 0;
        RLHIP_CHECK_HIP(hipMemcpy(d, &v, 4, hipMemcpyHostToDevice));
        RLHIP_CHECK_RCCL(g_rccl.all_reduce(d, d, 1, ncclInt32, ncclMin, c->nccl, nullptr));
        RLHIP_CHECK_HIP(hipStreamSynchronize(nullptr));
        RLHIP_CHECK_HIP(hipMemcpy(&v, d, 4, hipMemcpyDeviceToHost));
        *all_ok = v == 1;
        return RLHIP_OK;
    }
    if (!ok) return RLHIP_OK;  // stays away: the peers time out
    float one = 1.0f, got = 0.0f;
    RLHIP_CHECK_HIP(hipMemcpy(c->scratch, &one, 4, hipMemcpyHostToDevice));
    int32_t rc = rlhip_p2p_allreduce_f32(c->scratch, 1, c->cap, c->rank, c->world, c->peers, ++c->seq, 1ll << 22,
                                         c->status, nullptr);
    if (rc) return rc;
    RLHIP_CHECK_HIP(hipStreamSynchronize(nullptr));
    RLHIP_CHECK_HIP(hipMemcpy(&got, c->scratch, 4, hipMemcpyDeviceToHost));
    *all_ok = c->status[0] == 0 && got == (float)c->world;
    (void)round;
    return RLHIP_OK;
}

int32_t rlhip_p2p_setup(rlhip_comm_t comm, const uint8_t* handles_host, const int32_t* devices_host,
                        int32_t* active_out) {
    Comm* c = as_comm(comm);
    RLHIP_REQUIRE(c && active_out, "bad communicator / NULL argument");
    *active_out = 0;
    c->p2p_active = false;
    if (c->world == 1) return RLHIP_OK;
    RLHIP_REQUIRE(handles_host && devices_host, "handles / devices are NULL");
    // ---- 1. map the peers: nothing touches a peer buffer before the runtime says the two devices can address each other
    // and a host-driven 4-byte copy from the mapped flags read the zero they were initialised to
    bool ok = true;
    snprintf(c->why, sizeof(c->why), "ok");
    const int64_t flags_off = 2 * c->cap * (int64_t)sizeof(float);
    for (int p = 0; p < c->world && ok; ++p) {
        if (p == c->rank) continue;
        if (devices_host[p] != c->device) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, c->device, devices_host[p]) != hipSuccess || !can) {
                snprintf(c->why, sizeof(c->why), "hipDeviceCanAccessPeer(device %d -> device %d of rank %d) is false", c->device,
                         devices_host[p], p);
                ok = false;
                break;
            }
        }
        hipIpcMemHandle_t h;
        memcpy(&h, handles_host + 64 * (size_t)p, 64);
        void* q = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            snprintf(c->why, sizeof(c->why), "hipIpcOpenMemHandle(rank %d): %s", p, hipGetErrorString(e));
            (void)hipGetLastError();
            ok = false;
            break;
        }
        c->peers[p] = q;
        c->imported[p] = true;
        uint32_t val = 0xFFFFFFFFu;
        e = hipMemcpy(&val, (const char*)q + flags_off, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess || val != 0u) {
            snprintf(c->why, sizeof(c->why), "probe of rank %d's mapped buffer failed (%s, read %u)", p,
                     e == hipSuccess ? "ok" : hipGetErrorString(e), val);
            (void)hipGetLastError();
            ok = false;
        }
    }
    bool all_ok = false;
    int32_t rc = comm_agree(c, ok, 0, &all_ok);
    if (rc) return rc;
    if (!all_ok) {
        if (ok) snprintf(c->why, sizeof(c->why), "another rank could not map its peers (or never arrived)");
        // without RCCL the agreement itself ran over the peer-to-peer kernel: the ranks that waited for the one that
        // stayed away timed out and raised the status word.  That is the verdict "not active", not a failure of the run
        // (it made rlhip_comm_check report a timeout on some ranks only)
        c->status[0] = 0;
        return RLHIP_OK;
    }
    // ---- 2. self-test: three exchanges of known vectors; exact comparison with the rank-order sum evaluated on the host
    const int64_t nt = c->cap < 4099 ? c->cap : 4099;
    std::vector<float> x((size_t)nt), want((size_t)nt), got((size_t)nt);
    bool passed = true;
    for (int k = 0; k < 3; ++k) {
        for (int64_t i = 0; i < nt; ++i) {
            x[(size_t)i] = selftest_value(c->rank, k, i);
            float acc = 0.0f;
            for (int r = 0; r < c->world; ++r) acc += selftest_value(r, k, i);
            want[(size_t)i] = acc;
        }
        // a HIP error on THIS rank must not end the call here: the peers are (or will be) waiting in the agreement round below,
        // and with an RCCL communicator that round is a collective -- a rank that returned early would hang all the others.
        // Record the failure, keep consuming sequence numbers in step with the peers, vote "failed".
        hipError_t he = hipMemcpy(c->scratch, x.data(), (size_t)nt * 4, hipMemcpyHostToDevice);
        rc = he == hipSuccess ? rlhip_p2p_allreduce_f32(c->scratch, nt, c->cap, c->rank, c->world, c->peers, ++c->seq, 1ll << 22,
                                                        c->status, nullptr)
                              : (++c->seq, RLHIP_EHIP);
        if (rc == RLHIP_OK) he = hipStreamSynchronize(nullptr);
        if (rc == RLHIP_OK && he == hipSuccess) he = hipMemcpy(got.data(), c->scratch, (size_t)nt * 4, hipMemcpyDeviceToHost);
        if (rc != RLHIP_OK || he != hipSuccess) {
            if (passed)
                snprintf(c->why, sizeof(c->why), "self-test round %d: %s", k,
                         he != hipSuccess ? hipGetErrorString(he) : rlhip_last_error());  // a failed copy set rc by hand: its text is HIP's
            (void)hipGetLastError();
            passed = false;
        } else if (c->status[0] != 0) {
            if (passed) snprintf(c->why, sizeof(c->why), "self-test round %d: a peer's flag never arrived (timeout)", k);
            passed = false;
        } else if (memcmp(got.data(), want.data(), (size_t)nt * 4) != 0) {
            if (passed) snprintf(c->why, sizeof(c->why), "self-test round %d: wrong sum", k);
            passed = false;
        }
        // no early exit: a rank that stopped at round k would have consumed fewer sequence numbers than its peers
    }
    rc = comm_agree(c, passed, 1, &all_ok);
    if (rc) return rc;
    if (passed && !all_ok) snprintf(c->why, sizeof(c->why), "the self-test failed on another rank");
    c->status[0] = 0;  // a failed self-test is not a failure of the run: the library collective takes over
    c->p2p_active = all_ok;
    *active_out = all_ok ? 1 : 0;
    return RLHIP_OK;
}

/* a host that learns -- over its own transport -- that some rank failed its set-up switches the peer-to-peer path off on the
 * ranks where it validated, so that every rank takes the same transport (ncclAllReduce, or the host's own collective) */
int32_t rlhip_comm_disable_p2p(rlhip_comm_t comm, const char* why_host) {
    Comm* c = as_comm(comm);
    RLHIP_REQUIRE(c, "bad communicator");
    c->p2p_active = false;
    snprintf(c->why, sizeof(c->why), "%s", why_host && *why_host ? why_host : "switched off by the host");
    return RLHIP_OK;
}

int32_t rlhip_allreduce_grads(rlhip_comm_t comm, float* grad, int64_t n, rlhip_stream_t stream) {
    Comm* c = as_comm(comm);
    RLHIP_REQUIRE(c && (grad || n == 0) && n >= 0, "bad communicator / arguments");
    if ((c->world == 1 && !c->nccl) || n == 0) return RLHIP_OK;
    if (c->p2p_active && n <= c->cap)
        return rlhip_p2p_allreduce_f32(grad, n, c->cap, c->rank, c->world, c->peers, ++c->seq, c->timeout_polls,
                                       c->status, stream);
    if (c->nccl) {
        RLHIP_CHECK_RCCL(g_rccl.all_reduce(grad, grad, (size_t)n, ncclFloat32, ncclSum, c->nccl, as_stream(stream)));
        return RLHIP_OK;
    }
    set_error("no transport: the peer-to-peer path is not active (%s) and the communicator was created without RCCL",
              c->why);
    return RLHIP_ECOMM;
}

int32_t rlhip_comm_check(rlhip_comm_t comm) {
    Comm* c = as_comm(comm);
    RLHIP_REQUIRE(c, "bad communicator");
    if (__atomic_load_n(&c->status[0], __ATOMIC_ACQUIRE) != 0) {
        set_error("rank %d: a peer did not arrive at a gradient exchange within the timeout; the reduced gradient (and the "
                  "parameters updated from it) were overwritten with NaN on this rank",
                  c->rank);
        return RLHIP_ETIMEOUT;
    }
    return RLHIP_OK;
}

int32_t rlhip_comm_info(rlhip_comm_t comm, rlhip_comm_desc* out) {
    Comm* c = as_comm(comm);
    RLHIP_REQUIRE(c && out, "bad communicator / NULL argument");
    memset(out, 0, sizeof(*out));
    out->rank = c->rank;
    out->world = c->world;
    out->device = c->device;
    out->p2p_active = c->p2p_active ? 1 : 0;
    out->rccl_active = c->nccl ? 1 : 0;
    out->cap = c->cap;
    out->seq = c->seq;
    out->timeout_polls = c->timeout_polls;
    out->status = c->status;
    for (int p = 0; p < c->world; ++p) out->bufs[p] = c->peers[p];
    strncpy(out->why, c->why, sizeof(out->why) - 1);
    strncpy(out->rccl_path, g_rccl.path, sizeof(out->rccl_path) - 1);
    return RLHIP_OK;
}

int32_t rlhip_comm_set_timeout(rlhip_comm_t comm, int64_t timeout_polls) {
    Comm* c = as_comm(comm);
    RLHIP_REQUIRE(c && timeout_polls >= 1, "bad communicator / timeout");
    c->timeout_polls = timeout_polls;
    return RLHIP_OK;
}

/* the fused kernels (rlhip_ppo_update_comm_f32) consume sequence numbers themselves */
int32_t rlhip_comm_advance_seq(rlhip_comm_t comm, uint32_t n_steps) {
    Comm* c = as_comm(comm);
    RLHIP_REQUIRE(c, "bad communicator");
    c->seq += n_steps;
    return RLHIP_OK;
}

/* Tear-down in two phases, with a host barrier between them: rlhip_comm_unmap closes this rank's mappings of the peers'
 * buffers (and the RCCL communicator); only when EVERY rank has done that may any rank free its own, IPC-exported buffer
 * (rlhip_comm_destroy).  rlhip_comm_destroy alone does both for a caller that has no peers left (world = 1, error paths). */
int32_t rlhip_comm_unmap(rlhip_comm_t comm) {
    Comm* c = as_comm(comm);
    RLHIP_REQUIRE(c, "bad communicator");
    (void)hipDeviceSynchronize();
    c->p2p_active = false;
    for (int p = 0; p < c->world; ++p)
        if (c->imported[p] && c->peers[p]) {
            (void)hipIpcCloseMemHandle(c->peers[p]);
            c->peers[p] = nullptr;
            c->imported[p] = false;
        }
    if (c->nccl) {
        (void)g_rccl.comm_destroy(c->nccl);
        c->nccl = nullptr;
    }
    return RLHIP_OK;
}

int32_t rlhip_comm_destroy(rlhip_comm_t comm) {
    Comm* c = as_comm(comm);
    RLHIP_REQUIRE(c, "bad communicator");
    (void)rlhip_comm_unmap(comm);
    if (c->own) (void)hipFree(c->own);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->status) (void)hipHostFree(c->status);
    c->magic = 0;
    delete c;
    return RLHIP_OK;
}

}  // extern "C"
