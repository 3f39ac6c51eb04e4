// p2p.hip -- one-shot peer-to-peer all-reduce of the (tiny) flat policy gradient over xGMI, inside one kernel on the
// compute stream.
//
// Why: the exchange step of the data-parallel learner is a sum of a 13 KB vector per optimiser step.
// A library all-reduce costs a stream hop to the communicator's stream and back plus a ring / tree protocol whose
// latency is an order of magnitude above the transfer time of 13 KB; on MI355X every GPU has a direct xGMI link to
// each of its 7 peers, so the latency-optimal algorithm is "everyone reads everyone": each rank publishes its
// gradient in a buffer the peers have mapped (HIP IPC), raises a sequence flag, and sums the world's buffers IN RANK
// ORDER -- every rank adds the same numbers in the same order, so the replicas stay bit-identical without a
// broadcast.  RCCL stays the fallback (the host runs a self-test against it before trusting this path) and the
// transport for anything large.
//
// Protocol (pull model, double-buffered by the parity of the sequence number seq = 1, 2, ...):
//   comm buffer of rank r (uncached device memory, mapped by every peer):  slots[2][cap] floats | flags[2] u32
//   1. copy the local vector into slots[seq & 1], system-scope release, flags[seq & 1] := seq
//   2. for p = 0 .. world-1 (rank order): poll p's flags[seq & 1] == seq (system-scope acquire), add p's slot
//   3. write the sum back to the local vector
//   A slot is rewritten at seq + 2; by then every peer has finished reading seq (it had to publish seq + 1 first,
//   and kernels of one rank run in stream order), so two slots suffice.  A poll gives up after `timeout_polls`
//   iterations, raises status[0] and overwrites the result with NaN (the host polls the status word: comm.hip).
#include "common.h"

#include <string.h>

namespace rlhip {

constexpr int P2P_MAX_WORLD = 16;

struct P2PPeers {
    float* slot[P2P_MAX_WORLD];          // base of each rank's comm buffer (slot 0; slot 1 follows at +cap floats)
    unsigned int* flags[P2P_MAX_WORLD];  // each rank's two sequence flags
};

__global__ __launch_bounds__(1024) void p2p_allreduce_kernel(float* __restrict__ data, int n, int cap, int rank,
                                                             int world, P2PPeers peers, unsigned int seq,
                                                             long long timeout_polls, int* __restrict__ status) {
    __shared__ int l_fail;
    const int par = (int)(seq & 1u);
    float* mine = peers.slot[rank] + (int64_t)par * cap;
    // 1. publish
    for (int i = threadIdx.x; i < n; i += blockDim.x) __builtin_nontemporal_store(data[i], mine + i);
    if (threadIdx.x == 0) l_fail = 0;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(peers.flags[rank] + par, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // 2. wait for every rank's flag (one lane polls, the workgroup follows)
    if (threadIdx.x == 0) {
        for (int p = 0; p < world; ++p) {
            long long polls = 0;
            while (__hip_atomic_load(peers.flags[p] + par, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
                if (++polls > timeout_polls) {
                    l_fail = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (l_fail) break;
        }
    }
    __syncthreads();
    if (l_fail) {
        // a peer never arrived: an unreduced gradient must not reach the optimiser silently -- poison the result (the
        // next exchange spreads the NaNs to every replica) and raise the status word the host polls (comm.hip)
        if (threadIdx.x == 0) __hip_atomic_store(status, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int i = threadIdx.x; i < n; i += blockDim.x) data[i] = __builtin_nanf("");
        return;
    }
    __threadfence_system();  // acquire side for the whole workgroup
    // 3. sum in rank order (identical on every rank)
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float acc = 0.0f;
        for (int p = 0; p < world; ++p) acc += __builtin_nontemporal_load(peers.slot[p] + (int64_t)par * cap + i);
        data[i] = acc;
    }
}

}  // namespace rlhip

using namespace rlhip;

extern "C" {

int32_t rlhip_p2p_alloc(int64_t bytes, void** out) {
    RLHIP_REQUIRE(out != nullptr && bytes > 0, "bad arguments");
    RLHIP_CHECK_HIP(hipExtMallocWithFlags(out, (size_t)bytes, hipDeviceMallocUncached));
    RLHIP_CHECK_HIP(hipMemset(*out, 0, (size_t)bytes));
    RLHIP_CHECK_HIP(hipDeviceSynchronize());
    return RLHIP_OK;
}

int32_t rlhip_p2p_free(void* p) {
    if (p) RLHIP_CHECK_HIP(hipFree(p));
    return RLHIP_OK;
}

int32_t rlhip_p2p_export(void* p, uint8_t handle_out[64]) {
    RLHIP_REQUIRE(p && handle_out, "NULL argument");
    hipIpcMemHandle_t h;
    RLHIP_CHECK_HIP(hipIpcGetMemHandle(&h, p));
    memcpy((void*)handle_out, (const void*)&h, 64);
    return RLHIP_OK;
}

int32_t rlhip_p2p_import(const uint8_t handle[64], void** out) {
    RLHIP_REQUIRE(handle && out, "NULL argument");
    hipIpcMemHandle_t h;
    memcpy((void*)&h, (const void*)handle, 64);
    RLHIP_CHECK_HIP(hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess));
    return RLHIP_OK;
}

int32_t rlhip_p2p_close(void* p) {
    if (p) RLHIP_CHECK_HIP(hipIpcCloseMemHandle(p));
    return RLHIP_OK;
}

/* 1 if the current device can address memory of `peer_device` (hipDeviceCanAccessPeer), also 1 for itself */
int32_t rlhip_p2p_can_access(int32_t peer_device) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev == peer_device) return 1;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, dev, peer_device) != hipSuccess) return 0;
    return can ? 1 : 0;
}

/* host-driven probe of a mapped peer buffer: a 4-byte device-to-host copy from `p + byte_offset` (a broken mapping
 * shows up as an error code here instead of a memory fault inside a kernel) */
int32_t rlhip_p2p_probe(const void* p, int64_t byte_offset, uint32_t* value_out) {
    RLHIP_REQUIRE(p && value_out && byte_offset >= 0, "bad arguments");
    RLHIP_CHECK_HIP(hipMemcpy(value_out, (const char*)p + byte_offset, 4, hipMemcpyDeviceToHost));
    return RLHIP_OK;
}

/* comm buffer layout: float slots[2][cap] | uint32 flags[2] (+ padding); bytes = rlhip_p2p_comm_bytes(cap) */
int64_t rlhip_p2p_comm_bytes(int64_t cap) { return 2 * cap * (int64_t)sizeof(float) + 256; }

int32_t rlhip_p2p_allreduce_f32(float* data, int64_t n, int64_t cap, int32_t rank, int32_t world,
                                void* const* comm_bufs_host, uint32_t seq, int64_t timeout_polls, int32_t* status_dev,
                                rlhip_stream_t stream) {
    RLHIP_REQUIRE(data && comm_bufs_host && status_dev, "NULL argument");
    RLHIP_REQUIRE(world >= 1 && world <= P2P_MAX_WORLD && rank >= 0 && rank < world, "bad rank / world");
    RLHIP_REQUIRE(n >= 0 && n <= cap && cap <= (1 << 24), "vector does not fit the comm buffer");
    RLHIP_REQUIRE(seq >= 1, "sequence numbers start at 1");
    if (n == 0) return RLHIP_OK;
    P2PPeers pr;
    for (int p = 0; p < world; ++p) {
        RLHIP_REQUIRE(comm_bufs_host[p] != nullptr, "peer buffer is NULL");
        pr.slot[p] = (float*)comm_bufs_host[p];
        pr.flags[p] = (unsigned int*)((float*)comm_bufs_host[p] + 2 * cap);
    }
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(1), dim3(1024), 0, as_stream(stream), data, (int)n, (int)cap, rank, world,
                       pr, seq, (long long)timeout_polls, status_dev);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

}  // extern "C"
