// runtime.hip -- error reporting, device memory / stream / event helpers of the C ABI.
// These exist so that a host with no GPU allocator of its own (the Julia glue, julia/RLHip.jl)
// can own device buffers; the PyTorch-ROCm host passes torch allocations and streams instead.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace rlhip {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

__global__ void fill_uniform_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint32_t t,
                                    uint32_t tag) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t nq = (n + 3) / 4;
    for (; q < nq; q += stride) {
        u32x4 w = philox4x32_10(seed, (uint32_t)q, 0, t, tag);
        int64_t i = q * 4;
        if (i + 3 < n) {
            float4 v = make_float4(u01_f32(w.x), u01_f32(w.y), u01_f32(w.z), u01_f32(w.w));
            *reinterpret_cast<float4*>(out + i) = v;
        } else {
            uint32_t ws[4] = {w.x, w.y, w.z, w.w};
            for (int k = 0; k < 4 && i + k < n; ++k) out[i + k] = u01_f32(ws[k]);
        }
    }
}

__global__ void permutation_kernel(uint32_t* __restrict__ out, uint32_t n, uint64_t seed,
                                   uint32_t epoch) {
    PermKeys pk = perm_keys(seed, epoch, n);
    uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = permute(pk, i);
}
}  // namespace rlhip

using namespace rlhip;

extern "C" {

int32_t rlhip_abi_version(void) { return RLHIP_ABI_VERSION; }

const char* rlhip_last_error(void) { return g_err; }

int32_t rlhip_device_count(int32_t* n_out) {
    RLHIP_REQUIRE(n_out != nullptr, "n_out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *n_out = 0;
        set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return RLHIP_ENODEV;
    }
    *n_out = n;
    return RLHIP_OK;
}

int32_t rlhip_set_device(int32_t device) {
    RLHIP_CHECK_HIP(hipSetDevice(device));
    return RLHIP_OK;
}

int32_t rlhip_device_name(int32_t device, char* name_host, int32_t cap) {
    RLHIP_REQUIRE(name_host != nullptr && cap > 0, "bad name buffer");
    hipDeviceProp_t p;
    RLHIP_CHECK_HIP(hipGetDeviceProperties(&p, device));
    strncpy(name_host, p.gcnArchName, (size_t)cap - 1);
    name_host[cap - 1] = 0;
    return RLHIP_OK;
}

int32_t rlhip_malloc(void** ptr_out, size_t bytes) {
    RLHIP_REQUIRE(ptr_out != nullptr, "ptr_out is NULL");
    RLHIP_CHECK_HIP(hipMalloc(ptr_out, bytes));
    return RLHIP_OK;
}

int32_t rlhip_free(void* ptr) {
    RLHIP_CHECK_HIP(hipFree(ptr));
    return RLHIP_OK;
}

int32_t rlhip_memset(void* ptr, int32_t value, size_t bytes, rlhip_stream_t stream) {
    RLHIP_CHECK_HIP(hipMemsetAsync(ptr, value, bytes, as_stream(stream)));
    return RLHIP_OK;
}

int32_t rlhip_memcpy_h2d(void* dst, const void* src_host, size_t bytes, rlhip_stream_t stream) {
    RLHIP_CHECK_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    RLHIP_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    return RLHIP_OK;
}

int32_t rlhip_memcpy_d2h(void* dst_host, const void* src, size_t bytes, rlhip_stream_t stream) {
    RLHIP_CHECK_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    RLHIP_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    return RLHIP_OK;
}

int32_t rlhip_memcpy_d2d(void* dst, const void* src, size_t bytes, rlhip_stream_t stream) {
    RLHIP_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return RLHIP_OK;
}

int32_t rlhip_stream_create(rlhip_stream_t* stream_out) {
    RLHIP_REQUIRE(stream_out != nullptr, "stream_out is NULL");
    hipStream_t s;
    RLHIP_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream_out = (rlhip_stream_t)s;
    return RLHIP_OK;
}

int32_t rlhip_stream_destroy(rlhip_stream_t stream) {
    RLHIP_CHECK_HIP(hipStreamDestroy(as_stream(stream)));
    return RLHIP_OK;
}

int32_t rlhip_stream_sync(rlhip_stream_t stream) {
    RLHIP_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    return RLHIP_OK;
}

int32_t rlhip_event_create(rlhip_event_t* event_out) {
    RLHIP_REQUIRE(event_out != nullptr, "event_out is NULL");
    hipEvent_t e;
    RLHIP_CHECK_HIP(hipEventCreate(&e));
    *event_out = (rlhip_event_t)e;
    return RLHIP_OK;
}

int32_t rlhip_event_destroy(rlhip_event_t event) {
    RLHIP_CHECK_HIP(hipEventDestroy((hipEvent_t)event));
    return RLHIP_OK;
}

int32_t rlhip_event_record(rlhip_event_t event, rlhip_stream_t stream) {
    RLHIP_CHECK_HIP(hipEventRecord((hipEvent_t)event, as_stream(stream)));
    return RLHIP_OK;
}

int32_t rlhip_event_elapsed_ms(rlhip_event_t start, rlhip_event_t stop, float* ms_out) {
    RLHIP_REQUIRE(ms_out != nullptr, "ms_out is NULL");
    RLHIP_CHECK_HIP(hipEventSynchronize((hipEvent_t)stop));
    RLHIP_CHECK_HIP(hipEventElapsedTime(ms_out, (hipEvent_t)start, (hipEvent_t)stop));
    return RLHIP_OK;
}

int32_t rlhip_fill_uniform_f32(float* out, int64_t n, uint64_t seed, uint32_t t, uint32_t tag,
                               rlhip_stream_t stream) {
    RLHIP_REQUIRE(out != nullptr && n >= 0, "bad output");
    if (n == 0) return RLHIP_OK;
    RLHIP_REQUIRE(((uintptr_t)out & 15) == 0, "out must be 16-byte aligned");
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0,
                       as_stream(stream), out, n, seed, t, tag);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_permutation(uint32_t* out, uint32_t n, uint64_t seed, uint32_t epoch,
                          rlhip_stream_t stream) {
    RLHIP_REQUIRE(out != nullptr, "out is NULL");
    if (n == 0) return RLHIP_OK;
    hipLaunchKernelGGL(permutation_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream),
                       out, n, seed, epoch);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

}  // extern "C"
