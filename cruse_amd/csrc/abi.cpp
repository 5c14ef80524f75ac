// Error channel and version of the C ABI (include/cruse_hip.h).
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/cruse_hip.h"

static thread_local char g_err[512] = "";

extern "C" void cruse_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cruse_last_error(void) { return g_err; }
extern "C" int cruse_abi_version(void) { return CRUSE_ABI_VERSION; }

// ---- library options: explicit, set by the HOST through the C ABI (the library reads no environment variables) ---------
// name -> value; unset options take the default the call site passes to cruse_opt().
#include <atomic>
#include <limits.h>
#include <string.h>
namespace {
const char* const OPT_NAMES[] = {"gru_bwd_rs", "gru_fwd_lean", "gru_tf", "gru_poll_fwd", "gru_poll_bwd", "gru_wlo", "gru_dbg", "cm_dbg", "cm_nw", "cm_kint", "cm_swap",
                                 "pw_valu", "wg_dbg", "wg_rd", "wg_grid"};
constexpr int N_OPT = sizeof(OPT_NAMES) / sizeof(OPT_NAMES[0]);
std::atomic<int> g_opt[N_OPT];
struct OptInit { OptInit() { for (auto& o : g_opt) o.store(INT_MIN); } } g_opt_init;
int opt_index(const char* name) {
    for (int i = 0; i < N_OPT; ++i)
        if (name && strcmp(name, OPT_NAMES[i]) == 0) return i;
    return -1;
}
}  // namespace
int cruse_opt(const char* name, int dflt) {
    const int i = opt_index(name);
    if (i < 0) return dflt;
    const int v = g_opt[i].load(std::memory_order_relaxed);
    return v == INT_MIN ? dflt : v;
}
extern "C" int cruse_set_option(const char* name, int value, int unset) {
    const int i = opt_index(name);
    if (i < 0) { cruse_set_error("set_option: unknown option '%s'", name ? name : "(null)"); return CRUSE_E_SHAPE; }
    g_opt[i].store(unset ? INT_MIN : value);
    return CRUSE_OK;
}
extern "C" int cruse_get_option(const char* name, int* value, int* is_set) {
    const int i = opt_index(name);
    if (i < 0 || !value || !is_set) { cruse_set_error("get_option: unknown option '%s'", name ? name : "(null)"); return CRUSE_E_SHAPE; }
    const int v = g_opt[i].load();
    *is_set = v != INT_MIN;
    *value = v == INT_MIN ? 0 : v;
    return CRUSE_OK;
}

extern "C" void cruse_set_error(const char* fmt, ...);

int cruse_ensure_dyn_lds(const void* fn, size_t bytes, const char* name) {
    if (bytes <= 48 * 1024) return CRUSE_OK;
    static std::mutex mu;
    static std::map<const void*, size_t> done;
    std::lock_guard<std::mutex> lock(mu);
    auto it = done.find(fn);
    if (it != done.end() && it->second >= bytes) return CRUSE_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        cruse_set_error("%s: cannot set %zu B of dynamic LDS: %s", name, bytes, hipGetErrorString(e));
        return CRUSE_E_HIP;
    }
    done[fn] = bytes;
    return CRUSE_OK;
}

__global__ void cruse_zero_kernel(unsigned* p, size_t nwords) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

int cruse_zero_async(void* p, size_t bytes, hipStream_t stream, const char* name) {
    if (bytes == 0) return CRUSE_OK;
    const size_t nwords = (bytes + 3) / 4;
    size_t nblk = (nwords + 1023) / 1024;
    if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(cruse_zero_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, (unsigned*)p, nwords);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        cruse_set_error("%s: zero-fill launch failed: %s", name, hipGetErrorString(e));
        return CRUSE_E_HIP;
    }
    return CRUSE_OK;
}

// ---- CU-masked streams and a placement census (profiling / scheduling aid) -----------------------------------------
// hipExtStreamCreateWithCUMask: a stream whose kernels may only run on the CUs whose bit is set.  Used to keep the
// side-stream leaves off the CUs that hold the persistent GRU workgroups.
extern "C" int cruse_stream_create_masked(void** stream_out, const unsigned* mask, int nwords) {
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask);
    if (e != hipSuccess) {
        cruse_set_error("stream_create_masked: %s", hipGetErrorString(e));
        return CRUSE_E_HIP;
    }
    *stream_out = (void*)s;
    return CRUSE_OK;
}

// out[block] = {HW_REG_XCC_ID, HW_REG_HW_ID}: which XCD / SE / CU a block ran on; every block spins `spin` clock ticks so
// that a launch of <= 256 blocks is spread one block per CU.
__global__ void cruse_census_kernel(unsigned* out, unsigned spin) {
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);     // HW_REG_XCC_ID
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}
extern "C" int cruse_cu_census(unsigned* out, int nblocks, unsigned spin, void* stream) {
    hipLaunchKernelGGL(cruse_census_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, out, spin);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cruse_set_error("cu_census: %s", hipGetErrorString(e)); return CRUSE_E_HIP; }
    return CRUSE_OK;
}

// A workgroup that HOLDS a CU for `ticks` shader clocks: 128 KB of LDS per block, so no recurrence workgroup (45-60 KB) fits beside
// it -- what a collective's channels (RCCL) or another tenant's kernel do to the persistent recurrences, which need their teams
// co-resident.  Test / probe rig (tests/test_gpu_ddp.py, tools/host_contention_probe.py); not used by the product path.
__global__ __launch_bounds__(256) void cruse_hog_kernel(unsigned long long ticks, unsigned* sink) {
    extern __shared__ unsigned hog_lds[];
    hog_lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (sink && hog_lds[(threadIdx.x + 1) & 255] == 0xffffffffu) *sink = 1u;
}
extern "C" int cruse_cu_hog(int nblocks, unsigned long long ticks, void* stream) {
    if (nblocks <= 0 || nblocks > 256) { cruse_set_error("cu_hog: nblocks=%d (1..256)", nblocks); return CRUSE_E_SHAPE; }
    const size_t lds = 128 * 1024;
    int rc = cruse_ensure_dyn_lds(reinterpret_cast<const void*>(cruse_hog_kernel), lds, "cu_hog");
    if (rc) return rc;
    hipLaunchKernelGGL(cruse_hog_kernel, dim3(nblocks), dim3(256), lds, (hipStream_t)stream, ticks, (unsigned*)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cruse_set_error("cu_hog: %s", hipGetErrorString(e)); return CRUSE_E_HIP; }
    return CRUSE_OK;
}
