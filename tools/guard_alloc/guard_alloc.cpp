// Pluggable torch allocator for out-of-bounds hunts (tools/engine_guard_run.py): every tensor gets its OWN hipMalloc'ed block and sits at the very
// END of it (16-byte aligned), so a kernel that reads or writes past a tensor runs off the mapping and the process dies with a memory access fault
// instead of silently touching a neighbouring tensor of the caching allocator's pool.  Test tooling; nothing in cruse_amd/ uses it.
//   hipcc -shared -fPIC -o libguard_alloc.so guard_alloc.cpp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <unordered_map>

static std::mutex g_mu;
static std::unordered_map<void*, void*> g_base;       // tensor pointer -> block base
static const size_t GRAN = 2u << 20;                  // hipMalloc maps whole 2 MiB fragments: end-align inside a multiple of it

extern "C" void* guard_malloc(ssize_t size, int device, hipStream_t stream) {
    (void)stream;
    if (size <= 0) return nullptr;
    (void)hipSetDevice(device);
    const size_t need = ((size_t)size + 15) & ~(size_t)15;
    const size_t total = (need + GRAN - 1) / GRAN * GRAN;
    void* base = nullptr;
    if (hipMalloc(&base, total) != hipSuccess) return nullptr;
    void* p = (char*)base + (total - need);
    std::lock_guard<std::mutex> lk(g_mu);
    g_base[p] = base;
    return p;
}

extern "C" void guard_free(void* p, ssize_t size, int device, hipStream_t stream) {
    (void)size; (void)device; (void)stream;
    if (!p) return;
    void* base = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_base.find(p);
        if (it == g_base.end()) return;
        base = it->second;
        g_base.erase(it);
    }
    (void)hipDeviceSynchronize();
    (void)hipFree(base);
}
