// Hand-off latency between two workgroups through L2 with the recurrences' granule protocol (profiling aid, not part of the
// library): workgroup A publishes NST x (64 lanes x 16 B) tagged granules per wave and waits for B's 8-byte-per-lane reply;
// B sweeps A's granules with 16-byte sc1 loads until every tag carries the iteration number, then replies.  A's s_memtime
// difference over many iterations / 2 ~ one hop.  build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o handoff_probe.so
#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct Args { unsigned* buf; unsigned long long* out; int iters, nst, other, plain; };

template <int NW, int NST>
__global__ __launch_bounds__(NW * 64) void handoff_kernel(Args a) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool isA = blockIdx.x == 0, isB = (int)blockIdx.x == a.other;
    if (!isA && !isB) return;
    const unsigned big_bytes = (unsigned)(NW * NST) * 1024u;          // A -> B payload
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.buf, 0, 2 * (big_bytes + 4096) , 0x00020000);
    const unsigned rep_off = 2 * big_bytes;                               // B -> A reply: one 8-byte granule per lane of wave 0
    unsigned long long t0 = 0, t1 = 0;
    if (isA) {
        if (tid == 0) t0 = __builtin_amdgcn_s_memtime();
        for (int it = 1; it <= a.iters; ++it) {
            const unsigned par = (unsigned)(it & 1) * big_bytes;
#pragma unroll
            for (int s = 0; s < NST; ++s) {
                const u32x4 w = {(unsigned)it, 0x3f803f80u, (unsigned)it, 0x3f803f80u};
                const unsigned off = par + (unsigned)((wv * NST + s) * 1024 + lane * 16);
                if (a.plain) __builtin_amdgcn_raw_buffer_store_b128(w, rs, off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(w, rs, off, 0, 16);
            }
            if (wv == 0) {                                                // wait for the reply
                for (;;) {
                    const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rs, rep_off + (unsigned)(it & 1) * 512u + lane * 8, 0, 16);
                    if (__all(r.x == (unsigned)it)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
        }
        if (tid == 0) { t1 = __builtin_amdgcn_s_memtime(); a.out[0] = t1 - t0; }
    } else {
        unsigned long long polls = 0;
        for (int it = 1; it <= a.iters; ++it) {
            const unsigned par = (unsigned)(it & 1) * big_bytes;
            for (;;) {
                bool ok = true;
                u32x4 g[NST];
#pragma unroll
                for (int s = 0; s < NST; ++s) g[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, par + (unsigned)((wv * NST + s) * 1024 + lane * 16), 0, 16);
#pragma unroll
                for (int s = 0; s < NST; ++s) ok = ok & (g[s].x == (unsigned)it) & (g[s].z == (unsigned)it);
                ++polls;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
            if (wv == 0) {
                const u32x2 w = {(unsigned)it, 0u};
                const unsigned off = rep_off + (unsigned)(it & 1) * 512u + lane * 8;
                if (a.plain) __builtin_amdgcn_raw_buffer_store_b64(w, rs, off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b64(w, rs, off, 0, 16);
            }
        }
        if (tid == 0) a.out[1] = polls;
    }
}

extern "C" int handoff_run(void* buf, void* out, int iters, int nst, int nw, int other, int plain, void* stream) {
    Args a = {(unsigned*)buf, (unsigned long long*)out, iters, nst, other, plain};
    if (nw == 1 && nst == 1) hipLaunchKernelGGL((handoff_kernel<1, 1>), dim3(16), dim3(64), 0, (hipStream_t)stream, a);
    else if (nw == 1) hipLaunchKernelGGL((handoff_kernel<1, 5>), dim3(16), dim3(64), 0, (hipStream_t)stream, a);
    else if (nst == 1) hipLaunchKernelGGL((handoff_kernel<4, 1>), dim3(16), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((handoff_kernel<4, 5>), dim3(16), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
