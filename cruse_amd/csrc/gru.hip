// Persistent grouped-GRU recurrence for gfx950 (nn.GRU forward / backward at
// model/cruse_net.py:23-31,44,50; gate order r,z,n; n = tanh(gi_n + r*(W_hn h + b_hn));
// h' = (1-z)*n + z*h; h0 = 0).
//
// The T = 401 steps are strictly sequential, so step latency -- not FLOPs or bytes --
// bounds this kernel.  Design:
//   * a CHAIN = (batch group of Bg <= 16 clips) x (GRU group); chains are independent.
//   * a chain is served by a TEAM of P = Hg/32 workgroups; workgroup p owns hidden units
//     [32p, 32p+32) and keeps its slice of W_hh (96 x Hg forward, Hg... x 32 backward)
//     resident in REGISTERS as MFMA A-operand fragments for the whole sequence.
//   * per step a workgroup (1) polls the team's P flag words, (2) gathers h_{t-1}
//     [Bg, Hg] (the previous step's output rows of the result tensor itself -- every
//     step has its own address, so there is no buffer reuse hazard) into LDS,
//     (3) runs 16x16x32 MFMA tiles (K split over the 4 wavefronts, reduced through
//     LDS), (4) applies the gate math for its 32 units, (5) publishes h_t.
//   * inter-workgroup hand-off follows the guide's R1 recipe: payload written with
//     agent-scope relaxed atomic (sc1, write-through) 8-byte stores, every wave drains
//     vmcnt(0), barrier, ONE lane stores the flag; consumers poll the flag words with
//     relaxed agent-scope loads and read the payload with agent-scope (sc1) loads.
//     Nothing depends on dispatch order or workgroup->XCD placement; block ids are
//     arranged so that, with the observed id%8 placement, a chain sits on one XCD.
//   * all workgroups of a launch must be co-resident: grid <= number of CUs
//     (one 256-thread workgroup per CU); larger batches are split into launches.
#include "common.h"

namespace {

constexpr int U = 32;                 // hidden units per workgroup
constexpr int MAXG = 8;
constexpr unsigned SPIN_LIMIT = 1u << 22;

struct GruPtrs { const float* w_hh[MAXG]; const float* b_hh[MAXG]; };

struct GruArgs {
    // forward
    const float* gi; float* h; float* r; float* z; float* n; float* ghn;
    // backward
    const float* dout; const float* hs; const float* rs; const float* zs; const float* ns; const float* ghns;
    float* dgi; float* dgh;
    GruPtrs p;
    int B, T, G, Hg, Bg, nchains, P, bg_off;
    unsigned* flags; unsigned* status;
};

__device__ __forceinline__ void st_agent_f2(float* p, float a, float b) {
    const unsigned long long v = ((unsigned long long)__float_as_uint(b) << 32) | (unsigned long long)__float_as_uint(a);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld_agent_f2(const float* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}

// wait until all P flag words of the team are >= need (one wave polls, relaxed, bounded)
__device__ __forceinline__ void team_wait(unsigned* flags, int P, unsigned need, unsigned* status, int tid) {
    if (tid < 64) {
        unsigned spins = 0;
        for (;;) {
            unsigned v = 0xffffffffu;
            if (tid < P) v = __hip_atomic_load(flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(v >= need)) break;
            if (++spins >= SPIN_LIMIT) {
                if (tid == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

// publish: every wave drains its stores, barrier, one lane sets the flag (R1)
__device__ __forceinline__ void team_publish(unsigned* flag, unsigned epoch, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// gather a [Bg][ncol] f32 panel (rows bl -> src + bl*row_stride) into LDS [16][ld]
__device__ __forceinline__ void gather_panel(float* lds, int ld, const float* src, long long row_stride, int nb,
                                             int ncol, int tid) {
    const int half = ncol >> 1;
    for (int e = tid; e < nb * half; e += 256) {
        const int bl = e / half, k2 = e - bl * half;
        const float2 v = ld_agent_f2(src + (long long)bl * row_stride + 2 * k2);
        *reinterpret_cast<float2*>(lds + bl * ld + 2 * k2) = v;
    }
}

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
template <int PREC, int NKW>
__global__ __launch_bounds__(256) void gru_fwd_kernel(GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Hg = a.Hg, KS = Hg >> 5, LD = Hg + 4, H = a.G * Hg;
    float* hB = smem;                    // [16][LD]  B operand (h_{t-1}), rows >= nb stay zero
    float* red = smem + 16 * LD;         // [4 waves][6 tiles][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int chain = blockIdx.x % a.nchains, part = blockIdx.x / a.nchains;
    const int grp = chain % a.G, bgi = a.bg_off + chain / a.G;
    const int b0 = bgi * a.Bg, nb = min(a.Bg, a.B - b0);
    const int u0 = part * U;
    const float* W = a.p.w_hh[grp];
    const float* bh = a.p.b_hh[grp];
    unsigned* flags = a.flags + (size_t)chain * a.P;

    for (int i = tid; i < 16 * LD; i += 256) hB[i] = 0.f;

    // resident weight fragments: tile j = gate*2 + half; this wave's k-steps ks = wv + 4*i
    Frag<PREC> wf[6][NKW];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int row = (j >> 1) * Hg + u0 + (j & 1) * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < NKW; ++i) {
            const int ks = wv + 4 * i;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[e] = ks < KS ? W[(long long)row * Hg + ks * 32 + (lane >> 4) * 8 + e] : 0.f;
            wf[j][i].set(v);
        }
    }

    // items: (unit pair up, local batch bl); item id it = tid + 256*q
    const int up = tid & 15;
    const int u = 2 * up;                                   // even unit 0..30
    const int half = u >> 4, ru = u & 15;
    float bias[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g) { bias[g][0] = bh[g * Hg + u0 + u]; bias[g][1] = bh[g * Hg + u0 + u + 1]; }
    float hprev[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    __syncthreads();

    for (int t = 0; t < a.T; ++t) {
        float2 gir[2], giz[2], gin[2];
#pragma unroll
        for (int q = 0; q < 1; ++q) {
            const int bl = (tid >> 4) + 16 * q;
            if (bl < nb) {
                const float* gp = a.gi + (((long long)(b0 + bl) * a.T + t) * a.G + grp) * 3 * Hg + u0 + u;
                gir[q] = *reinterpret_cast<const float2*>(gp);
                giz[q] = *reinterpret_cast<const float2*>(gp + Hg);
                gin[q] = *reinterpret_cast<const float2*>(gp + 2 * Hg);
            }
        }
        if (t > 0) {
            team_wait(flags, a.P, (unsigned)t, a.status, tid);
            gather_panel(hB, LD, a.h + ((long long)b0 * a.T + (t - 1)) * H + grp * Hg, (long long)a.T * H, nb, Hg, tid);
            __syncthreads();
            f32x4 acc[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NKW; ++i) {
                const int ks = wv + 4 * i;
                if (ks < KS) {
                    const float* pb = hB + (lane & 15) * LD + ks * 32 + (lane >> 4) * 8;
                    const float4 b0v = *reinterpret_cast<const float4*>(pb);
                    const float4 b1v = *reinterpret_cast<const float4*>(pb + 4);
                    const float bv[8] = {b0v.x, b0v.y, b0v.z, b0v.w, b1v.x, b1v.y, b1v.z, b1v.w};
                    Frag<PREC> fb;
                    fb.set(bv);
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[j] = mma(wf[j][i], fb, acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 6; ++j)
                *reinterpret_cast<f32x4*>(red + ((wv * 6 + j) * 64 + lane) * 4) = acc[j];
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 1; ++q) {
            const int bl = (tid >> 4) + 16 * q;
            if (bl < nb) {
                float gh[3][2];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    float s0 = bias[g][0], s1 = bias[g][1];
                    if (t > 0) {
                        const int lp = (ru >> 2) * 16 + bl;
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const float2 pr = *reinterpret_cast<const float2*>(
                                red + ((w * 6 + g * 2 + half) * 64 + lp) * 4 + (ru & 3));
                            s0 += pr.x; s1 += pr.y;
                        }
                    }
                    gh[g][0] = s0; gh[g][1] = s1;
                }
                const float r0 = sigmoid_acc(gir[q].x + gh[0][0]), r1 = sigmoid_acc(gir[q].y + gh[0][1]);
                const float z0 = sigmoid_acc(giz[q].x + gh[1][0]), z1 = sigmoid_acc(giz[q].y + gh[1][1]);
                const float n0 = tanhf(gin[q].x + r0 * gh[2][0]), n1 = tanhf(gin[q].y + r1 * gh[2][1]);
                const float h0 = (1.f - z0) * n0 + z0 * hprev[q][0];
                const float h1 = (1.f - z1) * n1 + z1 * hprev[q][1];
                hprev[q][0] = h0; hprev[q][1] = h1;
                const long long o = ((long long)(b0 + bl) * a.T + t) * H + grp * Hg + u0 + u;
                st_agent_f2(a.h + o, h0, h1);
                if (a.r) {
                    *reinterpret_cast<float2*>(a.r + o) = make_float2(r0, r1);
                    *reinterpret_cast<float2*>(a.z + o) = make_float2(z0, z1);
                    *reinterpret_cast<float2*>(a.n + o) = make_float2(n0, n1);
                    *reinterpret_cast<float2*>(a.ghn + o) = make_float2(gh[2][0], gh[2][1]);
                }
            }
        }
        team_publish(flags + part, (unsigned)(t + 1), tid);
    }
}

// ---------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------
template <int PREC, int NKW>
__global__ __launch_bounds__(256) void gru_bwd_kernel(GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Hg = a.Hg, K = 3 * Hg, KS = K >> 5, LD = K + 4, H = a.G * Hg;
    float* dB = smem;                    // [16][LD]  B operand (dgh_{t+1})
    float* red = smem + 16 * LD;         // [4 waves][2 tiles][64][4]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int chain = blockIdx.x % a.nchains, part = blockIdx.x / a.nchains;
    const int grp = chain % a.G, bgi = a.bg_off + chain / a.G;
    const int b0 = bgi * a.Bg, nb = min(a.Bg, a.B - b0);
    const int u0 = part * U;
    const float* W = a.p.w_hh[grp];
    unsigned* flags = a.flags + (size_t)chain * a.P;

    for (int i = tid; i < 16 * LD; i += 256) dB[i] = 0.f;

    // A operand = W_hh^T slice: A[row = unit][k = gate row j] = W_hh[j][unit]
    Frag<PREC> wf[2][NKW];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = u0 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < NKW; ++i) {
            const int ks = wv + 4 * i;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[e] = ks < KS ? W[(long long)(ks * 32 + (lane >> 4) * 8 + e) * Hg + col] : 0.f;
            wf[j][i].set(v);
        }
    }

    const int up = tid & 15;
    const int u = 2 * up;
    const int half = u >> 4, ru = u & 15;
    float carry[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    __syncthreads();

    for (int t = a.T - 1; t >= 0; --t) {
        float2 rr[2], zz[2], nn[2], gg[2], hp[2], dd[2];
#pragma unroll
        for (int q = 0; q < 1; ++q) {
            const int bl = (tid >> 4) + 16 * q;
            if (bl < nb) {
                const long long o = ((long long)(b0 + bl) * a.T + t) * H + grp * Hg + u0 + u;
                rr[q] = *reinterpret_cast<const float2*>(a.rs + o);
                zz[q] = *reinterpret_cast<const float2*>(a.zs + o);
                nn[q] = *reinterpret_cast<const float2*>(a.ns + o);
                gg[q] = *reinterpret_cast<const float2*>(a.ghns + o);
                dd[q] = *reinterpret_cast<const float2*>(a.dout + o);
                hp[q] = t > 0 ? *reinterpret_cast<const float2*>(a.hs + o - H) : make_float2(0.f, 0.f);
            }
        }
        const bool have_next = t < a.T - 1;
        if (have_next) {
            team_wait(flags, a.P, (unsigned)(a.T - 1 - t), a.status, tid);
            gather_panel(dB, LD, a.dgh + (((long long)b0 * a.T + (t + 1)) * a.G + grp) * K, (long long)a.T * a.G * K,
                         nb, K, tid);
            __syncthreads();
            f32x4 acc[2];
            acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NKW; ++i) {
                const int ks = wv + 4 * i;
                if (ks < KS) {
                    const float* pb = dB + (lane & 15) * LD + ks * 32 + (lane >> 4) * 8;
                    const float4 b0v = *reinterpret_cast<const float4*>(pb);
                    const float4 b1v = *reinterpret_cast<const float4*>(pb + 4);
                    const float bv[8] = {b0v.x, b0v.y, b0v.z, b0v.w, b1v.x, b1v.y, b1v.z, b1v.w};
                    Frag<PREC> fb;
                    fb.set(bv);
                    acc[0] = mma(wf[0][i], fb, acc[0]);
                    acc[1] = mma(wf[1][i], fb, acc[1]);
                }
            }
            *reinterpret_cast<f32x4*>(red + ((wv * 2 + 0) * 64 + lane) * 4) = acc[0];
            *reinterpret_cast<f32x4*>(red + ((wv * 2 + 1) * 64 + lane) * 4) = acc[1];
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 1; ++q) {
            const int bl = (tid >> 4) + 16 * q;
            if (bl < nb) {
                float mm0 = 0.f, mm1 = 0.f;
                if (have_next) {
                    const int lp = (ru >> 2) * 16 + bl;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float2 pr = *reinterpret_cast<const float2*>(red + ((w * 2 + half) * 64 + lp) * 4 + (ru & 3));
                        mm0 += pr.x; mm1 += pr.y;
                    }
                }
                const float dh0 = dd[q].x + carry[q][0] + mm0, dh1 = dd[q].y + carry[q][1] + mm1;
                const float r0 = rr[q].x, r1 = rr[q].y, z0 = zz[q].x, z1 = zz[q].y, n0 = nn[q].x, n1 = nn[q].y;
                const float dn0 = dh0 * (1.f - z0), dn1 = dh1 * (1.f - z1);
                const float dz0 = dh0 * (hp[q].x - n0), dz1 = dh1 * (hp[q].y - n1);
                carry[q][0] = dh0 * z0; carry[q][1] = dh1 * z1;
                const float dnp0 = dn0 * (1.f - n0 * n0), dnp1 = dn1 * (1.f - n1 * n1);
                const float dzp0 = dz0 * z0 * (1.f - z0), dzp1 = dz1 * z1 * (1.f - z1);
                const float drp0 = dnp0 * gg[q].x * r0 * (1.f - r0), drp1 = dnp1 * gg[q].y * r1 * (1.f - r1);
                const long long o3 = (((long long)(b0 + bl) * a.T + t) * a.G + grp) * K + u0 + u;
                *reinterpret_cast<float2*>(a.dgi + o3) = make_float2(drp0, drp1);
                *reinterpret_cast<float2*>(a.dgi + o3 + Hg) = make_float2(dzp0, dzp1);
                *reinterpret_cast<float2*>(a.dgi + o3 + 2 * Hg) = make_float2(dnp0, dnp1);
                st_agent_f2(a.dgh + o3, drp0, drp1);
                st_agent_f2(a.dgh + o3 + Hg, dzp0, dzp1);
                st_agent_f2(a.dgh + o3 + 2 * Hg, dnp0 * r0, dnp1 * r1);
            }
        }
        team_publish(flags + part, (unsigned)(a.T - t), tid);
    }
}

int num_cus() {
    static int cached = 0;
    if (cached == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached = n;
    }
    return cached;
}

struct Plan { int Bg, P, nbg, chains_per_launch, nlaunch; };

int make_plan(int B, int G, int Hg, Plan& pl) {
    pl.P = Hg / U;
    const int maxblk = num_cus();
    if (G * pl.P > maxblk) return -1;
    pl.Bg = 8;
    pl.nbg = cdiv(B, pl.Bg);
    if (pl.nbg * G * pl.P > maxblk) { pl.Bg = 16; pl.nbg = cdiv(B, pl.Bg); }
    int bg_per_launch = maxblk / (G * pl.P);
    if (bg_per_launch > pl.nbg) bg_per_launch = pl.nbg;
    pl.chains_per_launch = bg_per_launch * G;
    pl.nlaunch = cdiv(pl.nbg, bg_per_launch);
    return 0;
}

template <typename Kern>
int launch_one(Kern k, const GruArgs& a, int grid, size_t lds, hipStream_t s, const char* name) {
    int rc0 = cruse_ensure_dyn_lds(reinterpret_cast<const void*>(k), lds, name);
    if (rc0) return rc0;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, a);
    CRUSE_LAUNCH_CHECK(name);
    return CRUSE_OK;
}

template <int PREC>
int dispatch_fwd(const GruArgs& a, int grid, size_t lds, hipStream_t s) {
    const int nkw = cdiv(a.Hg / 32, 4);
    if (nkw <= 2) return launch_one(gru_fwd_kernel<PREC, 2>, a, grid, lds, s, "gru_seq_fwd");
    if (nkw <= 3) return launch_one(gru_fwd_kernel<PREC, 3>, a, grid, lds, s, "gru_seq_fwd");
    if (nkw <= 5) return launch_one(gru_fwd_kernel<PREC, 5>, a, grid, lds, s, "gru_seq_fwd");
    return launch_one(gru_fwd_kernel<PREC, 8>, a, grid, lds, s, "gru_seq_fwd");
}
template <int PREC>
int dispatch_bwd(const GruArgs& a, int grid, size_t lds, hipStream_t s) {
    const int nkw = cdiv(3 * a.Hg / 32, 4);
    if (nkw <= 4) return launch_one(gru_bwd_kernel<PREC, 4>, a, grid, lds, s, "gru_seq_bwd");
    if (nkw <= 8) return launch_one(gru_bwd_kernel<PREC, 8>, a, grid, lds, s, "gru_seq_bwd");
    if (nkw <= 15) return launch_one(gru_bwd_kernel<PREC, 15>, a, grid, lds, s, "gru_seq_bwd");
    return launch_one(gru_bwd_kernel<PREC, 24>, a, grid, lds, s, "gru_seq_bwd");
}

int check_common(int B, int T, int G, int Hg, int prec, const char* name) {
    CRUSE_REQUIRE(B > 0 && T > 0 && G > 0 && G <= MAXG, CRUSE_E_SHAPE, "%s: bad shape B=%d T=%d G=%d", name, B, T, G);
    CRUSE_REQUIRE(Hg % 32 == 0 && Hg >= 32 && Hg <= 1024, CRUSE_E_SHAPE,
                  "%s: hidden size per group %d must be a multiple of 32 in [32,1024]", name, Hg);
    CRUSE_REQUIRE(prec == CRUSE_PREC_F32 || prec == CRUSE_PREC_BF16X3 || prec == CRUSE_PREC_BF16, CRUSE_E_DTYPE,
                  "%s: unknown precision %d", name, prec);
    return CRUSE_OK;
}

}  // namespace

extern "C" size_t cruse_gru_ws_bytes(int B, int G, int Hg) {
    // status word + one flag word per (chain, part); chains <= ceil(B/8)*G
    const size_t chains = (size_t)cdiv(B, 8) * G;
    return 256 + chains * (size_t)(Hg / U) * sizeof(unsigned);
}

extern "C" int cruse_gru_seq_fwd(const float* gi, const float* const* w_hh, const float* const* b_hh,
                                 float* h, float* r, float* z, float* n, float* ghn,
                                 int B, int T, int G, int Hg, int prec, void* ws, void* stream) {
    int rc = check_common(B, T, G, Hg, prec, "gru_seq_fwd");
    if (rc) return rc;
    CRUSE_REQUIRE((r == nullptr) == (z == nullptr) && (r == nullptr) == (n == nullptr) && (r == nullptr) == (ghn == nullptr),
                  CRUSE_E_SHAPE, "gru_seq_fwd: r, z, n, ghn must all be given or all be NULL");
    Plan pl;
    CRUSE_REQUIRE(make_plan(B, G, Hg, pl) == 0, CRUSE_E_SHAPE, "gru_seq_fwd: G*Hg/32 exceeds the CU count");
    hipStream_t s = (hipStream_t)stream;
    CRUSE_HIP(hipMemsetAsync(ws, 0, cruse_gru_ws_bytes(B, G, Hg), s), "gru_seq_fwd memset");
    GruArgs a = {};
    a.gi = gi; a.h = h; a.r = r; a.z = z; a.n = n; a.ghn = ghn;
    for (int g = 0; g < G; ++g) { a.p.w_hh[g] = w_hh[g]; a.p.b_hh[g] = b_hh[g]; }
    a.B = B; a.T = T; a.G = G; a.Hg = Hg; a.Bg = pl.Bg; a.P = pl.P;
    a.status = (unsigned*)ws;
    const size_t lds = ((size_t)16 * (Hg + 4) + 4 * 6 * 64 * 4) * sizeof(float);
    for (int L = 0; L < pl.nlaunch; ++L) {
        const int bg_off = L * (pl.chains_per_launch / G);
        const int nbg_here = (pl.nbg - bg_off) < (pl.chains_per_launch / G) ? (pl.nbg - bg_off) : (pl.chains_per_launch / G);
        a.bg_off = bg_off;
        a.nchains = nbg_here * G;
        a.flags = (unsigned*)((char*)ws + 256) + (size_t)bg_off * G * pl.P;
        const int grid = a.nchains * pl.P;
        if (prec == CRUSE_PREC_F32) rc = dispatch_fwd<CRUSE_PREC_F32>(a, grid, lds, s);
        else if (prec == CRUSE_PREC_BF16X3) rc = dispatch_fwd<CRUSE_PREC_BF16X3>(a, grid, lds, s);
        else rc = dispatch_fwd<CRUSE_PREC_BF16>(a, grid, lds, s);
        if (rc) return rc;
    }
    return CRUSE_OK;
}

extern "C" int cruse_gru_seq_bwd(const float* dout, const float* const* w_hh,
                                 const float* h, const float* r, const float* z, const float* n, const float* ghn,
                                 float* dgi, float* dgh,
                                 int B, int T, int G, int Hg, int prec, void* ws, void* stream) {
    int rc = check_common(B, T, G, Hg, prec, "gru_seq_bwd");
    if (rc) return rc;
    Plan pl;
    CRUSE_REQUIRE(make_plan(B, G, Hg, pl) == 0, CRUSE_E_SHAPE, "gru_seq_bwd: G*Hg/32 exceeds the CU count");
    hipStream_t s = (hipStream_t)stream;
    CRUSE_HIP(hipMemsetAsync(ws, 0, cruse_gru_ws_bytes(B, G, Hg), s), "gru_seq_bwd memset");
    GruArgs a = {};
    a.dout = dout; a.hs = h; a.rs = r; a.zs = z; a.ns = n; a.ghns = ghn; a.dgi = dgi; a.dgh = dgh;
    for (int g = 0; g < G; ++g) { a.p.w_hh[g] = w_hh[g]; a.p.b_hh[g] = nullptr; }
    a.B = B; a.T = T; a.G = G; a.Hg = Hg; a.Bg = pl.Bg; a.P = pl.P;
    a.status = (unsigned*)ws;
    const size_t lds = ((size_t)16 * (3 * Hg + 4) + 4 * 2 * 64 * 4) * sizeof(float);
    CRUSE_REQUIRE(lds <= 160 * 1024, CRUSE_E_SHAPE, "gru_seq_bwd: Hg=%d needs %zu B of LDS", Hg, lds);
    for (int L = 0; L < pl.nlaunch; ++L) {
        const int bg_off = L * (pl.chains_per_launch / G);
        const int nbg_here = (pl.nbg - bg_off) < (pl.chains_per_launch / G) ? (pl.nbg - bg_off) : (pl.chains_per_launch / G);
        a.bg_off = bg_off;
        a.nchains = nbg_here * G;
        a.flags = (unsigned*)((char*)ws + 256) + (size_t)bg_off * G * pl.P;
        const int grid = a.nchains * pl.P;
        if (prec == CRUSE_PREC_F32) rc = dispatch_bwd<CRUSE_PREC_F32>(a, grid, lds, s);
        else if (prec == CRUSE_PREC_BF16X3) rc = dispatch_bwd<CRUSE_PREC_BF16X3>(a, grid, lds, s);
        else rc = dispatch_bwd<CRUSE_PREC_BF16>(a, grid, lds, s);
        if (rc) return rc;
    }
    return CRUSE_OK;
}
