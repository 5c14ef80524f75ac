// Persistent grouped-GRU recurrence for gfx950 (nn.GRU forward / backward at
// model/cruse_net.py:23-31,44,50; gate order r,z,n; n = tanh(gi_n + r*(W_hn h + b_hn));
// h' = (1-z)*n + z*h; h0 = 0).
//
// The T = 401 steps are strictly sequential, so step latency -- not FLOPs or bytes --
// bounds this kernel.  Design:
//   * a CHAIN = (batch group of Bg <= 16 clips) x (GRU group); chains are independent.
//   * a chain is served by a TEAM of P = Hg/32 workgroups; workgroup p owns hidden units
//     [32p, 32p+32) and keeps its slice of W_hh (96 x Hg forward, its transpose 32 x 3Hg
//     backward) resident in REGISTERS as MFMA A-operand fragments for the whole sequence.
//   * hand-off between the workgroups of a team uses DATA-TAGGED GRANULES (the guide's R2
//     form): each exchanged value travels as an 8-byte {epoch tag, f32 value} pair written
//     with a write-through (sc1) 16-byte buffer store (two granules); consumers sweep the
//     team's panel with 16-byte sc1 buffer loads until every tag equals the step's epoch.
//     There is no separate flag, no release fence and no store drain on the critical path;
//     8-byte halves are single-copy atomic.  The panel is double-buffered by step parity: a
//     workgroup can only write epoch e+2 after every team mate published e+1, which implies
//     all of them finished reading e.  Nothing depends on dispatch order or workgroup->XCD
//     placement; block ids are arranged so that, with the observed id%8 placement, a chain
//     sits on one XCD (its panel then stays in that XCD's L2 / the Infinity Cache).
//   * forward step: sweep h_{t-1} [Bg,Hg] into LDS -> 16x16x32 MFMA tiles (K split over the
//     4 wavefronts, reduced through LDS) -> gate math for the 32 own units -> publish h_t.
//   * backward: with c = d(gh)/d(dh) coefficients saved by the forward pass
//     (dgh_t = dh_t * c_t elementwise), only dh_t [Bg,Hg] is exchanged per step -- the same
//     volume as forward -- and dh_{t-1} = dout_{t-1} + z_t*dh_t + (dh_t*c_t) W_hh is the same
//     MFMA shape.  dgi / dgh for the weight-gradient GEMMs are formed afterwards by an
//     elementwise kernel from dh (cruse_gru_gate_grads).
//   * all workgroups of a launch must be co-resident: grid <= number of CUs
//     (one 256-thread workgroup per CU); larger batches are split into launches.
#include "common.h"

namespace {

constexpr int U = 32;                 // hidden units per workgroup
constexpr int MAXG = 8;
constexpr int CH = 10;                // 16-byte granule pairs in flight per thread and sweep batch
constexpr unsigned SPIN_LIMIT = 1u << 21;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct GruPtrs { const float* w_hh[MAXG]; const float* b_hh[MAXG]; };

struct GruArgs {
    // forward
    const float* gi; float* h; float* coef; float* an; float* z;
    // backward
    const float* dout; const float* coefs; const float* zs; float* dh;
    GruPtrs p;
    int B, T, G, Hg, Bg, nchains, P, bg_off;
    unsigned long long* xg;           // granule panels [chain][parity][Bg][Hg]
    unsigned xg_bytes;
    unsigned* status;
};

// Sweep a team panel of granule pairs into LDS until every tag == epoch.
//   pair e = tid + 256*i covers values (bl, v), (bl, v+1) with e = bl*(Hg/2) + v/2.
// BWD == false: lds[bl*ld + v] = value                       (h_{t-1} panel, ld = Hg+4)
// BWD == true : lds[bl*ld + g*Hg + v] = value * coef[g][v]   (dh_t * c_t, ld = 3Hg+4), coef rows
//               prefetched from `cf` (+ bl*cf_row_stride) before the first poll.
template <bool BWD>
__device__ __forceinline__ void sweep_panel(float* lds, int ld, __amdgpu_buffer_rsrc_t rs, unsigned base_bytes,
                                            int npair, int Hg, unsigned epoch, const float* cf,
                                            long long cf_row_stride, unsigned* status, int tid) {
    const int hp = Hg >> 1;
    const int ni = (npair + 255) >> 8;
    for (int i0 = 0; i0 < ni; i0 += CH) {
        unsigned pend = 0;
        float2 c[BWD ? CH : 1][3];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int e = tid + 256 * (i0 + j);
            if (i0 + j < ni && e < npair) {
                pend |= 1u << j;
                if (BWD) {
                    const int bl = e / hp, v = 2 * (e - bl * hp);
                    const float* cp = cf + (long long)bl * cf_row_stride + v;
                    c[BWD ? j : 0][0] = *reinterpret_cast<const float2*>(cp);
                    c[BWD ? j : 0][1] = *reinterpret_cast<const float2*>(cp + Hg);
                    c[BWD ? j : 0][2] = *reinterpret_cast<const float2*>(cp + 2 * Hg);
                }
            }
        }
        unsigned spins = 0;
        for (;;) {
            u32x4 g[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (pend & (1u << j))
                    g[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, base_bytes + (unsigned)(tid + 256 * (i0 + j)) * 16u, 0, 16);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if ((pend & (1u << j)) && g[j].x == epoch && g[j].z == epoch) {
                    const int e = tid + 256 * (i0 + j);
                    const int bl = e / hp, v = 2 * (e - bl * hp);
                    const float v0 = __uint_as_float(g[j].y), v1 = __uint_as_float(g[j].w);
                    if (BWD) {
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            *reinterpret_cast<float2*>(lds + bl * ld + q * Hg + v) =
                                make_float2(v0 * c[BWD ? j : 0][q].x, v1 * c[BWD ? j : 0][q].y);
                    } else {
                        *reinterpret_cast<float2*>(lds + bl * ld + v) = make_float2(v0, v1);
                    }
                    pend &= ~(1u << j);
                }
            }
            if (__syncthreads_and(pend == 0)) break;
            if (++spins >= SPIN_LIMIT) {
                if (tid == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
}

__device__ __forceinline__ void publish_pair(__amdgpu_buffer_rsrc_t rs, unsigned off_bytes, unsigned epoch, float a, float b) {
    const u32x4 w = {epoch, __float_as_uint(a), epoch, __float_as_uint(b)};
    __builtin_amdgcn_raw_buffer_store_b128(w, rs, off_bytes, 0, 16);   // aux 16 = sc1 (write-through)
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
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.xg, 0, a.xg_bytes, 0x00020000);
    const unsigned panel_bytes = (unsigned)(a.Bg * Hg) * 8u;
    const unsigned cbase = (unsigned)chain * 2u * panel_bytes;

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

    // item of this thread: unit pair (u, u+1) of local clip bl
    const int u = 2 * (tid & 15);
    const int bl = tid >> 4;
    const bool active = bl < nb;
    const int half = u >> 4, ru = u & 15;
    const int lp = (ru >> 2) * 16 + bl;
    float bias[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g) { bias[g][0] = bh[g * Hg + u0 + u]; bias[g][1] = bh[g * Hg + u0 + u + 1]; }
    float hp0 = 0.f, hp1 = 0.f;
    __syncthreads();

    for (int t = 0; t < a.T; ++t) {
        float2 gir = make_float2(0.f, 0.f), giz = gir, gin = gir;
        if (active) {
            const float* gp = a.gi + (((long long)(b0 + bl) * a.T + t) * a.G + grp) * 3 * Hg + u0 + u;
            gir = *reinterpret_cast<const float2*>(gp);
            giz = *reinterpret_cast<const float2*>(gp + Hg);
            gin = *reinterpret_cast<const float2*>(gp + 2 * Hg);
        }
        float gh[3][2];
#pragma unroll
        for (int g = 0; g < 3; ++g) { gh[g][0] = bias[g][0]; gh[g][1] = bias[g][1]; }
        if (t > 0) {
            sweep_panel<false>(hB, LD, rs, cbase + (unsigned)((t - 1) & 1) * panel_bytes, nb * (Hg >> 1), Hg,
                               (unsigned)t, nullptr, 0, a.status, tid);
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
            if (active) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float2 pr = *reinterpret_cast<const float2*>(red + ((w * 6 + g * 2 + half) * 64 + lp) * 4 + (ru & 3));
                        gh[g][0] += pr.x; gh[g][1] += pr.y;
                    }
            }
        }
        if (active) {
            const float r0 = sigmoid_acc(gir.x + gh[0][0]), r1 = sigmoid_acc(gir.y + gh[0][1]);
            const float z0 = sigmoid_acc(giz.x + gh[1][0]), z1 = sigmoid_acc(giz.y + gh[1][1]);
            const float n0 = tanhf(gin.x + r0 * gh[2][0]), n1 = tanhf(gin.y + r1 * gh[2][1]);
            const float h0 = (1.f - z0) * n0 + z0 * hp0;
            const float h1 = (1.f - z1) * n1 + z1 * hp1;
            publish_pair(rs, cbase + (unsigned)(t & 1) * panel_bytes + (unsigned)(bl * Hg + u0 + u) * 8u,
                         (unsigned)(t + 1), h0, h1);
            const long long o = ((long long)(b0 + bl) * a.T + t) * H + grp * Hg + u0 + u;
            *reinterpret_cast<float2*>(a.h + o) = make_float2(h0, h1);
            if (a.coef) {
                // dgh = dh * (c_r, c_z, c_n); dgi_n = dh * a_n   (see header)
                const float an0 = (1.f - z0) * (1.f - n0 * n0), an1 = (1.f - z1) * (1.f - n1 * n1);
                const long long o3 = (((long long)(b0 + bl) * a.T + t) * a.G + grp) * 3 * Hg + u0 + u;
                *reinterpret_cast<float2*>(a.coef + o3) =
                    make_float2(an0 * gh[2][0] * r0 * (1.f - r0), an1 * gh[2][1] * r1 * (1.f - r1));
                *reinterpret_cast<float2*>(a.coef + o3 + Hg) =
                    make_float2((hp0 - n0) * z0 * (1.f - z0), (hp1 - n1) * z1 * (1.f - z1));
                *reinterpret_cast<float2*>(a.coef + o3 + 2 * Hg) = make_float2(an0 * r0, an1 * r1);
                *reinterpret_cast<float2*>(a.an + o) = make_float2(an0, an1);
                *reinterpret_cast<float2*>(a.z + o) = make_float2(z0, z1);
            }
            hp0 = h0; hp1 = h1;
        }
    }
}

// ---------------------------------------------------------------------------------
// backward: dh_s = dout_s + z_{s+1} * dh_{s+1} + (dh_{s+1} * c_{s+1}) W_hh
// ---------------------------------------------------------------------------------
template <int PREC, int NKW>
__global__ __launch_bounds__(256) void gru_bwd_kernel(GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Hg = a.Hg, K = 3 * Hg, KS = K >> 5, LD = K + 4, H = a.G * Hg;
    float* dB = smem;                    // [16][LD]  B operand (dh_{s+1} * c_{s+1})
    float* red = smem + 16 * LD;         // [4 waves][2 tiles][64][4]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int chain = blockIdx.x % a.nchains, part = blockIdx.x / a.nchains;
    const int grp = chain % a.G, bgi = a.bg_off + chain / a.G;
    const int b0 = bgi * a.Bg, nb = min(a.Bg, a.B - b0);
    const int u0 = part * U;
    const float* W = a.p.w_hh[grp];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.xg, 0, a.xg_bytes, 0x00020000);
    const unsigned panel_bytes = (unsigned)(a.Bg * Hg) * 8u;
    const unsigned cbase = (unsigned)chain * 2u * panel_bytes;

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

    const int u = 2 * (tid & 15);
    const int bl = tid >> 4;
    const bool active = bl < nb;
    const int half = u >> 4, ru = u & 15;
    const int lp = (ru >> 2) * 16 + bl;
    float dh0 = 0.f, dh1 = 0.f;          // dh_{s+1} of the own units
    __syncthreads();

    for (int k = 0; k < a.T; ++k) {
        const int s = a.T - 1 - k;
        float2 dd = make_float2(0.f, 0.f), zz = dd;
        const long long o = ((long long)(b0 + bl) * a.T + s) * H + grp * Hg + u0 + u;
        if (active) {
            dd = *reinterpret_cast<const float2*>(a.dout + o);
            if (k > 0) zz = *reinterpret_cast<const float2*>(a.zs + o + H);
        }
        float mm0 = 0.f, mm1 = 0.f;
        if (k > 0) {
            const float* cf = a.coefs + (((long long)b0 * a.T + (s + 1)) * a.G + grp) * K;
            sweep_panel<true>(dB, LD, rs, cbase + (unsigned)((k - 1) & 1) * panel_bytes, nb * (Hg >> 1), Hg,
                              (unsigned)k, cf, (long long)a.T * a.G * K, a.status, tid);
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
            if (active) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float2 pr = *reinterpret_cast<const float2*>(red + ((w * 2 + half) * 64 + lp) * 4 + (ru & 3));
                    mm0 += pr.x; mm1 += pr.y;
                }
            }
        }
        if (active) {
            dh0 = dd.x + zz.x * dh0 + mm0;
            dh1 = dd.y + zz.y * dh1 + mm1;
            publish_pair(rs, cbase + (unsigned)(k & 1) * panel_bytes + (unsigned)(bl * Hg + u0 + u) * 8u,
                         (unsigned)(k + 1), dh0, dh1);
            *reinterpret_cast<float2*>(a.dh + o) = make_float2(dh0, dh1);
        }
    }
}

// dgi = dh * (c_r, c_z, a_n), dgh = dh * (c_r, c_z, c_n); layouts [rows][G][3][Hg]
__global__ __launch_bounds__(256) void gru_gate_grads_kernel(const float* dh, const float* coef, const float* an,
                                                             float* dgi, float* dgh, long long rows, int G, int Hg) {
    const int H = G * Hg;
    const long long n = rows * H;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long row = i / H;
        const int c = (int)(i - row * H);
        const int g = c / Hg, j = c - g * Hg;
        const float d = dh[i];
        const long long o3 = (row * G + g) * 3 * Hg + j;
        const float cr = coef[o3], cz = coef[o3 + Hg], cn = coef[o3 + 2 * Hg];
        dgi[o3] = d * cr; dgi[o3 + Hg] = d * cz; dgi[o3 + 2 * Hg] = d * an[i];
        dgh[o3] = d * cr; dgh[o3 + Hg] = d * cz; dgh[o3 + 2 * Hg] = d * cn;
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

struct Plan { int Bg, P, nbg, bg_per_launch, nlaunch; };

int make_plan(int B, int G, int Hg, Plan& pl) {
    pl.P = Hg / U;
    const int maxblk = num_cus();
    if (G * pl.P > maxblk) return -1;
    pl.Bg = 8;
    pl.nbg = cdiv(B, pl.Bg);
    if (pl.nbg * G * pl.P > maxblk) { pl.Bg = 16; pl.nbg = cdiv(B, pl.Bg); }
    pl.bg_per_launch = maxblk / (G * pl.P);
    if (pl.bg_per_launch > pl.nbg) pl.bg_per_launch = pl.nbg;
    pl.nlaunch = cdiv(pl.nbg, pl.bg_per_launch);
    return 0;
}

size_t xg_bytes_total(int B, int G, int Hg) {
    // chains <= ceil(B/8)*G, two parities, up to 16 rows of Hg granules (8 bytes each)
    return (size_t)cdiv(B, 8) * G * 2 * 16 * Hg * 8;
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

template <bool FWD>
int run_launches(GruArgs& a, const Plan& pl, int G, int Hg, int prec, void* ws, size_t lds, hipStream_t s) {
    a.status = (unsigned*)ws;
    a.xg = (unsigned long long*)((char*)ws + 256);
    a.xg_bytes = (unsigned)xg_bytes_total(a.B, G, Hg);
    a.Bg = pl.Bg; a.P = pl.P;
    int rc = CRUSE_OK;
    for (int L = 0; L < pl.nlaunch; ++L) {
        const int bg_off = L * pl.bg_per_launch;
        const int nbg_here = (pl.nbg - bg_off) < pl.bg_per_launch ? (pl.nbg - bg_off) : pl.bg_per_launch;
        a.bg_off = bg_off;
        a.nchains = nbg_here * G;
        // every launch gets its own panel region: chain index inside the launch + offset
        a.xg = (unsigned long long*)((char*)ws + 256) + (size_t)bg_off * G * 2 * pl.Bg * Hg;
        a.xg_bytes = (unsigned)((size_t)a.nchains * 2 * pl.Bg * Hg * 8);
        const int grid = a.nchains * pl.P;
        if (FWD) {
            if (prec == CRUSE_PREC_F32) rc = dispatch_fwd<CRUSE_PREC_F32>(a, grid, lds, s);
            else if (prec == CRUSE_PREC_BF16X3) rc = dispatch_fwd<CRUSE_PREC_BF16X3>(a, grid, lds, s);
            else rc = dispatch_fwd<CRUSE_PREC_BF16>(a, grid, lds, s);
        } else {
            if (prec == CRUSE_PREC_F32) rc = dispatch_bwd<CRUSE_PREC_F32>(a, grid, lds, s);
            else if (prec == CRUSE_PREC_BF16X3) rc = dispatch_bwd<CRUSE_PREC_BF16X3>(a, grid, lds, s);
            else rc = dispatch_bwd<CRUSE_PREC_BF16>(a, grid, lds, s);
        }
        if (rc) return rc;
    }
    return rc;
}

}  // namespace

extern "C" size_t cruse_gru_ws_bytes(int B, int G, int Hg) {
    return 256 + xg_bytes_total(B, G, Hg);
}

extern "C" int cruse_gru_seq_fwd(const float* gi, const float* const* w_hh, const float* const* b_hh,
                                 float* h, float* coef, float* an, float* z,
                                 int B, int T, int G, int Hg, int prec, void* ws, void* stream) {
    int rc = check_common(B, T, G, Hg, prec, "gru_seq_fwd");
    if (rc) return rc;
    CRUSE_REQUIRE((coef == nullptr) == (an == nullptr) && (coef == nullptr) == (z == nullptr), CRUSE_E_SHAPE,
                  "gru_seq_fwd: coef, an, z must all be given or all be NULL");
    Plan pl;
    CRUSE_REQUIRE(make_plan(B, G, Hg, pl) == 0, CRUSE_E_SHAPE, "gru_seq_fwd: G*Hg/32 exceeds the CU count");
    hipStream_t s = (hipStream_t)stream;
    { int zrc = cruse_zero_async(ws, cruse_gru_ws_bytes(B, G, Hg), s, "gru_seq_fwd memset"); if (zrc) return zrc; }
    GruArgs a = {};
    a.gi = gi; a.h = h; a.coef = coef; a.an = an; a.z = z;
    for (int g = 0; g < G; ++g) { a.p.w_hh[g] = w_hh[g]; a.p.b_hh[g] = b_hh[g]; }
    a.B = B; a.T = T; a.G = G; a.Hg = Hg;
    const size_t lds = ((size_t)16 * (Hg + 4) + 4 * 6 * 64 * 4) * sizeof(float);
    return run_launches<true>(a, pl, G, Hg, prec, ws, lds, s);
}

extern "C" int cruse_gru_seq_bwd(const float* dout, const float* const* w_hh, const float* coef, const float* z,
                                 float* dh, int B, int T, int G, int Hg, int prec, void* ws, void* stream) {
    int rc = check_common(B, T, G, Hg, prec, "gru_seq_bwd");
    if (rc) return rc;
    Plan pl;
    CRUSE_REQUIRE(make_plan(B, G, Hg, pl) == 0, CRUSE_E_SHAPE, "gru_seq_bwd: G*Hg/32 exceeds the CU count");
    hipStream_t s = (hipStream_t)stream;
    { int zrc = cruse_zero_async(ws, cruse_gru_ws_bytes(B, G, Hg), s, "gru_seq_bwd memset"); if (zrc) return zrc; }
    GruArgs a = {};
    a.dout = dout; a.coefs = coef; a.zs = z; a.dh = dh;
    for (int g = 0; g < G; ++g) { a.p.w_hh[g] = w_hh[g]; a.p.b_hh[g] = nullptr; }
    a.B = B; a.T = T; a.G = G; a.Hg = Hg;
    const size_t lds = ((size_t)16 * (3 * Hg + 4) + 4 * 2 * 64 * 4) * sizeof(float);
    CRUSE_REQUIRE(lds <= 160 * 1024, CRUSE_E_SHAPE, "gru_seq_bwd: Hg=%d needs %zu B of LDS", Hg, lds);
    return run_launches<false>(a, pl, G, Hg, prec, ws, lds, s);
}

extern "C" int cruse_gru_gate_grads(const float* dh, const float* coef, const float* an, float* dgi, float* dgh,
                                    long long rows, int G, int Hg, void* stream) {
    CRUSE_REQUIRE(rows > 0 && G > 0 && Hg > 0, CRUSE_E_SHAPE, "gru_gate_grads: bad shape");
    long long nblk = (rows * G * Hg + 1023) / 1024;
    if (nblk > 4096) nblk = 4096;
    hipLaunchKernelGGL(gru_gate_grads_kernel, dim3((int)nblk), dim3(256), 0, (hipStream_t)stream, dh, coef, an, dgi, dgh,
                       rows, G, Hg);
    CRUSE_LAUNCH_CHECK("gru_gate_grads");
    return CRUSE_OK;
}
