// Frame-major direct convolutions for the CRUSE encoder / decoder / skip paths.
//
// Replaces nn.Conv2d((2,3),(1,2),(1,1)) + causal crop, nn.Conv2d((1,3)) skip and
// nn.ConvTranspose2d((1,3),(1,2)) + crop of model/cruse_net.py:138-143,149-164 and
// their autograd backward.  Activations are [B,T,C,F]: one row of C*F (= 640 at every
// U-Net level) floats per frame, so every global access is a contiguous frame row.
//
// Round-1 form: f32 VALU, weights + a tile of frames staged in LDS, register tiles
// of CO_T output channels x TT frames per thread.  HBM-bound in principle
// (2 x 640 floats per frame per layer); see DESIGN.md for the roofline.
#include "common.h"

namespace {

constexpr int TF = 8;        // frames per workgroup tile
constexpr int TT = 4;        // frames per thread item
constexpr int NSET = TF / TT;
constexpr int CONV_THREADS = 320;

struct ConvArgs {
    const float* x; const float* w; const float* bias; float* y;
    int B, T, Cin, Fin, Cout, Fout, KT, S, pad, w_layout, act, accum;
    double* sums;            // gather form only: BatchNorm batch sums of y ([2][Cout]) accumulated by the epilogue, or null
};

__device__ __forceinline__ float apply_act(float v, int act) {
    return act == 1 ? sigmoid_acc(v) : v;
}

// Stage `nrows` frame rows ([Cin][Fin] floats each, frames t_first .. of clip b; zero outside the clip) into LDS rows
// [Cin][Fin + 2] with a zero column on either side.  float4 global loads, several in flight per thread before the
// first LDS store (the element-wise loop it replaces issued one 4-byte load per iteration and waited for it: 16
// serialised HBM round trips per workgroup on the 8 -> 1 decoder layer, 59 us for a kernel that moves 82 MB).
__device__ __forceinline__ void stage_rows(float* xl, const float* x, int b, int t_first, int nrows, int Cin, int Fin, int T,
                                           int tid) {
    const int FinP = Fin + 2, rowraw = Cin * Fin, rowlen = Cin * FinP;
    for (int i = tid; i < nrows * Cin; i += CONV_THREADS) {          // the pad columns
        xl[i * FinP] = 0.f;
        xl[i * FinP + Fin + 1] = 0.f;
    }
    if ((Fin & 3) == 0 && (((uintptr_t)x) & 15) == 0) {
        constexpr int SV = 4;
        const int nvec = nrows * rowraw / 4;
        for (int base = 0; base < nvec; base += CONV_THREADS * SV) {
            float4 v[SV];
            int rr[SV];
#pragma unroll
            for (int q = 0; q < SV; ++q) {
                const int i = base + tid + q * CONV_THREADS;
                v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                rr[q] = 0;
                if (i < nvec) {
                    const int r = (i * 4) / rowraw;
                    const int t = t_first + r;
                    rr[q] = r;
                    if (t >= 0 && t < T)
                        v[q] = *reinterpret_cast<const float4*>(x + ((long long)b * T + t) * rowraw + (i * 4 - r * rowraw));
                }
            }
#pragma unroll
            for (int q = 0; q < SV; ++q) {
                const int i = base + tid + q * CONV_THREADS;
                if (i < nvec) {
                    const int j = i * 4 - rr[q] * rowraw;
                    const int ci = j / Fin, f = j - ci * Fin;            // Fin % 4 == 0: the four elements share ci
                    float* d = xl + rr[q] * rowlen + ci * FinP + 1 + f;
                    d[0] = v[q].x; d[1] = v[q].y; d[2] = v[q].z; d[3] = v[q].w;
                }
            }
        }
    } else {
        for (int i = tid; i < nrows * rowraw; i += CONV_THREADS) {
            const int r = i / rowraw, j = i - r * rowraw;
            const int ci = j / Fin, f = j - ci * Fin;
            const int t = t_first + r;
            xl[r * rowlen + ci * FinP + 1 + f] = (t >= 0 && t < T) ? x[((long long)b * T + t) * rowraw + j] : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------
// gather form
// ---------------------------------------------------------------------------
template <int CO_T>
__global__ __launch_bounds__(CONV_THREADS) void conv_gather_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int FinP = a.Fin + 2;                 // [left zero][Fin][right zero]
    const int nrows = TF + a.KT - 1;            // LDS frame 0 <-> t0-(KT-1)
    const int K3 = a.Cin * a.KT * 3;
    float* wl = smem;                           // [K3][Cout]
    float* xl = smem + ((K3 * a.Cout + 3) & ~3); // [nrows][Cin][FinP]
    const int tid = threadIdx.x;
    const int ntile = (a.T + TF - 1) / TF;
    const int b = blockIdx.x / ntile;
    const int t0 = (blockIdx.x % ntile) * TF;
    // f64 cells: adding f32-valued partials in f64 is exact, so the result does not depend on the order of the atomics
    // (f32 cells made the batch mean vary by an ulp from run to run -- enough to flip a ReLU decision sitting on it)
    __shared__ double s_stat[2][64];
    if (a.sums && tid < 128) (&s_stat[0][0])[tid] = 0.0;        // ordered by the staging barrier below

    // stage weights: wl[(ci*KT+kt)*3+kf][co]
    for (int i = tid; i < K3 * a.Cout; i += CONV_THREADS) {
        const int co = i % a.Cout;
        const int k = i / a.Cout;
        const int kf = k % 3, kt = (k / 3) % a.KT, ci = k / (3 * a.KT);
        float v;
        if (a.w_layout == 0) v = a.w[((co * a.Cin + ci) * a.KT + kt) * 3 + kf];
        else v = a.w[(ci * a.Cout + co) * 3 + (2 - kf)];
        wl[i] = v;
    }
    // stage input frames (zero outside the clip and in the pad columns)
    stage_rows(xl, a.x, b, t0 - (a.KT - 1), nrows, a.Cin, a.Fin, a.T, tid);
    __syncthreads();

    const int tpf = (a.Cout / CO_T) * a.Fout;
    // wave-uniform trip count (the statistics epilogue below uses wave reductions): lanes past the end idle
    for (int base = 0; base < NSET * tpf; base += CONV_THREADS) {
        const int item = base + tid;
        const bool valid = item < NSET * tpf;
        const int set = valid ? item / tpf : 0;
        const int o = valid ? item % tpf : 0;
        const int cq = o / a.Fout, fo = o % a.Fout;
        const int co0 = cq * CO_T;
        float st1[CO_T], st2[CO_T];
#pragma unroll
        for (int c = 0; c < CO_T; ++c) { st1[c] = 0.f; st2[c] = 0.f; }
        if (valid) {
        float acc[TT][CO_T];
#pragma unroll
        for (int c = 0; c < CO_T; ++c) {
            const float bv = a.bias ? a.bias[co0 + c] : 0.f;
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) acc[tt][c] = bv;
        }
        const int fbase = fo * a.S - a.pad + 1;
        for (int ci = 0; ci < a.Cin; ++ci) {
            for (int kt = 0; kt < a.KT; ++kt) {
                float wv[3][CO_T];
                const float* wp = wl + ((ci * a.KT + kt) * 3) * a.Cout + co0;
#pragma unroll
                for (int kf = 0; kf < 3; ++kf) {
                    if constexpr (CO_T == 4) {
                        const float4 q = *reinterpret_cast<const float4*>(wp + kf * a.Cout);
                        wv[kf][0] = q.x; wv[kf][1] = q.y; wv[kf][2] = q.z; wv[kf][3] = q.w;
                    } else {
#pragma unroll
                        for (int c = 0; c < CO_T; ++c) wv[kf][c] = wp[kf * a.Cout + c];
                    }
                }
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    const float* xp = xl + ((set * TT + tt + kt) * a.Cin + ci) * FinP + fbase;
                    const float x0 = xp[0], x1 = xp[1], x2 = xp[2];
#pragma unroll
                    for (int c = 0; c < CO_T; ++c)
                        acc[tt][c] += wv[0][c] * x0 + wv[1][c] * x1 + wv[2][c] * x2;
                }
            }
        }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int t = t0 + set * TT + tt;
            if (t >= a.T) continue;
#pragma unroll
            for (int c = 0; c < CO_T; ++c) {
                const long long idx = (((long long)b * a.T + t) * a.Cout + co0 + c) * a.Fout + fo;
                float v = acc[tt][c];
                if (a.accum) v += a.y[idx]; else v = apply_act(v, a.act);
                a.y[idx] = v;
                st1[c] += v; st2[c] += v * v;
            }
        }
        }   // valid
        if (a.sums) {
            // a wave's 64 consecutive items span at most a few channel groups: one wave reduction (fixed order) per
            // group present, then ONE lane adds the group's 2 x CO_T sums to the LDS cells.  (Every lane adding its own
            // values made 320 threads queue on 16 cells: 95 us for the 1 -> 8 layer against 53 without statistics.)
            unsigned long long left = __ballot(1);
            while (left) {
                const int lead = __ffsll((long long)left) - 1;
                const int key = __shfl(cq, lead, 64);
                const bool mine = cq == key;
                float r1[CO_T], r2[CO_T];
#pragma unroll
                for (int c = 0; c < CO_T; ++c) { r1[c] = wave_sum(mine ? st1[c] : 0.f); r2[c] = wave_sum(mine ? st2[c] : 0.f); }
                if ((int)(threadIdx.x & 63) == lead) {
#pragma unroll
                    for (int c = 0; c < CO_T; ++c) {
                        atomicAdd(&s_stat[0][key * CO_T + c], (double)r1[c]);
                        atomicAdd(&s_stat[1][key * CO_T + c], (double)r2[c]);
                    }
                }
                left &= ~__ballot(mine);
            }
        }
    }
    if (a.sums) {       // Cout <= 64 (host-checked): 8 frames x Fout values per channel in f32, then one f64 atomic each
        __syncthreads();
        if (tid < 2 * a.Cout) {
            const int which = tid / a.Cout, co = tid - which * a.Cout;
            atomicAdd(a.sums + (size_t)(blockIdx.x % CRUSE_BN_STAT_REPLICAS) * 2 * a.Cout + which * a.Cout + co, s_stat[which][co]);
        }
    }
}

// ---------------------------------------------------------------------------
// scatter form, frequency stride 2 (outputs handled as (2m, 2m+1) pairs)
// ---------------------------------------------------------------------------
template <int CO_T>
__global__ __launch_bounds__(CONV_THREADS) void conv_scatter2_kernel(ConvArgs a) {
    // here: Cin = Cs (summed channels), Fin = Fg, Fout = 2*Fg
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int FgP = a.Fin + 2;
    const int nrows = TF + a.KT - 1;            // LDS frame 0 <-> t0
    const int K3 = a.Cin * a.KT * 3;
    float* wl = smem;                           // [K3][Cout]
    float* gl = smem + ((K3 * a.Cout + 3) & ~3);
    const int tid = threadIdx.x;
    const int ntile = (a.T + TF - 1) / TF;
    const int b = blockIdx.x / ntile;
    const int t0 = (blockIdx.x % ntile) * TF;

    for (int i = tid; i < K3 * a.Cout; i += CONV_THREADS) {
        const int co = i % a.Cout;
        const int k = i / a.Cout;               // (cs*KT+kt)*3+kf
        const int cs = k / (3 * a.KT), rem = k % (3 * a.KT);
        wl[i] = a.w[(cs * a.Cout + co) * (a.KT * 3) + rem];
    }
    stage_rows(gl, a.x, b, t0, nrows, a.Cin, a.Fin, a.T, tid);
    __syncthreads();

    const int Fg = a.Fin;
    const int tpf = (a.Cout / CO_T) * Fg;
    for (int item = tid; item < NSET * tpf; item += CONV_THREADS) {
        const int set = item / tpf;
        const int o = item % tpf;
        const int cq = o / Fg, m = o % Fg;
        const int co0 = cq * CO_T;
        float acc0[TT][CO_T], acc1[TT][CO_T];   // fo = 2m, 2m+1
#pragma unroll
        for (int c = 0; c < CO_T; ++c) {
            const float bv = a.bias ? a.bias[co0 + c] : 0.f;
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) { acc0[tt][c] = bv; acc1[tt][c] = bv; }
        }
        for (int cs = 0; cs < a.Cin; ++cs) {
            for (int kt = 0; kt < a.KT; ++kt) {
                float wv[3][CO_T];
                const float* wp = wl + ((cs * a.KT + kt) * 3) * a.Cout + co0;
#pragma unroll
                for (int kf = 0; kf < 3; ++kf)
#pragma unroll
                    for (int c = 0; c < CO_T; ++c) wv[kf][c] = wp[kf * a.Cout + c];
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    const int r = set * TT + tt + (a.KT - 1) - kt;
                    const float* gp = gl + (r * a.Cin + cs) * FgP + m;   // gp[0]=g[m-1], gp[1]=g[m], gp[2]=g[m+1]
                    const float gm1 = gp[0], g0 = gp[1], gp1 = gp[2];
                    if (a.pad == 0) {
#pragma unroll
                        for (int c = 0; c < CO_T; ++c) {
                            acc0[tt][c] += wv[0][c] * g0 + wv[2][c] * gm1;
                            acc1[tt][c] += wv[1][c] * g0;
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < CO_T; ++c) {
                            acc0[tt][c] += wv[1][c] * g0;
                            acc1[tt][c] += wv[0][c] * gp1 + wv[2][c] * g0;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int t = t0 + set * TT + tt;
            if (t >= a.T) continue;
#pragma unroll
            for (int c = 0; c < CO_T; ++c) {
                const long long idx = (((long long)b * a.T + t) * a.Cout + co0 + c) * a.Fout + 2 * m;
                float v0 = acc0[tt][c], v1 = acc1[tt][c];
                if (a.accum) { v0 += a.y[idx]; v1 += a.y[idx + 1]; }
                else { v0 = apply_act(v0, a.act); v1 = apply_act(v1, a.act); }
                *reinterpret_cast<float2*>(a.y + idx) = make_float2(v0, v1);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------
struct WgradArgs {
    const float* a; const float* bt; float* partial;
    int B, T, Ca, Fa, Cb, Fb, S, pad, ntiles_total;
};

constexpr int WG_THREADS = 256;
constexpr int WG_MAX_BLOCKS = 1024;

template <int CB_T, int KT>
__global__ __launch_bounds__(WG_THREADS) void conv_wgrad_kernel(WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CA_T = 4;
    constexpr int NACC = CA_T * CB_T * KT * 3;
    const int CaP = p.Ca + 4;                         // channel-fastest rows, 16B aligned
    const int CbP = (CB_T == 4) ? p.Cb + 4 : p.Cb;
    const int FbP = p.Fb + 2;
    const int nrowb = TF + KT - 1;
    float* al = smem;                                 // [TF][Fa][CaP]
    float* bl = smem + TF * p.Fa * CaP;               // [nrowb][FbP][CbP]
    const int tid = threadIdx.x;
    const int nta = p.Ca / CA_T, ntb = p.Cb / CB_T;
    const int NT = nta * ntb;
    const int ntile_t = (p.T + TF - 1) / TF;

    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;

    const int npos = TF * p.Fa;
    for (int tile = blockIdx.x; tile < p.ntiles_total; tile += gridDim.x) {
        const int b = tile / ntile_t;
        const int t0 = (tile % ntile_t) * TF;
        __syncthreads();
        for (int i = tid; i < TF * p.Ca * p.Fa; i += WG_THREADS) {
            const int fa = i % p.Fa, ca = (i / p.Fa) % p.Ca, r = i / (p.Fa * p.Ca);
            const int t = t0 + r;
            float v = 0.f;
            if (t < p.T) v = p.a[(((long long)b * p.T + t) * p.Ca + ca) * p.Fa + fa];
            al[(r * p.Fa + fa) * CaP + ca] = v;
        }
        for (int i = tid; i < nrowb * p.Cb * FbP; i += WG_THREADS) {
            const int fp = i % FbP, cb = (i / FbP) % p.Cb, r = i / (FbP * p.Cb);
            const int t = t0 - (KT - 1) + r;
            float v = 0.f;
            if (t >= 0 && t < p.T && fp >= 1 && fp <= p.Fb)
                v = p.bt[(((long long)b * p.T + t) * p.Cb + cb) * p.Fb + (fp - 1)];
            bl[(r * FbP + fp) * CbP + cb] = v;
        }
        __syncthreads();
        // work items: (tile-of-outputs, position); output tile fastest so a wave shares a position
        for (int w = tid; w < NT * npos; w += WG_THREADS) {
            const int ot = w % NT, pos = w / NT;
            const int ia = ot / ntb, ib = ot % ntb;
            const int r = pos / p.Fa, fa = pos % p.Fa;
            const float4 av = *reinterpret_cast<const float4*>(al + (r * p.Fa + fa) * CaP + ia * CA_T);
            const float a4[4] = {av.x, av.y, av.z, av.w};
            const int fb0 = fa * p.S - p.pad + 1;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int kf = 0; kf < 3; ++kf) {
                    const int fidx = fb0 + kf;
                    float bv[CB_T];
                    if (fidx >= 0 && fidx < FbP) {
                        const float* bp = bl + ((r + kt) * FbP + fidx) * CbP + ib * CB_T;
                        if constexpr (CB_T == 4) {
                            const float4 q = *reinterpret_cast<const float4*>(bp);
                            bv[0] = q.x; bv[1] = q.y; bv[2] = q.z; bv[3] = q.w;
                        } else {
                            bv[0] = bp[0];
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < CB_T; ++c) bv[c] = 0.f;
                    }
#pragma unroll
                    for (int i = 0; i < CA_T; ++i)
#pragma unroll
                        for (int j = 0; j < CB_T; ++j)
                            acc[((i * CB_T + j) * KT + kt) * 3 + kf] += a4[i] * bv[j];
                }
            }
        }
    }
    // NOTE: with w = tid + n*256 and NT dividing 256 (or 256 dividing NT*k) a thread keeps the
    // same output tile for all its items only when 256 % NT == 0; the host guarantees that.
    __syncthreads();
    float* red = smem;                                   // [Ca*Cb*KT*3]
    const int nout = p.Ca * p.Cb * KT * 3;
    for (int i = tid; i < nout; i += WG_THREADS) red[i] = 0.f;
    __syncthreads();
    {
        const int ot = tid % NT;
        const int ia = ot / ntb, ib = ot % ntb;
#pragma unroll
        for (int i = 0; i < CA_T; ++i)
#pragma unroll
            for (int j = 0; j < CB_T; ++j)
#pragma unroll
                for (int k = 0; k < KT * 3; ++k) {
                    const int ca = ia * CA_T + i, cb = ib * CB_T + j;
                    atomicAdd(&red[(ca * p.Cb + cb) * (KT * 3) + k], acc[(i * CB_T + j) * (KT * 3) + k]);
                }
    }
    __syncthreads();
    for (int i = tid; i < nout; i += WG_THREADS) p.partial[(long long)blockIdx.x * nout + i] = red[i];
}

__global__ void wgrad_reduce_kernel(const float* partial, int nblk, int nout, float* dw) {
    // blockIdx.y owns a chunk of 16 partial slabs; one atomic per output and chunk
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nout) return;
    const int k0 = blockIdx.y * 16, k1 = min(nblk, k0 + 16);
    float s = 0.f;
#pragma unroll 8
    for (int k = k0; k < k1; ++k) s += partial[(long long)k * nout + i];
    atomicAdd(&dw[i], s);
}

// ---------------------------------------------------------------------------
// channel sum: out[c] += sum_{rows,f} g[row,c,f]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* g, long long rows, int C, int F, int ld,
                                                          float* out) {
    // grid.x strides over rows; each thread owns a fixed column j of the C*F-wide row (row stride ld);
    // 4 independent accumulators keep 4 loads in flight per thread
    __shared__ float sacc[2048];
    const int CF = C * F;
    const int tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) sacc[c] = 0.f;
    __syncthreads();
    const long long G = gridDim.x;
    for (int j0 = 0; j0 < CF; j0 += 256) {
        const int j = j0 + tid;
        if (j < CF) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            long long r = blockIdx.x;
            for (; r + 3 * G < rows; r += 4 * G) {
                s0 += g[r * ld + j]; s1 += g[(r + G) * ld + j]; s2 += g[(r + 2 * G) * ld + j]; s3 += g[(r + 3 * G) * ld + j];
            }
            for (; r < rows; r += G) s0 += g[r * ld + j];
            atomicAdd(&sacc[j / F], (s0 + s1) + (s2 + s3));
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) atomicAdd(&out[c], sacc[c]);
}

template <typename K>
int set_smem(K kern, size_t bytes, const char* name) {
    return cruse_ensure_dyn_lds(reinterpret_cast<const void*>(kern), bytes, name);
}

}  // namespace

int cruse_conv_mfma_try(int scatter, const float* x, const float* w, const float* bias, float* y,
                        int B, int T, int Cin, int Fin, int Cout, int Fout, int KT, int S, int pad,
                        int w_layout, int act, int accum, int prec, double* bn_sums, const CruseBnBwd* bnb, int x_bf16, int y_bf16,
                        const CruseBnIn* bni, hipStream_t stream, const CruseBnBwdIn* bbi = nullptr);
extern "C" int cruse_bn_act_bwd_apply(const float* dout, const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                      const double* sums, int sum_replicas, long long rows, int C, int F, int relu, int training, int dout_dtype,
                                      void* dy, int dy_dtype, float* dgamma, float* dbeta, float* dbias, void* stream);
extern "C" int cruse_bn_act_bwd_reduce(const float* dout, const float* y, const float* mean, const float* rstd,
                                       const float* gamma, const float* beta, long long rows, int C, int F,
                                       int relu, double* sums, int zeroed, void* stream);

namespace {

int conv_gather_impl(const float* x, const float* w, const float* bias, float* y,
                     int B, int T, int Cin, int Fin, int Cout, int Fout,
                     int KT, int S, int pad, int w_layout, int act, int accum, int prec, double* bn_sums, void* stream,
                     const CruseBnBwd* bnb = nullptr, int x_dtype = CRUSE_DT_F32, int y_dtype = CRUSE_DT_F32, const CruseBnIn* bni = nullptr,
                     const CruseBnBwdIn* bbi = nullptr) {
    CRUSE_REQUIRE((x_dtype == CRUSE_DT_F32 || x_dtype == CRUSE_DT_BF16) && (y_dtype == CRUSE_DT_F32 || y_dtype == CRUSE_DT_BF16), CRUSE_E_DTYPE,
                  "conv_gather: x_dtype %d / y_dtype %d (f32 or bf16)", x_dtype, y_dtype);
    CRUSE_REQUIRE(B > 0 && T > 0 && Cin > 0 && Cout > 0 && Fin > 0 && Fout > 0, CRUSE_E_SHAPE,
                  "conv_gather: empty shape B=%d T=%d Cin=%d Cout=%d Fin=%d Fout=%d", B, T, Cin, Cout, Fin, Fout);
    CRUSE_REQUIRE((KT == 1 || KT == 2) && (S == 1 || S == 2) && (pad == 0 || pad == 1), CRUSE_E_SHAPE,
                  "conv_gather: unsupported KT=%d S=%d pad=%d", KT, S, pad);
    CRUSE_REQUIRE((Fout - 1) * S - pad + 2 <= Fin, CRUSE_E_SHAPE,
                  "conv_gather: Fout=%d reads past Fin=%d (+1 zero column)", Fout, Fin);
    CRUSE_REQUIRE(w_layout == 0 || (KT == 1 && S == 1), CRUSE_E_SHAPE, "conv_gather: w_layout 1 needs KT=1,S=1");
    CRUSE_REQUIRE(!(accum && act), CRUSE_E_SHAPE, "conv_gather: accum with activation");
    if (prec >= 0) {
        const int r = cruse_conv_mfma_try(0, x, w, bias, y, B, T, Cin, Fin, Cout, Fout, KT, S, pad, w_layout, act, accum,
                                          prec, bn_sums, bnb, x_dtype == CRUSE_DT_BF16, y_dtype == CRUSE_DT_BF16, bni, (hipStream_t)stream, bbi);
        if (r != 0) return r < 0 ? r : CRUSE_OK;
    }
    if (bbi != nullptr) return 1;                    // (not handled: the caller runs the separate BatchNorm-backward pass)
    CRUSE_REQUIRE(bni == nullptr, CRUSE_E_SHAPE, "conv_gather_bnin: the fused input BatchNorm needs the MFMA kernel in the bf16 mode (Cin %d, Cout %d, prec %d)", Cin, Cout, prec);
    CRUSE_REQUIRE(x_dtype == CRUSE_DT_F32 && y_dtype == CRUSE_DT_F32, CRUSE_E_DTYPE,
                  "conv_gather: a bf16 input / output needs the MFMA kernel in the bf16 data-gradient mode (Cin %d, Cout %d, prec %d)", Cin, Cout, prec);
    const bool fuse = bn_sums && Cout <= 64 && bnb == nullptr;
    ConvArgs a{x, w, bias, y, B, T, Cin, Fin, Cout, Fout, KT, S, pad, w_layout, act, accum, fuse ? bn_sums : nullptr};
    const size_t lds = (((size_t)Cin * KT * 3 * Cout + 3) & ~(size_t)3) * 4 +
                       (size_t)(TF + KT - 1) * Cin * (Fin + 2) * 4;
    CRUSE_REQUIRE(lds <= 160 * 1024, CRUSE_E_SHAPE, "conv_gather: tile needs %zu B of LDS", lds);
    const int grid = B * cdiv(T, TF);
    int rc;
    if (Cout % 4 == 0) {
        if ((rc = set_smem(conv_gather_kernel<4>, lds, "conv_gather"))) return rc;
        hipLaunchKernelGGL(conv_gather_kernel<4>, dim3(grid), dim3(CONV_THREADS), lds, (hipStream_t)stream, a);
    } else {
        if ((rc = set_smem(conv_gather_kernel<1>, lds, "conv_gather"))) return rc;
        hipLaunchKernelGGL(conv_gather_kernel<1>, dim3(grid), dim3(CONV_THREADS), lds, (hipStream_t)stream, a);
    }
    CRUSE_LAUNCH_CHECK("conv_gather");
    // no statistics epilogue on this path: one more pass (the backward sums go to replica 0, the others stay zero)
    if (bnb) return cruse_bn_act_bwd_reduce(y, bnb->y, bnb->mean, bnb->rstd, bnb->gamma, bnb->beta, (long long)B * T, Cout, Fout,
                                            bnb->relu, bn_sums, 1, stream);
    if (bn_sums && !fuse) return cruse_bn_stats(y, (long long)B * T, Cout, Fout, bn_sums, 1, stream);
    return CRUSE_OK;
}

int conv_scatter2_impl(const float* g, const float* w, const float* bias, float* y,
                       int B, int T, int Cs, int Fg, int Cout, int Fout,
                       int KT, int pad, int act, int accum, int prec, double* bn_sums, void* stream,
                       const CruseBnBwd* bnb = nullptr, int x_dtype = CRUSE_DT_F32, int y_dtype = CRUSE_DT_F32, const CruseBnIn* bni = nullptr,
                       const CruseBnBwdIn* bbi = nullptr) {
    CRUSE_REQUIRE((x_dtype == CRUSE_DT_F32 || x_dtype == CRUSE_DT_BF16) && (y_dtype == CRUSE_DT_F32 || y_dtype == CRUSE_DT_BF16), CRUSE_E_DTYPE,
                  "conv_scatter2: x_dtype %d / y_dtype %d (f32 or bf16)", x_dtype, y_dtype);
    CRUSE_REQUIRE(B > 0 && T > 0 && Cs > 0 && Cout > 0 && Fg > 0, CRUSE_E_SHAPE, "conv_scatter2: empty shape");
    CRUSE_REQUIRE(Fout == 2 * Fg, CRUSE_E_SHAPE, "conv_scatter2: Fout=%d must be 2*Fg=%d", Fout, 2 * Fg);
    CRUSE_REQUIRE((KT == 1 || KT == 2) && (pad == 0 || pad == 1), CRUSE_E_SHAPE,
                  "conv_scatter2: unsupported KT=%d pad=%d", KT, pad);
    CRUSE_REQUIRE(!(accum && act), CRUSE_E_SHAPE, "conv_scatter2: accum with activation");
    if (prec >= 0) {
        const int r = cruse_conv_mfma_try(1, g, w, bias, y, B, T, Cs, Fg, Cout, Fout, KT, 2, pad, 0, act, accum, prec,
                                          bn_sums, bnb, x_dtype == CRUSE_DT_BF16, y_dtype == CRUSE_DT_BF16, bni, (hipStream_t)stream, bbi);
        if (r != 0) return r < 0 ? r : CRUSE_OK;
    }
    if (bbi != nullptr) return 1;                    // (not handled: the caller runs the separate BatchNorm-backward pass)
    CRUSE_REQUIRE(bni == nullptr, CRUSE_E_SHAPE, "conv_scatter2_bnin: the fused input BatchNorm needs the MFMA kernel in the bf16 mode (Cs %d, Cout %d, prec %d)", Cs, Cout, prec);
    CRUSE_REQUIRE(x_dtype == CRUSE_DT_F32 && y_dtype == CRUSE_DT_F32, CRUSE_E_DTYPE,
                  "conv_scatter2: a bf16 input / output needs the MFMA kernel in the bf16 data-gradient mode (Cs %d, Cout %d, prec %d)", Cs, Cout, prec);
    ConvArgs a{g, w, bias, y, B, T, Cs, Fg, Cout, Fout, KT, 2, pad, 0, act, accum, nullptr};
    const size_t lds = (((size_t)Cs * KT * 3 * Cout + 3) & ~(size_t)3) * 4 +
                       (size_t)(TF + KT - 1) * Cs * (Fg + 2) * 4;
    CRUSE_REQUIRE(lds <= 160 * 1024, CRUSE_E_SHAPE, "conv_scatter2: tile needs %zu B of LDS", lds);
    const int grid = B * cdiv(T, TF);
    int rc;
    if (Cout % 2 == 0) {
        if ((rc = set_smem(conv_scatter2_kernel<2>, lds, "conv_scatter2"))) return rc;
        hipLaunchKernelGGL(conv_scatter2_kernel<2>, dim3(grid), dim3(CONV_THREADS), lds, (hipStream_t)stream, a);
    } else {
        if ((rc = set_smem(conv_scatter2_kernel<1>, lds, "conv_scatter2"))) return rc;
        hipLaunchKernelGGL(conv_scatter2_kernel<1>, dim3(grid), dim3(CONV_THREADS), lds, (hipStream_t)stream, a);
    }
    CRUSE_LAUNCH_CHECK("conv_scatter2");
    // the VALU scatter kernel has no statistics epilogue: one more pass over y
    if (bnb) return cruse_bn_act_bwd_reduce(y, bnb->y, bnb->mean, bnb->rstd, bnb->gamma, bnb->beta, (long long)B * T, Cout, Fout,
                                            bnb->relu, bn_sums, 1, stream);
    if (bn_sums) return cruse_bn_stats(y, (long long)B * T, Cout, Fout, bn_sums, 1, stream);
    return CRUSE_OK;
}

int prep_sums(double* sums, int Cout, int zeroed, void* stream, const char* who) {
    CRUSE_REQUIRE(sums != nullptr, CRUSE_E_SHAPE, "%s: sums is NULL", who);
    if (!zeroed) return cruse_zero_async(sums, 2 * (size_t)Cout * CRUSE_BN_STAT_REPLICAS * sizeof(double), (hipStream_t)stream, who);
    return CRUSE_OK;
}

}  // namespace

extern "C" int cruse_conv_gather(const float* x, const float* w, const float* bias, float* y,
                                 int B, int T, int Cin, int Fin, int Cout, int Fout,
                                 int KT, int S, int pad, int w_layout, int act, int accum, int prec, int x_dtype, int y_dtype, void* stream) {
    return conv_gather_impl(x, w, bias, y, B, T, Cin, Fin, Cout, Fout, KT, S, pad, w_layout, act, accum, prec, nullptr, stream, nullptr,
                            x_dtype, y_dtype);
}

extern "C" int cruse_conv_scatter2(const float* g, const float* w, const float* bias, float* y,
                                   int B, int T, int Cs, int Fg, int Cout, int Fout,
                                   int KT, int pad, int act, int accum, int prec, int x_dtype, int y_dtype, void* stream) {
    return conv_scatter2_impl(g, w, bias, y, B, T, Cs, Fg, Cout, Fout, KT, pad, act, accum, prec, nullptr, stream, nullptr, x_dtype, y_dtype);
}

extern "C" int cruse_conv_gather_bnstats(const float* x, const float* w, const float* bias, float* y,
                                         int B, int T, int Cin, int Fin, int Cout, int Fout,
                                         int KT, int S, int pad, int prec, double* sums, int zeroed, void* stream) {
    int rc = prep_sums(sums, Cout, zeroed, stream, "conv_gather_bnstats");
    if (rc) return rc;
    return conv_gather_impl(x, w, bias, y, B, T, Cin, Fin, Cout, Fout, KT, S, pad, 0, 0, 0, prec, sums, stream);
}

extern "C" int cruse_conv_scatter2_bnstats(const float* g, const float* w, const float* bias, float* y,
                                           int B, int T, int Cs, int Fg, int Cout, int Fout,
                                           int KT, int pad, int prec, double* sums, int zeroed, void* stream) {
    int rc = prep_sums(sums, Cout, zeroed, stream, "conv_scatter2_bnstats");
    if (rc) return rc;
    return conv_scatter2_impl(g, w, bias, y, B, T, Cs, Fg, Cout, Fout, KT, pad, 0, 0, prec, sums, stream);
}

extern "C" int cruse_conv_gather_bnbwd(const float* x, const float* w, float* y, int B, int T, int Cin, int Fin, int Cout, int Fout,
                                       int KT, int S, int pad, int w_layout, int accum, int prec,
                                       const float* bn_y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                       int relu, double* sums, int zeroed, int x_dtype, int y_dtype, void* stream) {
    CRUSE_REQUIRE(bn_y && mean && rstd && gamma && beta, CRUSE_E_SHAPE, "conv_gather_bnbwd: BatchNorm tensors missing");
    int rc = prep_sums(sums, Cout, zeroed, stream, "conv_gather_bnbwd");
    if (rc) return rc;
    const CruseBnBwd bnb = {bn_y, mean, rstd, gamma, beta, relu};
    return conv_gather_impl(x, w, nullptr, y, B, T, Cin, Fin, Cout, Fout, KT, S, pad, w_layout, 0, accum, prec, sums, stream, &bnb, x_dtype, y_dtype);
}

extern "C" int cruse_conv_scatter2_bnbwd(const float* g, const float* w, float* y, int B, int T, int Cs, int Fg, int Cout, int Fout,
                                         int KT, int pad, int accum, int prec,
                                         const float* bn_y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                         int relu, double* sums, int zeroed, int x_dtype, int y_dtype, void* stream) {
    CRUSE_REQUIRE(bn_y && mean && rstd && gamma && beta, CRUSE_E_SHAPE, "conv_scatter2_bnbwd: BatchNorm tensors missing");
    int rc = prep_sums(sums, Cout, zeroed, stream, "conv_scatter2_bnbwd");
    if (rc) return rc;
    const CruseBnBwd bnb = {bn_y, mean, rstd, gamma, beta, relu};
    return conv_scatter2_impl(g, w, nullptr, y, B, T, Cs, Fg, Cout, Fout, KT, pad, 0, accum, prec, sums, stream, &bnb, x_dtype, y_dtype);
}

extern "C" int cruse_conv_gather_bnin(const float* x_pre, const double* in_sums, int in_replicas, long long in_count, float eps, float momentum,
                                      const float* in_gamma, const float* in_beta, float* in_mean, float* in_rstd, float* in_running_mean,
                                      float* in_running_var, const float* in_add, void* in_copy_bf16,
                                      const float* w, const float* bias, float* y, int B, int T, int Cin, int Fin, int Cout, int Fout,
                                      int KT, int S, int pad, int prec, double* out_sums, int zeroed, void* stream) {
    CRUSE_REQUIRE(x_pre && in_sums && in_gamma && in_beta && in_replicas >= 1 && in_count > 0, CRUSE_E_SHAPE, "conv_gather_bnin: input BatchNorm tensors missing");
    CRUSE_REQUIRE((in_mean == nullptr) == (in_rstd == nullptr) && (in_running_mean == nullptr) == (in_running_var == nullptr), CRUSE_E_SHAPE,
                  "conv_gather_bnin: mean / rstd and the running statistics come in pairs");
    if (out_sums) { int rc = prep_sums(out_sums, Cout, zeroed, stream, "conv_gather_bnin"); if (rc) return rc; }
    const CruseBnIn bni = {in_sums, in_replicas, in_count, eps, momentum, in_gamma, in_beta, in_mean, in_rstd, in_running_mean, in_running_var,
                           in_add, in_copy_bf16};
    return conv_gather_impl(x_pre, w, bias, y, B, T, Cin, Fin, Cout, Fout, KT, S, pad, 0, 0, 0, prec, out_sums, stream, nullptr, CRUSE_DT_F32,
                            CRUSE_DT_F32, &bni);
}

extern "C" int cruse_conv_scatter2_bnin(const float* g_pre, const double* in_sums, int in_replicas, long long in_count, float eps, float momentum,
                                        const float* in_gamma, const float* in_beta, float* in_mean, float* in_rstd, float* in_running_mean,
                                        float* in_running_var, const float* in_add, void* in_copy_bf16,
                                        const float* w, const float* bias, float* y, int B, int T, int Cs, int Fg, int Cout, int Fout,
                                        int KT, int pad, int prec, double* out_sums, int zeroed, void* stream) {
    CRUSE_REQUIRE(g_pre && in_sums && in_gamma && in_beta && in_replicas >= 1 && in_count > 0, CRUSE_E_SHAPE, "conv_scatter2_bnin: input BatchNorm tensors missing");
    CRUSE_REQUIRE((in_mean == nullptr) == (in_rstd == nullptr) && (in_running_mean == nullptr) == (in_running_var == nullptr), CRUSE_E_SHAPE,
                  "conv_scatter2_bnin: mean / rstd and the running statistics come in pairs");
    if (out_sums) { int rc = prep_sums(out_sums, Cout, zeroed, stream, "conv_scatter2_bnin"); if (rc) return rc; }
    const CruseBnIn bni = {in_sums, in_replicas, in_count, eps, momentum, in_gamma, in_beta, in_mean, in_rstd, in_running_mean, in_running_var,
                           in_add, in_copy_bf16};
    return conv_scatter2_impl(g_pre, w, bias, y, B, T, Cs, Fg, Cout, Fout, KT, pad, 0, 0, prec, out_sums, stream, nullptr, CRUSE_DT_F32,
                              CRUSE_DT_F32, &bni);
}

// Data-gradient convolutions with the BatchNorm(+ReLU) BACKWARD of their input applied while staging (see CruseBnBwdIn): one entry point =
// cruse_bn_act_bwd_apply(dout -> dy_bf16, parameter gradients) + cruse_conv_*[_bnbwd](dy_bf16 -> y).  Shapes / modes the MFMA kernel does not
// take run exactly those two calls.
static int bnbwd_in_fallback(const CruseBnBwdIn& bb, const void* dout, int dout_dtype, long long rows, int C, int F, void* stream) {
    CRUSE_REQUIRE(bb.copy_bf16 != nullptr, CRUSE_E_SHAPE, "conv_*_bnbwd_in: dy_bf16 is required");
    return cruse_bn_act_bwd_apply(reinterpret_cast<const float*>(dout), bb.y, bb.mean, bb.rstd, bb.gamma, bb.beta, bb.sums, bb.nrep, rows, C, F,
                                  bb.relu, bb.training, dout_dtype, bb.copy_bf16, CRUSE_DT_BF16, bb.dgamma, bb.dbeta, bb.dbias, stream);
}

extern "C" int cruse_conv_gather_bnbwd_in(const void* dout, int dout_dtype, const float* in_y, const float* in_mean, const float* in_rstd,
                                          const float* in_gamma, const float* in_beta, const double* in_sums, int in_replicas, int in_relu,
                                          int in_training, void* dy_bf16, float* in_dgamma, float* in_dbeta, float* in_dbias,
                                          const float* w, void* y, int B, int T, int Cin, int Fin, int Cout, int Fout, int KT, int S, int pad,
                                          int w_layout, int accum, int prec,
                                          const float* bn_y, const float* mean, const float* rstd, const float* gamma, const float* beta, int relu,
                                          double* sums, int zeroed, int y_dtype, void* stream) {
    CRUSE_REQUIRE(dout && in_y && in_mean && in_rstd && in_gamma && in_beta && in_sums && in_replicas >= 1, CRUSE_E_SHAPE,
                  "conv_gather_bnbwd_in: input BatchNorm tensors missing");
    CRUSE_REQUIRE((bn_y == nullptr) == (sums == nullptr), CRUSE_E_SHAPE, "conv_gather_bnbwd_in: bn_y and sums come together");
    if (sums) { int rc = prep_sums(sums, Cout, zeroed, stream, "conv_gather_bnbwd_in"); if (rc) return rc; }
    const CruseBnBwdIn bb = {in_y, in_sums, in_replicas, (long long)B * T * Fin, in_mean, in_rstd, in_gamma, in_beta, in_relu, in_training, dy_bf16,
                             in_dgamma, in_dbeta, in_dbias};
    const CruseBnBwd bnb = {bn_y, mean, rstd, gamma, beta, relu};
    if (dout_dtype == CRUSE_DT_BF16) {
        const int rc = conv_gather_impl(reinterpret_cast<const float*>(dout), w, nullptr, reinterpret_cast<float*>(y), B, T, Cin, Fin, Cout, Fout, KT, S,
                                        pad, w_layout, 0, accum, prec, sums, stream, bn_y ? &bnb : nullptr, CRUSE_DT_BF16, y_dtype, nullptr, &bb);
        if (rc <= 0) return rc;
    }
    int rc = bnbwd_in_fallback(bb, dout, dout_dtype, (long long)B * T, Cin, Fin, stream);
    if (rc) return rc;
    return conv_gather_impl(reinterpret_cast<const float*>(dy_bf16), w, nullptr, reinterpret_cast<float*>(y), B, T, Cin, Fin, Cout, Fout, KT, S, pad,
                            w_layout, 0, accum, prec, sums, stream, bn_y ? &bnb : nullptr, CRUSE_DT_BF16, y_dtype);
}

extern "C" int cruse_conv_scatter2_bnbwd_in(const void* dout, int dout_dtype, const float* in_y, const float* in_mean, const float* in_rstd,
                                            const float* in_gamma, const float* in_beta, const double* in_sums, int in_replicas, int in_relu,
                                            int in_training, void* dy_bf16, float* in_dgamma, float* in_dbeta, float* in_dbias,
                                            const float* w, void* y, int B, int T, int Cs, int Fg, int Cout, int Fout, int KT, int pad, int accum,
                                            int prec,
                                            const float* bn_y, const float* mean, const float* rstd, const float* gamma, const float* beta, int relu,
                                            double* sums, int zeroed, int y_dtype, void* stream) {
    CRUSE_REQUIRE(dout && in_y && in_mean && in_rstd && in_gamma && in_beta && in_sums && in_replicas >= 1, CRUSE_E_SHAPE,
                  "conv_scatter2_bnbwd_in: input BatchNorm tensors missing");
    CRUSE_REQUIRE((bn_y == nullptr) == (sums == nullptr), CRUSE_E_SHAPE, "conv_scatter2_bnbwd_in: bn_y and sums come together");
    if (sums) { int rc = prep_sums(sums, Cout, zeroed, stream, "conv_scatter2_bnbwd_in"); if (rc) return rc; }
    const CruseBnBwdIn bb = {in_y, in_sums, in_replicas, (long long)B * T * Fg, in_mean, in_rstd, in_gamma, in_beta, in_relu, in_training, dy_bf16,
                             in_dgamma, in_dbeta, in_dbias};
    const CruseBnBwd bnb = {bn_y, mean, rstd, gamma, beta, relu};
    if (dout_dtype == CRUSE_DT_BF16) {
        const int rc = conv_scatter2_impl(reinterpret_cast<const float*>(dout), w, nullptr, reinterpret_cast<float*>(y), B, T, Cs, Fg, Cout, Fout, KT, pad,
                                          0, accum, prec, sums, stream, bn_y ? &bnb : nullptr, CRUSE_DT_BF16, y_dtype, nullptr, &bb);
        if (rc <= 0) return rc;
    }
    int rc = bnbwd_in_fallback(bb, dout, dout_dtype, (long long)B * T, Cs, Fg, stream);
    if (rc) return rc;
    return conv_scatter2_impl(reinterpret_cast<const float*>(dy_bf16), w, nullptr, reinterpret_cast<float*>(y), B, T, Cs, Fg, Cout, Fout, KT, pad, 0,
                              accum, prec, sums, stream, bn_y ? &bnb : nullptr, CRUSE_DT_BF16, y_dtype);
}

extern "C" size_t cruse_conv_wgrad_ws_bytes(int Ca, int Cb, int KT) {
    // partial slabs: [slab][Ca][Cb][KT][3] (LDS-staged and VALU kernels) or accumulator images of 16 x 16 tiles (register-direct kernel)
    return (size_t)WG_MAX_BLOCKS * ((Ca + 15) / 16 * 16) * ((Cb + 15) / 16 * 16) * KT * 3 * sizeof(float);
}

int cruse_wgrad_mfma_try(const float* a, const float* bt, float* partial, int max_slabs,
                         int B, int T, int Ca, int Fa, int Cb, int Fb, int KT, int S, int pad, int prec, int a_bf16, int bt_bf16,
                         int* nblk_out, hipStream_t stream);
int cruse_wgrad_rd_try(const float* a, const float* bt, float* partial, size_t ws_bytes, float* dw,
                       int B, int T, int Ca, int Fa, int Cb, int Fb, int KT, int S, int pad, int prec, int a_bf16, int bt_bf16,
                       int* nblk_out, hipStream_t stream);

extern "C" int cruse_conv_wgrad(const float* a, const float* bt, float* dw,
                                int B, int T, int Ca, int Fa, int Cb, int Fb,
                                int KT, int S, int pad, int prec, int a_dtype, int bt_dtype, void* ws, void* stream) {
    CRUSE_REQUIRE(B > 0 && T > 0 && Ca > 0 && Cb > 0, CRUSE_E_SHAPE, "conv_wgrad: empty shape");
    CRUSE_REQUIRE((a_dtype == CRUSE_DT_F32 || a_dtype == CRUSE_DT_BF16) && (bt_dtype == CRUSE_DT_F32 || bt_dtype == CRUSE_DT_BF16),
                  CRUSE_E_DTYPE, "conv_wgrad: operand dtypes %d / %d (f32 or bf16)", a_dtype, bt_dtype);
    CRUSE_REQUIRE((KT == 1 || KT == 2) && (S == 1 || S == 2) && (pad == 0 || pad == 1), CRUSE_E_SHAPE,
                  "conv_wgrad: unsupported KT=%d S=%d pad=%d", KT, S, pad);
    if (prec >= 0) {
        int nblk = 0;
        // register-direct stream (plain bf16 mode, the bench shapes), else the LDS-staged kernel
        int r = cruse_wgrad_rd_try(a, bt, (float*)ws, cruse_conv_wgrad_ws_bytes(Ca, Cb, KT), dw, B, T, Ca, Fa, Cb, Fb, KT, S, pad, prec,
                                   a_dtype == CRUSE_DT_BF16, bt_dtype == CRUSE_DT_BF16, &nblk, (hipStream_t)stream);
        if (r == 0) r = cruse_wgrad_mfma_try(a, bt, (float*)ws, WG_MAX_BLOCKS, B, T, Ca, Fa, Cb, Fb, KT, S, pad, prec,
                                             a_dtype == CRUSE_DT_BF16, bt_dtype == CRUSE_DT_BF16, &nblk, (hipStream_t)stream);
        if (r < 0) return r;
        if (r == 1 && nblk == 0) return CRUSE_OK;      // (the register-direct kernel reduces its slabs itself)
        if (r == 1) {
            const int nout = Ca * Cb * KT * 3;
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(nout, 256), cdiv(nblk, 16)), dim3(256), 0, (hipStream_t)stream,
                               (const float*)ws, nblk, nout, dw);
            CRUSE_LAUNCH_CHECK("conv_wgrad_reduce");
            return CRUSE_OK;
        }
    }
    CRUSE_REQUIRE(a_dtype == CRUSE_DT_F32 && bt_dtype == CRUSE_DT_F32, CRUSE_E_DTYPE,
                  "conv_wgrad: bf16 operands need the MFMA kernel (Ca %d, Cb %d, prec %d)", Ca, Cb, prec);
    CRUSE_REQUIRE(Ca % 4 == 0 && (Cb % 4 == 0 || Cb == 1), CRUSE_E_SHAPE,
                  "conv_wgrad: Ca=%d must be a multiple of 4 and Cb=%d a multiple of 4 or 1", Ca, Cb);
    CRUSE_REQUIRE((KT == 1 || KT == 2) && (S == 1 || S == 2) && (pad == 0 || pad == 1), CRUSE_E_SHAPE,
                  "conv_wgrad: unsupported KT=%d S=%d pad=%d", KT, S, pad);
    const int cbt = (Cb % 4 == 0) ? 4 : 1;
    const int NT = (Ca / 4) * (Cb / cbt);
    CRUSE_REQUIRE(WG_THREADS % NT == 0 || NT % WG_THREADS == 0, CRUSE_E_SHAPE,
                  "conv_wgrad: %d output tiles do not divide the %d-thread block", NT, WG_THREADS);
    // a thread must keep one output tile: items w = tid + n*256 have ot = w % NT = tid % NT iff 256 % NT == 0
    CRUSE_REQUIRE(WG_THREADS % NT == 0, CRUSE_E_SHAPE, "conv_wgrad: %d output tiles > %d threads", NT, WG_THREADS);
    const int CaP = Ca + 4, CbP = (cbt == 4) ? Cb + 4 : Cb;
    const size_t lds_stage = ((size_t)TF * Fa * CaP + (size_t)(TF + KT - 1) * (Fb + 2) * CbP) * 4;
    const size_t lds_red = (size_t)Ca * Cb * KT * 3 * 4;
    const size_t lds = lds_stage > lds_red ? lds_stage : lds_red;
    CRUSE_REQUIRE(lds <= 160 * 1024, CRUSE_E_SHAPE, "conv_wgrad: needs %zu B of LDS", lds);
    const int ntiles = B * cdiv(T, TF);
    const int grid = ntiles < WG_MAX_BLOCKS ? ntiles : WG_MAX_BLOCKS;
    WgradArgs p{a, bt, (float*)ws, B, T, Ca, Fa, Cb, Fb, S, pad, ntiles};
    int rc;
#define LAUNCH_WG(CBT, KTT)                                                                             \
    do {                                                                                                \
        if ((rc = set_smem(conv_wgrad_kernel<CBT, KTT>, lds, "conv_wgrad"))) return rc;                 \
        hipLaunchKernelGGL((conv_wgrad_kernel<CBT, KTT>), dim3(grid), dim3(WG_THREADS), lds,            \
                           (hipStream_t)stream, p);                                                     \
    } while (0)
    if (cbt == 4 && KT == 2) LAUNCH_WG(4, 2);
    else if (cbt == 4 && KT == 1) LAUNCH_WG(4, 1);
    else if (cbt == 1 && KT == 2) LAUNCH_WG(1, 2);
    else LAUNCH_WG(1, 1);
#undef LAUNCH_WG
    CRUSE_LAUNCH_CHECK("conv_wgrad");
    const int nout = Ca * Cb * KT * 3;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(nout, 256), cdiv(grid, 16)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)ws, grid, nout, dw);
    CRUSE_LAUNCH_CHECK("conv_wgrad_reduce");
    return CRUSE_OK;
}

extern "C" int cruse_channel_sum(const float* g, long long rows, int C, int F, float* out, void* stream) {
    CRUSE_REQUIRE(rows > 0 && C > 0 && F > 0 && C <= 2048, CRUSE_E_SHAPE,
                  "channel_sum: bad shape rows=%lld C=%d F=%d (C <= 2048)", rows, C, F);
    long long nb = rows < 512 ? rows : 512;
    hipLaunchKernelGGL(channel_sum_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, g, rows, C, F, C * F, out);
    CRUSE_LAUNCH_CHECK("channel_sum");
    return CRUSE_OK;
}

extern "C" int cruse_col_sum(const float* g, long long rows, int ncol, int ld, float* out, void* stream) {
    CRUSE_REQUIRE(rows > 0 && ncol > 0 && ncol <= 2048 && ld >= ncol, CRUSE_E_SHAPE,
                  "col_sum: bad shape rows=%lld ncol=%d ld=%d (ncol <= 2048)", rows, ncol, ld);
    long long nb = rows < 512 ? rows : 512;
    hipLaunchKernelGGL(channel_sum_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, g, rows, ncol, 1, ld, out);
    CRUSE_LAUNCH_CHECK("col_sum");
    return CRUSE_OK;
}
