// HBM-bound frame-row kernels: BatchNorm2d (training statistics, +ReLU, +skip add),
// LayerNorm (+ group interleave, + residual), magnitude-mask application + WO-MALE loss,
// fused Adam.  All activations are [rows = B*T][C][F] f32 with C*F contiguous per frame.
//
// Reference call sites: model/cruse_net.py:141-142,149-152,161-163 (BatchNorm2d + ReLU + skip
// add), :32-33,43-51 (LayerNorm + stack/flatten interleave), :160 (gru + skip4),
// utils/utils.py:418-420 (mask application), loss_func/loss.py:121-148 (WO-MALE),
// tools/train_stand.py:68-71 (Adam).
#include "common.h"
#include <stdlib.h>

namespace {

// ------------------------------------------------------------------ BatchNorm ---
// sums[c] += sum v, sums[C+c] += sum v2 where (v, v2) come from functor fn per element.
// Blocks are RED_THREADS wide so that few blocks (<= one per CU) end in global atomics on the same 2C addresses --
// with 1024 small blocks those same-address atomics, serialised in L2, took longer than streaming the tensor.
// One WAVE per row (grid-stride over rows): lane l owns the float4 column groups l, l+64, l+128, ... of the
// C*F-wide frame row, so every access is a 16-byte load and a lane's partial sums stay in registers across
// all its rows (f32 runs of 32 rows folded into f64); the per-channel reduction happens once at the end.
constexpr int RED_THREADS = 1024;
constexpr int CPR_MAXG = 3;          // float4 groups per lane: C*F <= 768

template <typename Fn>
__device__ __forceinline__ void channel_pair_reduce(long long rows, int C, int F, double* sums, Fn fn) {
    // Per-COLUMN f64 sums in LDS first, folded per channel once at the end.  (The first version added each lane's
    // partials straight into per-channel LDS cells: with F = 80 columns per channel 20 neighbouring lanes hit the
    // same cell, the f64 LDS atomics serialised, and the pass ran at 1.9 TB/s against 6.4 for the BatchNorm forward.)
    __shared__ double c1[64 * CPR_MAXG * 4], c2[64 * CPR_MAXG * 4];
    __shared__ double s1[256], s2[256];
    const int CF = C * F, ng = CF >> 2;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int c = tid; c < C; c += blockDim.x) { s1[c] = 0.0; s2[c] = 0.0; }
    for (int i = tid; i < CF; i += blockDim.x) { c1[i] = 0.0; c2[i] = 0.0; }
    __syncthreads();
    float a[CPR_MAXG][4], b[CPR_MAXG][4];
#pragma unroll
    for (int g = 0; g < CPR_MAXG; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[g][e] = b[g][e] = 0.f; }
    auto flush = [&]() {                       // f32 runs of <= 32 rows are folded into the f64 LDS column sums
#pragma unroll
        for (int g = 0; g < CPR_MAXG; ++g) {
            const int q = lane + 64 * g;
            if (q < ng) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    atomicAdd(&c1[q * 4 + e], (double)a[g][e]);
                    atomicAdd(&c2[q * 4 + e], (double)b[g][e]);
                    a[g][e] = b[g][e] = 0.f;
                }
            }
        }
    };
    const int nw = blockDim.x >> 6;
    const long long wave = (long long)blockIdx.x * nw + (tid >> 6), nwave = (long long)gridDim.x * nw;
    int n = 0;
    // two rows per trip: their loads are independent and issue back to back (twice the bytes in flight per wave)
    for (long long r = wave; r < rows; r += 2 * nwave) {
        const bool two = r + nwave < rows;
#pragma unroll
        for (int g = 0; g < CPR_MAXG; ++g) {
            const int q = lane + 64 * g;
            if (q < ng) {
                float v[4], v2[4], w[4], w2[4];
                fn(r * CF + q * 4, g, v, v2);
                if (two) fn((r + nwave) * CF + q * 4, g, w, w2);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { w[e] = 0.f; w2[e] = 0.f; }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[g][e] += v[e] + w[e]; b[g][e] += v2[e] + w2[e]; }
            }
        }
        n += 2;
        if (n >= 32) { flush(); n = 0; }
    }
    flush();
    __syncthreads();
    for (int i = tid; i < CF; i += blockDim.x) {           // once per workgroup: contention here is immaterial
        const int c = i / F;
        atomicAdd(&s1[c], c1[i]);
        atomicAdd(&s2[c], c2[i]);
    }
    __syncthreads();
    for (int c = tid; c < C; c += blockDim.x) {
        atomicAdd(&sums[c], s1[c]);
        atomicAdd(&sums[C + c], s2[c]);
    }
}

__global__ __launch_bounds__(RED_THREADS) void bn_stats_kernel(const float* y, long long rows, int C, int F, double* sums) {
    channel_pair_reduce(rows, C, F, sums, [&](long long i, int, float (&v)[4], float (&v2)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(y + i);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) v2[e] = v[e] * v[e];
    });
}

__global__ void bn_finalize_kernel(const double* sums, long long count, int C, float eps, float momentum,
                                   float* mean, float* rstd, float* rmean, float* rvar) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = sums[c] / (double)count;
    double var = sums[C + c] / (double)count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
        const double unb = count > 1 ? var * (double)count / (double)(count - 1) : var;
        rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * m);
        rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
    }
}

__global__ void bn_eval_stats_kernel(const float* rmean, const float* rvar, int C, float eps, float* mean, float* rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = rmean[c];
    rstd[c] = 1.0f / sqrtf(rvar[c] + eps);
}

__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* y, const float* mean, const float* rstd,
                                                         const float* gamma, const float* beta, const float* skip,
                                                         float* out, long long rows, int C, int F, int relu) {
    extern __shared__ float tab[];  // [4][C]
    for (int c = threadIdx.x; c < C; c += 256) {
        tab[c] = mean[c]; tab[C + c] = rstd[c]; tab[2 * C + c] = gamma[c]; tab[3 * C + c] = beta[c];
    }
    __syncthreads();
    const int CF = C * F;
    const long long n4 = rows * CF / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(y)[i];
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (skip) s = reinterpret_cast<const float4*>(skip)[i];
        const int j = (int)((i * 4) % CF);
        float in[4] = {v.x, v.y, v.z, v.w};
        float sk[4] = {s.x, s.y, s.z, s.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (j + e) / F;
            float t = (in[e] - tab[c]) * tab[C + c] * tab[2 * C + c] + tab[3 * C + c];
            if (relu) t = fmaxf(t, 0.f);
            o[e] = t + sk[e];
        }
        reinterpret_cast<float4*>(out)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// the 2-byte operand copy the next gate GEMM reads: bf16, or f16 (copy_f16: the single-pass f16 gate projection, cruse_gemm_f16_nt)
__device__ __forceinline__ void store_copy4(void* base, long long elem, const float o0, const float o1, const float o2, const float o3,
                                            const int copy_f16) {
    if (copy_f16) {
        typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_;
        const f16x4_ b = {(_Float16)o0, (_Float16)o1, (_Float16)o2, (_Float16)o3};
        *reinterpret_cast<f16x4_*>(reinterpret_cast<_Float16*>(base) + elem) = b;
    } else {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
        const bf16x4_ b = {(__bf16)o0, (__bf16)o1, (__bf16)o2, (__bf16)o3};
        *reinterpret_cast<bf16x4_*>(reinterpret_cast<__bf16*>(base) + elem) = b;
    }
}

// bn_finalize + bn_act_fwd in ONE launch: every block derives mean / rstd of the (<= 256) channels from the f64 batch sums
// itself; block 0 also publishes them (the backward pass reads them) and updates the running statistics.
__global__ __launch_bounds__(256) void bn_fin_act_fwd_kernel(const float* y, const double* sums, int nrep, double inv_count, double unb,
                                                             float eps, float momentum, const float* gamma,
                                                             const float* beta, const float* skip, float* out, __bf16* out_bf,
                                                             float* mean_o, float* rstd_o, float* rmean, float* rvar,
                                                             long long rows, int C, int F, int relu, int copy_f16) {
    extern __shared__ float tab[];  // [4][C]
    for (int c = threadIdx.x; c < C; c += 256) {
        double t1 = 0.0, t2 = 0.0;
        for (int r = 0; r < nrep; ++r) { t1 += sums[(long long)r * 2 * C + c]; t2 += sums[(long long)r * 2 * C + C + c]; }
        const double m = t1 * inv_count;
        double var = t2 * inv_count - m * m;
        if (var < 0.0) var = 0.0;
        const float mf = (float)m, rs = (float)(1.0 / sqrt(var + (double)eps));
        tab[c] = mf; tab[C + c] = rs; tab[2 * C + c] = gamma[c]; tab[3 * C + c] = beta[c];
        if (blockIdx.x == 0) {
            mean_o[c] = mf; rstd_o[c] = rs;
            if (rmean) {
                rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * m);
                rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * var * unb);
            }
        }
    }
    __syncthreads();
    const int CF = C * F;
    if (CF <= 64 * CPR_MAXG * 4) {
        // ONE WAVE PER FRAME ROW, float4 per lane, the lane's channel constants in registers (its columns never change across rows):
        // the grid-stride form below spends a runtime division and four LDS look-ups per ELEMENT and ran at 3.6 TB/s (36.6 us per
        // 65.7 MB level against 25 for the backward apply pass, which is written this way); all of a row's loads are issued first
        const int lane = threadIdx.x & 63, ng = CF >> 2;
        float pm[CPR_MAXG][4], pr[CPR_MAXG][4], pg[CPR_MAXG][4], pb[CPR_MAXG][4];
#pragma unroll
        for (int g = 0; g < CPR_MAXG; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = (lane + 64 * g) * 4 + e;
                const int c = col < CF ? col / F : 0;
                pm[g][e] = tab[c]; pr[g][e] = tab[C + c]; pg[g][e] = tab[2 * C + c]; pb[g][e] = tab[3 * C + c];
            }
        const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (long long)gridDim.x * 4;
        for (long long r = wave; r < rows; r += nwave) {
            float4 vv[CPR_MAXG], ss[CPR_MAXG];
            if (skip) {
#pragma unroll
                for (int g = 0; g < CPR_MAXG; ++g) {
                    const int q = lane + 64 * g;
                    if (q < ng) { vv[g] = *reinterpret_cast<const float4*>(y + r * CF + q * 4); ss[g] = *reinterpret_cast<const float4*>(skip + r * CF + q * 4); }
                }
            } else {
#pragma unroll
                for (int g = 0; g < CPR_MAXG; ++g) {
                    const int q = lane + 64 * g;
                    if (q < ng) { vv[g] = *reinterpret_cast<const float4*>(y + r * CF + q * 4); ss[g] = make_float4(0.f, 0.f, 0.f, 0.f); }
                }
            }
#pragma unroll
            for (int g = 0; g < CPR_MAXG; ++g) {
                const int q = lane + 64 * g;
                if (q < ng) {
                    const long long i = r * CF + q * 4;
                    const float in[4] = {vv[g].x, vv[g].y, vv[g].z, vv[g].w}, sk[4] = {ss[g].x, ss[g].y, ss[g].z, ss[g].w};
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = (in[e] - pm[g][e]) * pr[g][e] * pg[g][e] + pb[g][e];
                        if (relu) t = fmaxf(t, 0.f);
                        o[e] = t + sk[e];
                    }
                    *reinterpret_cast<float4*>(out + i) = make_float4(o[0], o[1], o[2], o[3]);
                    if (out_bf) store_copy4(out_bf, i, o[0], o[1], o[2], o[3], copy_f16);     // the operand copy the next gate GEMM reads (saves a cast pass)
                }
            }
        }
        return;
    }
    const long long n4 = rows * CF / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(y)[i];
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (skip) s = reinterpret_cast<const float4*>(skip)[i];
        const int j = (int)((i * 4) % CF);
        const float in[4] = {v.x, v.y, v.z, v.w};
        const float sk[4] = {s.x, s.y, s.z, s.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (j + e) / F;
            float t = (in[e] - tab[c]) * tab[C + c] * tab[2 * C + c] + tab[3 * C + c];
            if (relu) t = fmaxf(t, 0.f);
            o[e] = t + sk[e];
        }
        reinterpret_cast<float4*>(out)[i] = make_float4(o[0], o[1], o[2], o[3]);
        if (out_bf) store_copy4(out_bf, i * 4, o[0], o[1], o[2], o[3], copy_f16);
    }
}

__global__ __launch_bounds__(RED_THREADS) void bn_act_bwd_reduce_kernel(const float* dout, const float* y, const float* mean,
                                                                const float* rstd, const float* gamma,
                                                                const float* beta, long long rows, int C, int F,
                                                                int relu, double* sums) {
    // per-lane column constants (the lane's columns never change across rows)
    __shared__ float tab[4][256];
    float pm[CPR_MAXG][4], pr[CPR_MAXG][4], pg[CPR_MAXG][4], pb[CPR_MAXG][4];
    const int lane = threadIdx.x & 63;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        tab[0][c] = mean[c]; tab[1][c] = rstd[c]; tab[2][c] = gamma[c]; tab[3][c] = beta[c];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < CPR_MAXG; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = (lane + 64 * g) * 4 + e;
            const int c = col < C * F ? col / F : 0;
            pm[g][e] = tab[0][c]; pr[g][e] = tab[1][c]; pg[g][e] = tab[2][c]; pb[g][e] = tab[3][c];
        }
    channel_pair_reduce(rows, C, F, sums, [&](long long i, int g, float (&v)[4], float (&v2)[4]) {
        const float4 yy = *reinterpret_cast<const float4*>(y + i);
        const float4 dd = *reinterpret_cast<const float4*>(dout + i);
        const float yv[4] = {yy.x, yy.y, yy.z, yy.w}, dv[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (yv[e] - pm[g][e]) * pr[g][e];
            float gr = dv[e];
            if (relu && !(xh * pg[g][e] + pb[g][e] > 0.f)) gr = 0.f;
            v[e] = gr; v2[e] = gr * xh;
        }
    });
}

// wave-per-row like channel_pair_reduce: a lane's columns are fixed, so its six per-channel constants live in
// registers.  The gradient of the conv bias that feeds this BN, sum(dy) per channel, needs no pass over the data:
// it is gamma * rstd * (sum g - count * mean g - mean(g xhat) * sum xhat), i.e. exactly 0 with batch statistics
// (sum xhat = 0; autograd produces rounding noise there) and gamma * rstd * sum g with running statistics.
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const float* dout, const float* y, const float* mean,
                                                               const float* rstd, const float* gamma,
                                                               const float* beta, const double* sums, int nrep, long long rows,
                                                               int C, int F, int relu, int training, float* dy, int dy_bf16,
                                                               int dout_bf16, float* dgamma, float* dbeta, float* dbias) {
    __shared__ float tab[6][256];
    const double cnt = (double)rows * F;
    const int tid = threadIdx.x, lane = tid & 63;
    const int CF = C * F, ng = CF >> 2;
    for (int c = tid; c < C; c += 256) {
        double sg = 0.0, sgx = 0.0;                       // the statistic is the sum over the replicas
        for (int r = 0; r < nrep; ++r) { sg += sums[(size_t)r * 2 * C + c]; sgx += sums[(size_t)r * 2 * C + C + c]; }
        tab[0][c] = mean[c]; tab[1][c] = rstd[c]; tab[2][c] = gamma[c]; tab[3][c] = beta[c];
        tab[4][c] = training ? (float)(sg / cnt) : 0.f;
        tab[5][c] = training ? (float)(sgx / cnt) : 0.f;
        if (blockIdx.x == 0) {
            if (dgamma) dgamma[c] += (float)sgx;
            if (dbeta) dbeta[c] += (float)sg;
            if (dbias && !training) dbias[c] += gamma[c] * rstd[c] * (float)sg;
        }
    }
    __syncthreads();
    float pm[CPR_MAXG][4], pr[CPR_MAXG][4], pg[CPR_MAXG][4], pb[CPR_MAXG][4], p1[CPR_MAXG][4], p2[CPR_MAXG][4];
#pragma unroll
    for (int g = 0; g < CPR_MAXG; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = (lane + 64 * g) * 4 + e;
            const int c = col < CF ? col / F : 0;
            pm[g][e] = tab[0][c]; pr[g][e] = tab[1][c]; pg[g][e] = tab[2][c]; pb[g][e] = tab[3][c];
            p1[g][e] = tab[4][c]; p2[g][e] = tab[5][c];
        }
    const long long wave = (long long)blockIdx.x * 4 + (tid >> 6), nwave = (long long)gridDim.x * 4;
    for (long long r = wave; r < rows; r += nwave) {
        // all of the row's loads first: the stores below may alias as far as the compiler knows, and would otherwise
        // serialise load -> compute -> store per float4 group
        float4 vv[CPR_MAXG], dv[CPR_MAXG];
        if (dout_bf16) {                                   // (dtype branch outside the group loop: the loads stay one batch)
#pragma unroll
            for (int g = 0; g < CPR_MAXG; ++g) {
                const int q = lane + 64 * g;
                if (q < ng) {
                    const long long i = r * CF + q * 4;
                    vv[g] = *reinterpret_cast<const float4*>(y + i);
                    const float2 w2 = *reinterpret_cast<const float2*>(reinterpret_cast<const __bf16*>(dout) + i);
                    const unsigned w0 = __float_as_uint(w2.x), w1 = __float_as_uint(w2.y);
                    dv[g] = make_float4(__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u), __uint_as_float(w1 << 16),
                                        __uint_as_float(w1 & 0xffff0000u));
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < CPR_MAXG; ++g) {
                const int q = lane + 64 * g;
                if (q < ng) {
                    const long long i = r * CF + q * 4;
                    vv[g] = *reinterpret_cast<const float4*>(y + i);
                    dv[g] = *reinterpret_cast<const float4*>(dout + i);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < CPR_MAXG; ++g) {
            const int q = lane + 64 * g;
            if (q < ng) {
                const long long i = r * CF + q * 4;
                const float in[4] = {vv[g].x, vv[g].y, vv[g].z, vv[g].w}, dd[4] = {dv[g].x, dv[g].y, dv[g].z, dv[g].w};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (in[e] - pm[g][e]) * pr[g][e];
                    float gr = dd[e];
                    if (relu && !(xh * pg[g][e] + pb[g][e] > 0.f)) gr = 0.f;
                    o[e] = pg[g][e] * pr[g][e] * (gr - p1[g][e] - xh * p2[g][e]);
                }
                if (dy_bf16) {                             // a backward-only tensor: its consumers round it to bf16 anyway
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
                    bf16x4_ h;
                    h[0] = (__bf16)o[0]; h[1] = (__bf16)o[1]; h[2] = (__bf16)o[2]; h[3] = (__bf16)o[3];
                    *reinterpret_cast<bf16x4_*>(reinterpret_cast<__bf16*>(dy) + i) = h;
                } else {
                    *reinterpret_cast<float4*>(dy + i) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ LayerNorm ---
constexpr int LN_MAXE = 16;  // H <= 1024

__device__ __forceinline__ int ln_perm(int c, int H, int g) {
    if (g <= 1) return c;
    const int Hg = H / g;
    return (c % Hg) * g + c / Hg;
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* x, const float* gamma, const float* beta,
                                                     const float* res, float* y, float* mean, float* rstd,
                                                     long long rows, int H, int g, float eps) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nwave = (long long)gridDim.x * 4;
    for (long long r = wave; r < rows; r += nwave) {
        float v[LN_MAXE];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < LN_MAXE; ++e) {
            const int c = lane + 64 * e;
            v[e] = c < H ? x[r * H + c] : 0.f;
            s += v[e];
        }
        const float m = wave_sum(s) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < LN_MAXE; ++e) {
            const int c = lane + 64 * e;
            const float d = c < H ? v[e] - m : 0.f;
            q += d * d;
        }
        const float rs = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
        if (lane == 0) { if (mean) mean[r] = m; if (rstd) rstd[r] = rs; }
#pragma unroll
        for (int e = 0; e < LN_MAXE; ++e) {
            const int c = lane + 64 * e;
            if (c < H) {
                const int p = ln_perm(c, H, g);
                float o = (v[e] - m) * rs * gamma[p] + beta[p];
                if (res) o += res[r * H + p];
                y[r * H + p] = o;
            }
        }
    }
}

__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* dy, const float* x, const float* mean,
                                                     const float* rstd, const float* gamma, long long rows, int H,
                                                     int g, float* dx, float* dgamma, float* dbeta) {
    __shared__ float sdg[1024], sdb[1024];
    const int lane = threadIdx.x & 63;
    for (int c = threadIdx.x; c < H; c += 256) { sdg[c] = 0.f; sdb[c] = 0.f; }
    __syncthreads();
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nwave = (long long)gridDim.x * 4;
    float adg[LN_MAXE], adb[LN_MAXE];
    int perm[LN_MAXE];
    float gm[LN_MAXE];
#pragma unroll
    for (int e = 0; e < LN_MAXE; ++e) {
        adg[e] = 0.f; adb[e] = 0.f;
        const int c = lane + 64 * e;
        perm[e] = c < H ? ln_perm(c, H, g) : 0;
        gm[e] = c < H ? gamma[perm[e]] : 0.f;
    }
    for (long long r = wave; r < rows; r += nwave) {
        const float m = mean[r], rs = rstd[r];
        float xh[LN_MAXE], gd[LN_MAXE];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < LN_MAXE; ++e) {
            const int c = lane + 64 * e;
            if (c < H) {
                const float d = dy[r * H + perm[e]];
                xh[e] = (x[r * H + c] - m) * rs;
                gd[e] = d * gm[e];
                adg[e] += d * xh[e];
                adb[e] += d;
                s1 += gd[e];
                s2 += gd[e] * xh[e];
            } else { xh[e] = 0.f; gd[e] = 0.f; }
        }
        const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H;
#pragma unroll
        for (int e = 0; e < LN_MAXE; ++e) {
            const int c = lane + 64 * e;
            if (c < H) dx[r * H + c] = rs * (gd[e] - m1 - xh[e] * m2);
        }
    }
#pragma unroll
    for (int e = 0; e < LN_MAXE; ++e) {
        const int c = lane + 64 * e;
        if (c < H) { atomicAdd(&sdg[perm[e]], adg[e]); atomicAdd(&sdb[perm[e]], adb[e]); }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) {
        if (dgamma) atomicAdd(&dgamma[c], sdg[c]);
        if (dbeta) atomicAdd(&dbeta[c], sdb[c]);
    }
}

// g == 1, H % 4 == 0 forms: a lane owns float4 groups lane, lane+64, ... of the row (16-byte accesses) and a wave
// works on NR rows at once so that their loads and the two reductions of one row overlap the other's.
constexpr int LNV_MAXQ = 4;   // H <= 1024
constexpr int LNV_NR = 2;     // forward; the backward kernel keeps one row per wave in flight (register budget)

struct RowSeg { int len; long long stride, off; };       // logical row r -> (r / len) * stride + off + r % len; len == 0: identity
__device__ __forceinline__ long long seg_row(const RowSeg& sg, long long r) {
    if (sg.len == 0) return r;
    const long long q = r / sg.len;
    return q * sg.stride + sg.off + (r - q * sg.len);
}

__global__ __launch_bounds__(256) void ln_fwd_vec_kernel(const float* x, const float* gamma, const float* beta,
                                                         const float* res, float* y, __bf16* y_bf, float* mean, float* rstd,
                                                         long long rows, int H, float eps, RowSeg sg, int copy_f16) {
    const int lane = threadIdx.x & 63, nq = H >> 2;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nwave = (long long)gridDim.x * 4;
    float4 gm[LNV_MAXQ], bt[LNV_MAXQ];
#pragma unroll
    for (int e = 0; e < LNV_MAXQ; ++e) {
        const int q = lane + 64 * e;
        gm[e] = q < nq ? reinterpret_cast<const float4*>(gamma)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        bt[e] = q < nq ? reinterpret_cast<const float4*>(beta)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long long r0 = wave * LNV_NR; r0 < rows; r0 += nwave * LNV_NR) {
        float4 v[LNV_NR][LNV_MAXQ], rv[LNV_NR][LNV_MAXQ];
        float s[LNV_NR], m[LNV_NR], qq[LNV_NR];
#pragma unroll
        for (int k = 0; k < LNV_NR; ++k) {
            const long long rl = r0 + k, r = seg_row(sg, rl < rows ? rl : rows - 1);
            s[k] = 0.f;
#pragma unroll
            for (int e = 0; e < LNV_MAXQ; ++e) {
                const int q = lane + 64 * e;
                const bool ok = q < nq && rl < rows;
                v[k][e] = ok ? reinterpret_cast<const float4*>(x + r * H)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
                rv[k][e] = (ok && res) ? reinterpret_cast<const float4*>(res + r * H)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
                s[k] += (v[k][e].x + v[k][e].y) + (v[k][e].z + v[k][e].w);
            }
        }
#pragma unroll
        for (int k = 0; k < LNV_NR; ++k) m[k] = wave_sum(s[k]) / (float)H;
#pragma unroll
        for (int k = 0; k < LNV_NR; ++k) {
            qq[k] = 0.f;
#pragma unroll
            for (int e = 0; e < LNV_MAXQ; ++e) {
                if (lane + 64 * e < nq) {
                    const float a = v[k][e].x - m[k], b = v[k][e].y - m[k], c = v[k][e].z - m[k], d = v[k][e].w - m[k];
                    qq[k] += (a * a + b * b) + (c * c + d * d);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < LNV_NR; ++k) {
            const float rs = 1.0f / sqrtf(wave_sum(qq[k]) / (float)H + eps);
            if (r0 + k >= rows) continue;
            const long long r = seg_row(sg, r0 + k);
            if (lane == 0) { if (mean) mean[r] = m[k]; if (rstd) rstd[r] = rs; }
#pragma unroll
            for (int e = 0; e < LNV_MAXQ; ++e) {
                const int q = lane + 64 * e;
                if (q < nq) {
                    float4 o;
                    o.x = (v[k][e].x - m[k]) * rs * gm[e].x + bt[e].x + rv[k][e].x;
                    o.y = (v[k][e].y - m[k]) * rs * gm[e].y + bt[e].y + rv[k][e].y;
                    o.z = (v[k][e].z - m[k]) * rs * gm[e].z + bt[e].z + rv[k][e].z;
                    o.w = (v[k][e].w - m[k]) * rs * gm[e].w + bt[e].w + rv[k][e].w;
                    reinterpret_cast<float4*>(y + r * H)[q] = o;
                    if (y_bf) store_copy4(y_bf, r * H + 4 * q, o.x, o.y, o.z, o.w, copy_f16);
                }
            }
        }
    }
}

// NQ float4 groups per lane (H <= 256 * NQ), NR rows per wave in flight: (3, 2) for the step's H = 640 -- with the
// generic (4, 1) a wave had one row's loads outstanding at a time and the pass ran at 3.5 TB/s
template <int NQ, int LNB_NR>
__global__ __launch_bounds__(256) void ln_bwd_vec_kernel(const float* dy, const float* x, const float* mean,
                                                         const float* rstd, const float* gamma, long long rows, int H,
                                                         float* dx, float* dgamma, float* dbeta, RowSeg sg) {
    __shared__ float sdg[1024], sdb[1024];
    const int lane = threadIdx.x & 63, nq = H >> 2;
    for (int c = threadIdx.x; c < H; c += blockDim.x) { sdg[c] = 0.f; sdb[c] = 0.f; }
    __syncthreads();
    const int nw = blockDim.x >> 6;
    const long long wave = (long long)blockIdx.x * nw + (threadIdx.x >> 6);
    const long long nwave = (long long)gridDim.x * nw;
    float4 gm[NQ], adg[NQ], adb[NQ];
#pragma unroll
    for (int e = 0; e < NQ; ++e) {
        const int q = lane + 64 * e;
        gm[e] = q < nq ? reinterpret_cast<const float4*>(gamma)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        adg[e] = make_float4(0.f, 0.f, 0.f, 0.f); adb[e] = adg[e];
    }
    for (long long r0 = wave * LNB_NR; r0 < rows; r0 += nwave * LNB_NR) {
        float4 xh[LNB_NR][NQ], gd[LNB_NR][NQ];
        float s1[LNB_NR], s2[LNB_NR], rsv[LNB_NR];
#pragma unroll
        for (int k = 0; k < LNB_NR; ++k) {
            const bool rok = r0 + k < rows;
            const long long r = seg_row(sg, rok ? r0 + k : rows - 1);
            const float m = rok ? mean[r] : 0.f;
            rsv[k] = rok ? rstd[r] : 0.f;
            s1[k] = 0.f; s2[k] = 0.f;
#pragma unroll
            for (int e = 0; e < NQ; ++e) {
                const int q = lane + 64 * e;
                if (q < nq && rok) {
                    const float4 d = reinterpret_cast<const float4*>(dy + r * H)[q];
                    const float4 xv = reinterpret_cast<const float4*>(x + r * H)[q];
                    float4 h, gdd;
                    h.x = (xv.x - m) * rsv[k]; h.y = (xv.y - m) * rsv[k]; h.z = (xv.z - m) * rsv[k]; h.w = (xv.w - m) * rsv[k];
                    gdd.x = d.x * gm[e].x; gdd.y = d.y * gm[e].y; gdd.z = d.z * gm[e].z; gdd.w = d.w * gm[e].w;
                    adg[e].x += d.x * h.x; adg[e].y += d.y * h.y; adg[e].z += d.z * h.z; adg[e].w += d.w * h.w;
                    adb[e].x += d.x; adb[e].y += d.y; adb[e].z += d.z; adb[e].w += d.w;
                    s1[k] += (gdd.x + gdd.y) + (gdd.z + gdd.w);
                    s2[k] += (gdd.x * h.x + gdd.y * h.y) + (gdd.z * h.z + gdd.w * h.w);
                    xh[k][e] = h; gd[k][e] = gdd;
                } else {
                    xh[k][e] = make_float4(0.f, 0.f, 0.f, 0.f); gd[k][e] = xh[k][e];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < LNB_NR; ++k) { s1[k] = wave_sum(s1[k]) / (float)H; s2[k] = wave_sum(s2[k]) / (float)H; }
#pragma unroll
        for (int k = 0; k < LNB_NR; ++k) {
            if (r0 + k >= rows) continue;
            const long long r = seg_row(sg, r0 + k);
#pragma unroll
            for (int e = 0; e < NQ; ++e) {
                const int q = lane + 64 * e;
                if (q < nq) {
                    float4 o;
                    o.x = rsv[k] * (gd[k][e].x - s1[k] - xh[k][e].x * s2[k]);
                    o.y = rsv[k] * (gd[k][e].y - s1[k] - xh[k][e].y * s2[k]);
                    o.z = rsv[k] * (gd[k][e].z - s1[k] - xh[k][e].z * s2[k]);
                    o.w = rsv[k] * (gd[k][e].w - s1[k] - xh[k][e].w * s2[k]);
                    reinterpret_cast<float4*>(dx + r * H)[q] = o;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < NQ; ++e) {
        const int q = lane + 64 * e;
        if (q < nq) {
            atomicAdd(&sdg[q * 4], adg[e].x); atomicAdd(&sdg[q * 4 + 1], adg[e].y);
            atomicAdd(&sdg[q * 4 + 2], adg[e].z); atomicAdd(&sdg[q * 4 + 3], adg[e].w);
            atomicAdd(&sdb[q * 4], adb[e].x); atomicAdd(&sdb[q * 4 + 1], adb[e].y);
            atomicAdd(&sdb[q * 4 + 2], adb[e].z); atomicAdd(&sdb[q * 4 + 3], adb[e].w);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        if (dgamma) atomicAdd(&dgamma[c], sdg[c]);
        if (dbeta) atomicAdd(&dbeta[c], sdb[c]);
    }
}

// ------------------------------------------------------------------ mask + loss ---
__global__ __launch_bounds__(256) void mask_loss_kernel(const float* mask, const float* nre, const float* nim,
                                                        const float* cmag, long long rows, int Fn, int Fs,
                                                        float alpha, float beta, double* loss_sum, float* dmask,
                                                        float* dlogit, float* est_re, float* est_im) {
    __shared__ double sred[4];
    const long long n = rows * Fs;
    const float inv_n = (float)(1.0 / (double)n);
    const float inv_ln10 = 0.43429448190325176f;
    double acc = 0.0;
    float part = 0.f;
    int cnt = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / Fs;
        const int f = (int)(i % Fs);
        const float re = nre[i], im = nim[i];
        const float m = f < Fn ? mask[r * Fn + f] : 0.f;
        const float er = m * re, ei = m * im;
        const float mag_est = sqrtf(er * er + ei * ei);
        const float mag_ref = cmag[i];
        const float mag_unp = sqrtf(re * re + im * im);
        const float iam = mag_ref / mag_unp;
        const float w = expf(alpha / (beta + iam));
        const float d = log10f(mag_est + 1.f) - log10f(mag_ref + 1.f);
        part += w * fabsf(d);
        if (++cnt == 32) { acc += part; part = 0.f; cnt = 0; }
        if (est_re) est_re[i] = er;
        if (est_im) est_im[i] = ei;
        if (f < Fn && (dmask || dlogit)) {
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            const float dm = w * sgn * inv_ln10 / (mag_est + 1.f) * mag_unp * inv_n;
            if (dmask) dmask[r * Fn + f] = dm;
            if (dlogit) dlogit[r * Fn + f] = dm * m * (1.f - m);
        }
    }
    acc += part;
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, sred[0] + sred[1] + sred[2] + sred[3]);
}

__global__ void sigmoid_bwd_kernel(const float* dmask, const float* mask, float* dlogit, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float m = mask[i];
        dlogit[i] = dmask[i] * m * (1.f - m);
    }
}

__global__ void axpby_kernel(float* out, const float* x, const float* y, float a, float b, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}

// host bookkeeping that must not leave the stream: f64 accumulation of per-step loss sums, BatchNorm batch counters
__global__ void accum_f64_kernel(double* acc, const double* x, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += x[i];
}
// Per-step health word of the training step (engine.py): LATCHES the sticky GRU status word into health[0] and clears it
// (one transient hand-off time-out then costs exactly the step it happened in), health[1] = the loss sum is not finite;
// the running loss is accumulated with the step's own normalisation (steps of different shapes share one accumulator).
__global__ void step_health_kernel(unsigned* gru_status, const double* loss_sum, unsigned* health, double* loss_acc,
                                   double loss_scale) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned to = 0u;
    if (gru_status) {
        to = __hip_atomic_load(gru_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ? 1u : 0u;
        if (to) __hip_atomic_store(gru_status, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const double l = loss_sum ? *loss_sum : 0.0;
    health[0] = to;
    health[1] = isfinite(l) ? 0u : 1u;
    if (loss_acc && isfinite(l)) *loss_acc += l * loss_scale;
}
struct CounterPtrs { long long* p[32]; };
__global__ void counters_add_kernel(CounterPtrs c, int n, long long v) {
    const int i = threadIdx.x;
    if (i < n) *c.p[i] += v;
}

// guard words (all optional): the step is SKIPPED -- parameters and moments untouched, *skipped += 1 -- when
// *skip_flag != 0 (a GRU hand-off timed out: the gradients are garbage), when *loss_check is not finite, or when the
// gradient norm is not finite.  gsumsq (sum of squares of g BEFORE grad_scale) with max_norm > 0 applies
// torch.nn.utils.clip_grad_norm_'s coefficient min(1, max_norm / (norm + 1e-6)) on top of grad_scale.
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2,
                            float eps, float wd, int step, float gscale, float max_norm,
                            const double* gsumsq, const unsigned* skip_flag, int n_skip_words, const double* loss_check,
                            unsigned* skipped, const double* loss_sum, double loss_scale, double* loss_acc) {
    // Adam's step counts the APPLIED steps: skipped[0] is read before this launch can change it (only the skip path below does)
    const int step_eff = max(1, step - (skipped ? (int)*skipped : 0));
    const float bc1 = (float)(1.0 - pow((double)b1, (double)step_eff));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)step_eff));
    bool skip = false;
    for (int i = 0; i < n_skip_words; ++i)
        if (skip_flag[i] != 0u) skip = true;
    if (loss_check && !isfinite(*loss_check)) skip = true;
    if (gsumsq) {
        const double norm = sqrt(*gsumsq) * (double)gscale;
        if (!isfinite(norm)) skip = true;
        else if (max_norm > 0.f) {
            const double coef = (double)max_norm / (norm + 1e-6);
            if (coef < 1.0) gscale = (float)((double)gscale * coef);
        }
    }
    if (skip) {
        if (skipped && blockIdx.x == 0 && threadIdx.x == 0) {
            atomicAdd(skipped, 1u);
            for (int i = 0; i < n_skip_words && n_skip_words > 1; ++i)          // per-reason counters behind the total
                if (skip_flag[i] != 0u) atomicAdd(skipped + 1 + i, 1u);
        }
        return;
    }
    if (loss_acc && loss_sum && blockIdx.x == 0 && threadIdx.x == 0) { loss_acc[0] += *loss_sum * loss_scale; loss_acc[1] += 1.0; }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float gi = g[i] * gscale;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

// sum of squares of a flat f32 buffer in f64 (the total gradient norm of clip_grad_norm_): float4 streams,
// one same-address f64 atomic per 1024-thread block
__global__ __launch_bounds__(1024) void sumsq_kernel(const float* x, long long n, double* out) {
    __shared__ double part[16];
    double acc = 0.0;
    const long long n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 a = x4[i];
        acc += (double)(a.x * a.x + a.y * a.y) + (double)(a.z * a.z + a.w * a.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float a = x[(n4 << 2) + threadIdx.x]; acc += (double)a * a; }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        double t = threadIdx.x < (blockDim.x >> 6) ? part[threadIdx.x] : 0.0;
        t = wave_sum_d(t);
        if (threadIdx.x == 0) atomicAdd(out, t);
    }
}

inline int grid_for(long long n, int per_block, int cap = 4096) {
    long long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace

extern "C" int cruse_cast_bf16_split(const float* x, void* y, void* y_lo, long long n, void* stream);

#define ST(s) ((hipStream_t)(s))

extern "C" int cruse_bn_stats(const float* y, long long rows, int C, int F, double* sums, int zeroed, void* stream) {
    CRUSE_REQUIRE(rows > 0 && C > 0 && C <= 256 && F > 0, CRUSE_E_SHAPE, "bn_stats: bad shape rows=%lld C=%d F=%d", rows, C, F);
    CRUSE_REQUIRE((C * F) % 4 == 0 && C * F <= 768, CRUSE_E_SHAPE, "bn_stats: C*F=%d must be a multiple of 4 and <= 768", C * F);
    if (!zeroed) { int zrc = cruse_zero_async(sums, 2 * C * sizeof(double), ST(stream), "bn_stats memset"); if (zrc) return zrc; }
    hipLaunchKernelGGL(bn_stats_kernel, dim3(grid_for(rows, 64, 256)), dim3(RED_THREADS), 0, ST(stream), y, rows, C, F, sums);
    CRUSE_LAUNCH_CHECK("bn_stats");
    return CRUSE_OK;
}

extern "C" int cruse_bn_finalize(const double* sums, long long count, int C, float eps, float momentum,
                                 float* mean, float* rstd, float* running_mean, float* running_var, void* stream) {
    CRUSE_REQUIRE(count > 0 && C > 0, CRUSE_E_SHAPE, "bn_finalize: bad shape");
    CRUSE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), CRUSE_E_SHAPE, "bn_finalize: running stats");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, ST(stream), sums, count, C, eps, momentum,
                       mean, rstd, running_mean, running_var);
    CRUSE_LAUNCH_CHECK("bn_finalize");
    return CRUSE_OK;
}

extern "C" int cruse_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps,
                                   float* mean, float* rstd, void* stream) {
    CRUSE_REQUIRE(C > 0, CRUSE_E_SHAPE, "bn_eval_stats: bad shape");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(C, 64)), dim3(64), 0, ST(stream), running_mean, running_var, C,
                       eps, mean, rstd);
    CRUSE_LAUNCH_CHECK("bn_eval_stats");
    return CRUSE_OK;
}

extern "C" int cruse_bn_act_fwd(const float* y, const float* mean, const float* rstd, const float* gamma,
                                const float* beta, const float* skip, float* out,
                                long long rows, int C, int F, int relu, void* stream) {
    CRUSE_REQUIRE(rows > 0 && C > 0 && F > 0, CRUSE_E_SHAPE, "bn_act_fwd: bad shape");
    CRUSE_REQUIRE((C * F) % 4 == 0, CRUSE_E_ALIGN, "bn_act_fwd: C*F=%d must be a multiple of 4", C * F);
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(grid_for(rows * C * F / 4, 1024)), dim3(256), 4 * C * sizeof(float),
                       ST(stream), y, mean, rstd, gamma, beta, skip, out, rows, C, F, relu);
    CRUSE_LAUNCH_CHECK("bn_act_fwd");
    return CRUSE_OK;
}

extern "C" int cruse_bn_finalize_act_fwd_c(const float* y, const double* sums, int sum_replicas, long long count, float eps, float momentum,
                                           const float* gamma, const float* beta, const float* skip, float* out, void* out_copy, int copy_dtype,
                                           float* mean, float* rstd, float* running_mean, float* running_var,
                                           long long rows, int C, int F, int relu, void* stream);

extern "C" int cruse_bn_finalize_act_fwd(const float* y, const double* sums, int sum_replicas, long long count, float eps, float momentum,
                                         const float* gamma, const float* beta, const float* skip, float* out, void* out_bf16,
                                         float* mean, float* rstd, float* running_mean, float* running_var,
                                         long long rows, int C, int F, int relu, void* stream) {
    return cruse_bn_finalize_act_fwd_c(y, sums, sum_replicas, count, eps, momentum, gamma, beta, skip, out, out_bf16, CRUSE_DT_BF16, mean, rstd,
                                       running_mean, running_var, rows, C, F, relu, stream);
}

extern "C" int cruse_bn_finalize_act_fwd_c(const float* y, const double* sums, int sum_replicas, long long count, float eps, float momentum,
                                           const float* gamma, const float* beta, const float* skip, float* out, void* out_bf16, int copy_dtype,
                                           float* mean, float* rstd, float* running_mean, float* running_var,
                                           long long rows, int C, int F, int relu, void* stream) {
    CRUSE_REQUIRE(copy_dtype == CRUSE_DT_BF16 || copy_dtype == CRUSE_DT_F16, CRUSE_E_DTYPE, "bn_finalize_act_fwd: copy_dtype %d (bf16 or f16)", copy_dtype);
    CRUSE_REQUIRE(rows > 0 && C > 0 && C <= 256 && F > 0 && count > 0 && sum_replicas >= 1, CRUSE_E_SHAPE, "bn_finalize_act_fwd: bad shape");
    CRUSE_REQUIRE((C * F) % 4 == 0, CRUSE_E_ALIGN, "bn_finalize_act_fwd: C*F=%d must be a multiple of 4", C * F);
    CRUSE_REQUIRE((running_mean == nullptr) == (running_var == nullptr) && mean && rstd, CRUSE_E_SHAPE, "bn_finalize_act_fwd: statistics");
    const double unb = count > 1 ? (double)count / (double)(count - 1) : 1.0;
    const bool rowwise = C * F <= 64 * CPR_MAXG * 4 && (C * F) % 4 == 0;
    hipLaunchKernelGGL(bn_fin_act_fwd_kernel, dim3(rowwise ? grid_for(rows, 8, 2048) : grid_for(rows * C * F / 4, 1024)), dim3(256), 4 * C * sizeof(float), ST(stream),
                       y, sums, sum_replicas, 1.0 / (double)count, unb, eps, momentum, gamma, beta, skip, out, (__bf16*)out_bf16, mean, rstd, running_mean,
                       running_var, rows, C, F, relu, copy_dtype == CRUSE_DT_F16 ? 1 : 0);
    CRUSE_LAUNCH_CHECK("bn_finalize_act_fwd");
    return CRUSE_OK;
}

extern "C" int cruse_bn_act_bwd_reduce(const float* dout, const float* y, const float* mean, const float* rstd,
                                       const float* gamma, const float* beta, long long rows, int C, int F,
                                       int relu, double* sums, int zeroed, void* stream) {
    CRUSE_REQUIRE(rows > 0 && C > 0 && C <= 256 && F > 0, CRUSE_E_SHAPE, "bn_act_bwd_reduce: bad shape");
    CRUSE_REQUIRE((C * F) % 4 == 0 && C * F <= 768, CRUSE_E_SHAPE, "bn_act_bwd_reduce: C*F=%d must be a multiple of 4 and <= 768", C * F);
    if (!zeroed) { int zrc = cruse_zero_async(sums, 2 * C * sizeof(double), ST(stream), "bn_act_bwd_reduce memset"); if (zrc) return zrc; }
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3(grid_for(rows, 64, 256)), dim3(RED_THREADS), 0, ST(stream), dout, y, mean,
                       rstd, gamma, beta, rows, C, F, relu, sums);
    CRUSE_LAUNCH_CHECK("bn_act_bwd_reduce");
    return CRUSE_OK;
}

extern "C" int cruse_bn_act_bwd_apply(const float* dout, const float* y, const float* mean, const float* rstd,
                                      const float* gamma, const float* beta, const double* sums, int sum_replicas,
                                      long long rows, int C, int F, int relu, int training, int dout_dtype,
                                      void* dy, int dy_dtype, float* dgamma, float* dbeta, float* dbias, void* stream) {
    CRUSE_REQUIRE((dy_dtype == CRUSE_DT_F32 || dy_dtype == CRUSE_DT_BF16) && (dout_dtype == CRUSE_DT_F32 || dout_dtype == CRUSE_DT_BF16),
                  CRUSE_E_DTYPE, "bn_act_bwd_apply: dout_dtype %d / dy_dtype %d (f32 or bf16)", dout_dtype, dy_dtype);
    CRUSE_REQUIRE(rows > 0 && C > 0 && C <= 256 && F > 0 && sum_replicas >= 1, CRUSE_E_SHAPE, "bn_act_bwd_apply: bad shape");
    CRUSE_REQUIRE((C * F) % 4 == 0 && C * F <= 768, CRUSE_E_SHAPE,
                  "bn_act_bwd_apply: C*F=%d must be a multiple of 4 and <= 768", C * F);
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(grid_for(rows, 8, 2048)), dim3(256), 0, ST(stream), dout, y,
                       mean, rstd, gamma, beta, sums, sum_replicas, rows, C, F, relu, training, (float*)dy,
                       dy_dtype == CRUSE_DT_BF16 ? 1 : 0, dout_dtype == CRUSE_DT_BF16 ? 1 : 0, dgamma, dbeta, dbias);
    CRUSE_LAUNCH_CHECK("bn_act_bwd_apply");
    return CRUSE_OK;
}

extern "C" int cruse_ln_fwd_c(const float* x, const float* gamma, const float* beta, const float* res,
                              float* y, void* y_copy, int copy_dtype, float* mean, float* rstd, long long rows, int H, int interleave_g,
                              float eps, int seg_len, long long seg_stride, long long seg_off, void* stream);

extern "C" int cruse_ln_fwd(const float* x, const float* gamma, const float* beta, const float* res,
                            float* y, void* y_bf16, float* mean, float* rstd, long long rows, int H, int interleave_g,
                            float eps, int seg_len, long long seg_stride, long long seg_off, void* stream) {
    return cruse_ln_fwd_c(x, gamma, beta, res, y, y_bf16, CRUSE_DT_BF16, mean, rstd, rows, H, interleave_g, eps, seg_len, seg_stride, seg_off,
                          stream);
}

extern "C" int cruse_ln_fwd_c(const float* x, const float* gamma, const float* beta, const float* res,
                              float* y, void* y_bf16, int copy_dtype, float* mean, float* rstd, long long rows, int H, int interleave_g,
                              float eps, int seg_len, long long seg_stride, long long seg_off, void* stream) {
    CRUSE_REQUIRE(copy_dtype == CRUSE_DT_BF16 || copy_dtype == CRUSE_DT_F16, CRUSE_E_DTYPE, "ln_fwd: copy_dtype %d (bf16 or f16)", copy_dtype);
    CRUSE_REQUIRE(seg_len >= 0 && (seg_len == 0 || (seg_stride >= seg_len && seg_off >= 0 && rows % seg_len == 0)), CRUSE_E_SHAPE,
                  "ln_fwd: bad row segments (len %d stride %lld off %lld, rows %lld)", seg_len, seg_stride, seg_off, rows);
    const RowSeg sg = {seg_len, seg_stride, seg_off};
    CRUSE_REQUIRE(rows > 0 && H > 0 && H <= 64 * LN_MAXE, CRUSE_E_SHAPE, "ln_fwd: H=%d must be in 1..%d", H, 64 * LN_MAXE);
    CRUSE_REQUIRE(interleave_g >= 1 && H % interleave_g == 0, CRUSE_E_SHAPE, "ln_fwd: groups=%d must divide H=%d", interleave_g, H);
    const bool vec = interleave_g == 1 && H % 4 == 0 && H <= 256 * LNV_MAXQ &&
                     ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)res) & 15) == 0);
    const bool bf_inline = vec && y_bf16 && (((uintptr_t)y_bf16 & 7) == 0);
    if (vec)
        hipLaunchKernelGGL(ln_fwd_vec_kernel, dim3(grid_for(rows, 16, 2048)), dim3(256), 0, ST(stream), x, gamma, beta,
                           res, y, bf_inline ? (__bf16*)y_bf16 : nullptr, mean, rstd, rows, H, eps, sg, copy_dtype == CRUSE_DT_F16 ? 1 : 0);
    else {
        CRUSE_REQUIRE(seg_len == 0, CRUSE_E_SHAPE, "ln_fwd: row segments need the vector form (one group, H %% 4 == 0, aligned)");
        hipLaunchKernelGGL(ln_fwd_kernel, dim3(grid_for(rows, 4, 2048)), dim3(256), 0, ST(stream), x, gamma, beta, res, y,
                           mean, rstd, rows, H, interleave_g, eps);
    }
    CRUSE_LAUNCH_CHECK("ln_fwd");
    CRUSE_REQUIRE(seg_len == 0 || !y_bf16 || bf_inline, CRUSE_E_ALIGN, "ln_fwd: row segments need an 8-byte aligned bf16 copy");
    CRUSE_REQUIRE(copy_dtype == CRUSE_DT_BF16 || !y_bf16 || bf_inline, CRUSE_E_DTYPE, "ln_fwd: the f16 copy needs the vector form (one group)");
    if (y_bf16 && !bf_inline) return cruse_cast_bf16_split(y, y_bf16, nullptr, rows * H, stream);     // interleaved form: one more pass
    return CRUSE_OK;
}

static int lnb_grid() { return cruse_opt("lnb_grid", 512); }

static int ln_bwd_impl(const float* dy, const float* x, const float* mean, const float* rstd,
                       const float* gamma, long long rows, int H, int interleave_g,
                       float* dx, float* dgamma, float* dbeta, int seg_len, long long seg_stride, long long seg_off, void* stream) {
    CRUSE_REQUIRE(rows > 0 && H > 0 && H <= 64 * LN_MAXE, CRUSE_E_SHAPE, "ln_bwd: H=%d must be in 1..%d", H, 64 * LN_MAXE);
    CRUSE_REQUIRE(interleave_g >= 1 && H % interleave_g == 0, CRUSE_E_SHAPE, "ln_bwd: groups=%d must divide H=%d", interleave_g, H);
    CRUSE_REQUIRE(seg_len >= 0 && (seg_len == 0 || (seg_stride >= seg_len && seg_off >= 0 && rows % seg_len == 0)), CRUSE_E_SHAPE,
                  "ln_bwd: bad row segments (len %d stride %lld off %lld, rows %lld)", seg_len, seg_stride, seg_off, rows);
    const RowSeg sg = {seg_len, seg_stride, seg_off};
    const bool vec = interleave_g == 1 && H % 4 == 0 && H <= 256 * LNV_MAXQ &&
                     ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)gamma) & 15) == 0);
    if (vec && H <= 768)
        hipLaunchKernelGGL((ln_bwd_vec_kernel<3, 2>), dim3(grid_for(rows, 16, lnb_grid())), dim3(256), 0, ST(stream), dy, x, mean,
                           rstd, gamma, rows, H, dx, dgamma, dbeta, sg);
    else if (vec)
        hipLaunchKernelGGL((ln_bwd_vec_kernel<4, 1>), dim3(grid_for(rows, 16, lnb_grid())), dim3(256), 0, ST(stream), dy, x, mean,
                           rstd, gamma, rows, H, dx, dgamma, dbeta, sg);
    else {
        CRUSE_REQUIRE(seg_len == 0, CRUSE_E_SHAPE, "ln_bwd: row segments need the vector form (one group, H %% 4 == 0, aligned)");
        hipLaunchKernelGGL(ln_bwd_kernel, dim3(grid_for(rows, 32, 1024)), dim3(256), 0, ST(stream), dy, x, mean, rstd,
                           gamma, rows, H, interleave_g, dx, dgamma, dbeta);
    }
    CRUSE_LAUNCH_CHECK("ln_bwd");
    return CRUSE_OK;
}

extern "C" int cruse_ln_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                            const float* gamma, long long rows, int H, int interleave_g,
                            float* dx, float* dgamma, float* dbeta, void* stream) {
    return ln_bwd_impl(dy, x, mean, rstd, gamma, rows, H, interleave_g, dx, dgamma, dbeta, 0, 0, 0, stream);
}

extern "C" int cruse_mask_loss_fwd(const float* mask, const float* nre, const float* nim, const float* cmag,
                                   long long rows, int Fn, int Fs, float alpha, float beta,
                                   double* loss_sum, float* dmask, float* dlogit, float* est_re, float* est_im,
                                   void* stream) {
    CRUSE_REQUIRE(rows > 0 && Fn > 0 && Fs >= Fn, CRUSE_E_SHAPE, "mask_loss: bad shape rows=%lld Fn=%d Fs=%d", rows, Fn, Fs);
    { int zrc = cruse_zero_async(loss_sum, sizeof(double), ST(stream), "mask_loss memset"); if (zrc) return zrc; }
    hipLaunchKernelGGL(mask_loss_kernel, dim3(grid_for(rows * Fs, 2048, 2048)), dim3(256), 0, ST(stream), mask, nre, nim,
                       cmag, rows, Fn, Fs, alpha, beta, loss_sum, dmask, dlogit, est_re, est_im);
    CRUSE_LAUNCH_CHECK("mask_loss");
    return CRUSE_OK;
}

extern "C" int cruse_sigmoid_bwd(const float* dmask, const float* mask, float* dlogit, long long n, void* stream) {
    CRUSE_REQUIRE(n > 0, CRUSE_E_SHAPE, "sigmoid_bwd: n=%lld", n);
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, ST(stream), dmask, mask, dlogit, n);
    CRUSE_LAUNCH_CHECK("sigmoid_bwd");
    return CRUSE_OK;
}

extern "C" int cruse_axpby(float* out, const float* x, const float* y, float a, float b, long long n, void* stream) {
    CRUSE_REQUIRE(n > 0, CRUSE_E_SHAPE, "axpby: n=%lld", n);
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, ST(stream), out, x, y, a, b, n);
    CRUSE_LAUNCH_CHECK("axpby");
    return CRUSE_OK;
}

extern "C" int cruse_accum_f64(double* acc, const double* x, int n, void* stream) {
    CRUSE_REQUIRE(n > 0, CRUSE_E_SHAPE, "accum_f64: n=%d", n);
    hipLaunchKernelGGL(accum_f64_kernel, dim3(cdiv(n, 64)), dim3(64), 0, ST(stream), acc, x, n);
    CRUSE_LAUNCH_CHECK("accum_f64");
    return CRUSE_OK;
}

extern "C" int cruse_step_health(unsigned* gru_status, const double* loss_sum, unsigned* health, double* loss_acc,
                                 double loss_scale, void* stream) {
    CRUSE_REQUIRE(health != nullptr, CRUSE_E_SHAPE, "step_health: health is required");
    hipLaunchKernelGGL(step_health_kernel, dim3(1), dim3(64), 0, ST(stream), gru_status, loss_sum, health, loss_acc, loss_scale);
    CRUSE_LAUNCH_CHECK("step_health");
    return CRUSE_OK;
}

extern "C" int cruse_counters_add(long long* const* counters, int n, long long v, void* stream) {
    CRUSE_REQUIRE(n > 0 && n <= 32, CRUSE_E_SHAPE, "counters_add: n=%d (1..32)", n);
    CounterPtrs c = {};
    for (int i = 0; i < n; ++i) c.p[i] = counters[i];
    hipLaunchKernelGGL(counters_add_kernel, dim3(1), dim3(64), 0, ST(stream), c, n, v);
    CRUSE_LAUNCH_CHECK("counters_add");
    return CRUSE_OK;
}

extern "C" int cruse_zero(void* p, size_t bytes, void* stream) {
    CRUSE_REQUIRE(bytes % 4 == 0, CRUSE_E_SHAPE, "zero: %zu bytes is not a multiple of 4", bytes);
    return cruse_zero_async(p, bytes, ST(stream), "zero");
}

extern "C" int cruse_adam_step_guarded(float* p, const float* g, float* m, float* v, long long n,
                                       float lr, float beta1, float beta2, float eps, float weight_decay,
                                       int step, float grad_scale, float max_norm, const double* gsumsq,
                                       const unsigned* skip_flag, int n_skip_words, const double* loss_check,
                                       unsigned* skipped, const double* loss_sum, double loss_scale, double* loss_acc,
                                       void* stream) {
    CRUSE_REQUIRE(n > 0 && step >= 1, CRUSE_E_SHAPE, "adam_step: n=%lld step=%d", n, step);
    CRUSE_REQUIRE(n_skip_words >= 0 && n_skip_words <= 8 && (n_skip_words == 0 || skip_flag != nullptr), CRUSE_E_SHAPE,
                  "adam_step: n_skip_words=%d (0..8, with skip_flag)", n_skip_words);
    CRUSE_REQUIRE(max_norm <= 0.f || gsumsq != nullptr, CRUSE_E_SHAPE, "adam_step: max_norm needs the gradient sum of squares");
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, ST(stream), p, g, m, v, n, lr, beta1, beta2,
                       eps, weight_decay, step, grad_scale, max_norm, gsumsq, skip_flag,
                       skip_flag ? n_skip_words : 0, loss_check, skipped, loss_sum, loss_scale, loss_acc);
    CRUSE_LAUNCH_CHECK("adam_step");
    return CRUSE_OK;
}

extern "C" int cruse_adam_step(float* p, const float* g, float* m, float* v, long long n,
                               float lr, float beta1, float beta2, float eps, float weight_decay,
                               int step, float grad_scale, void* stream) {
    return cruse_adam_step_guarded(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, 0.f, nullptr,
                                   nullptr, 0, nullptr, nullptr, nullptr, 1.0, nullptr, stream);
}

extern "C" int cruse_sumsq(const float* x, long long n, double* out, int accumulate, void* stream) {
    CRUSE_REQUIRE(n > 0, CRUSE_E_SHAPE, "sumsq: n=%lld", n);
    CRUSE_REQUIRE(((uintptr_t)x & 15) == 0, CRUSE_E_ALIGN, "sumsq: x must be 16-byte aligned");
    if (!accumulate) { int zrc = cruse_zero_async(out, sizeof(double), ST(stream), "sumsq memset"); if (zrc) return zrc; }
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n, 4096 * 4, 256)), dim3(1024), 0, ST(stream), x, n, out);
    CRUSE_LAUNCH_CHECK("sumsq");
    return CRUSE_OK;
}
