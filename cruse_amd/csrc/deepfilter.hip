// DeepFilter complex-coefficient head (model/deep_filter.py:15-41, BASELINE config 4).
// The reference unfolds BOTH the spectrum and the filter maps with an identity-kernel conv2d over a
// (2*f_dim+1) x (2*t_dim+1) neighbourhood and sums the tap-wise complex products, i.e.
//     out[b,f,t] = sum_{|df|<=f_dim, |dt|<=t_dim} X[b,f+df,t+dt] * H[b,f+df,t+dt]      (zero outside)
// -- a box sum of the element-wise complex product.  The 33 unfolded channels are never materialised:
// a [32 x 64] (f x t) tile of products plus halo is formed in LDS and each thread sums its window.
// Imaginary part uses the correct product xr*hi + xi*hr (the reference line :38 writes xr*hi twice;
// decision recorded in SURVEY.md 8a row a15 and in the oracle).  Layout is the reference's [B,F,T].
#include "common.h"

namespace {

constexpr int TFq = 32, TTq = 64;

// MODE 0: out_r/out_i = box(X*H).  MODE 1: (g_r, g_i) = box(dout) then the four operand gradients.
template <int MODE>
__global__ __launch_bounds__(256) void deepfilter_kernel(const float* xr, const float* xi, const float* hr, const float* hi,
                                                         const float* dor, const float* doi, int B, int F, int T, int fd,
                                                         int td, float* o0, float* o1, float* o2, float* o3) {
    extern __shared__ float sm[];
    const int HF = TFq + 2 * fd, HT = TTq + 2 * td;
    float* pr = sm;
    float* pi = sm + HF * HT;
    const int ntf = (F + TFq - 1) / TFq, ntt = (T + TTq - 1) / TTq;
    const int b = blockIdx.x / (ntf * ntt);
    const int rem = blockIdx.x % (ntf * ntt);
    const int f0 = (rem / ntt) * TFq, t0 = (rem % ntt) * TTq;
    const long long base = (long long)b * F * T;
    for (int i = threadIdx.x; i < HF * HT; i += 256) {
        const int lf = i / HT, lt = i % HT;
        const int f = f0 - fd + lf, t = t0 - td + lt;
        float a = 0.f, c = 0.f;
        if (f >= 0 && f < F && t >= 0 && t < T) {
            const long long o = base + (long long)f * T + t;
            if (MODE == 0) {
                const float ar = xr[o], ai = xi[o], br = hr[o], bi = hi[o];
                a = ar * br - ai * bi;
                c = ar * bi + ai * br;
            } else {
                a = dor[o]; c = doi[o];
            }
        }
        pr[i] = a; pi[i] = c;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TFq * TTq; i += 256) {
        const int lf = i / TTq, lt = i % TTq;
        const int f = f0 + lf, t = t0 + lt;
        if (f >= F || t >= T) continue;
        float sr = 0.f, si = 0.f;
        for (int df = 0; df <= 2 * fd; ++df)
            for (int dt = 0; dt <= 2 * td; ++dt) {
                sr += pr[(lf + df) * HT + lt + dt];
                si += pi[(lf + df) * HT + lt + dt];
            }
        const long long o = base + (long long)f * T + t;
        if (MODE == 0) {
            o0[o] = sr; o1[o] = si;
        } else {
            const float ar = xr[o], ai = xi[o], br = hr[o], bi = hi[o];
            o0[o] = sr * br + si * bi;        // d xr
            o1[o] = -sr * bi + si * br;       // d xi
            o2[o] = sr * ar + si * ai;        // d hr
            o3[o] = -sr * ai + si * ar;       // d hi
        }
    }
}

}  // namespace

extern "C" int cruse_deepfilter_fwd(const float* xr, const float* xi, const float* hr, const float* hi,
                                    int B, int F, int T, int f_dim, int t_dim, float* out_r, float* out_i, void* stream) {
    CRUSE_REQUIRE(B > 0 && F > 0 && T > 0 && f_dim >= 0 && t_dim >= 0 && f_dim <= 16 && t_dim <= 8, CRUSE_E_SHAPE,
                  "deepfilter: bad shape B=%d F=%d T=%d f_dim=%d t_dim=%d", B, F, T, f_dim, t_dim);
    const size_t lds = (size_t)2 * (TFq + 2 * f_dim) * (TTq + 2 * t_dim) * sizeof(float);
    const int grid = B * cdiv(F, TFq) * cdiv(T, TTq);
    hipLaunchKernelGGL(deepfilter_kernel<0>, dim3(grid), dim3(256), lds, (hipStream_t)stream, xr, xi, hr, hi,
                       (const float*)nullptr, (const float*)nullptr, B, F, T, f_dim, t_dim, out_r, out_i, (float*)nullptr,
                       (float*)nullptr);
    CRUSE_LAUNCH_CHECK("deepfilter_fwd");
    return CRUSE_OK;
}

extern "C" int cruse_deepfilter_bwd(const float* dout_r, const float* dout_i, const float* xr, const float* xi,
                                    const float* hr, const float* hi, int B, int F, int T, int f_dim, int t_dim,
                                    float* dxr, float* dxi, float* dhr, float* dhi, void* stream) {
    CRUSE_REQUIRE(B > 0 && F > 0 && T > 0 && f_dim >= 0 && t_dim >= 0 && f_dim <= 16 && t_dim <= 8, CRUSE_E_SHAPE,
                  "deepfilter_bwd: bad shape");
    const size_t lds = (size_t)2 * (TFq + 2 * f_dim) * (TTq + 2 * t_dim) * sizeof(float);
    const int grid = B * cdiv(F, TFq) * cdiv(T, TTq);
    hipLaunchKernelGGL(deepfilter_kernel<1>, dim3(grid), dim3(256), lds, (hipStream_t)stream, xr, xi, hr, hi, dout_r, dout_i,
                       B, F, T, f_dim, t_dim, dxr, dxi, dhr, dhi);
    CRUSE_LAUNCH_CHECK("deepfilter_bwd");
    return CRUSE_OK;
}
