// Time-domain losses in the loop: the waveform estimate is iSTFT(mask * noisy spectrum)
// and the criterion is SI-SNR -- `si_snr_loss` of train_base/loss.py:7-25, the only loss reachable through
// tools/train_stand.py:73-75 besides L1/MSE (loss_func/loss.py:48-56 `sisnr` is the same quantity without
// mean removal and with a different eps placement).
//
//   x_zm = x - mean(x); s_zm = s - mean(s); t = <x_zm,s_zm> s_zm / (|s_zm|^2 + eps)
//   loss = -mean_b 20 log10(eps + |t| / (|x_zm - t| + eps))
//
// Everything reduces to five moments per clip (sum x, s, x^2, s^2, xs; f64), so the forward is one streaming
// pass, and the gradient is dL/dx = A_b (x - mean x) + C_b (s - mean s) with per-clip scalars: a second
// streaming pass.  Both are HBM-bound (8 resp. 12 bytes per sample).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sisnr_moments_kernel(const float* x, const float* s, int B, int L, double* mom) {
    // grid = (chunks, B); mom[b][5] accumulated with f64 atomics (zeroed by the host wrapper)
    __shared__ double red[5][4];
    const int b = blockIdx.y;
    const float* xb = x + (long long)b * L;
    const float* sb = s + (long long)b * L;
    double m[5] = {0, 0, 0, 0, 0};
    float p[5] = {0, 0, 0, 0, 0};
    int n = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) {
        const float a = xb[i], c = sb[i];
        p[0] += a; p[1] += c; p[2] += a * a; p[3] += c * c; p[4] += a * c;
        if (++n == 16) {
#pragma unroll
            for (int k = 0; k < 5; ++k) { m[k] += p[k]; p[k] = 0.f; }
            n = 0;
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        m[k] = wave_sum_d(m[k] + (double)p[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = m[k];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const int k = threadIdx.x;
        atomicAdd(&mom[b * 5 + k], red[k][0] + red[k][1] + red[k][2] + red[k][3]);
    }
}

// per clip: loss term and the gradient scalars.  coef[b] = {A, C, mean_x, mean_s}
__global__ void sisnr_finalize_kernel(const double* mom, int B, int L, double eps, double* loss, float* coef) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double n = (double)L;
    const double sx = mom[b * 5 + 0], ss = mom[b * 5 + 1], sxx = mom[b * 5 + 2], sss = mom[b * 5 + 3], sxs = mom[b * 5 + 4];
    const double mx = sx / n, ms = ss / n;
    const double P = sxs - n * mx * ms;                 // <x_zm, s_zm>
    double S2 = sss - n * ms * ms; if (S2 < 0) S2 = 0;  // |s_zm|^2
    double X2 = sxx - n * mx * mx; if (X2 < 0) X2 = 0;  // |x_zm|^2
    const double S = S2 + eps;
    const double alpha = P / S;
    const double nt = fabs(alpha) * sqrt(S2);           // |t|
    double e2 = X2 - 2.0 * alpha * P + alpha * alpha * S2; if (e2 < 0) e2 = 0;
    const double ne = sqrt(e2);
    const double r = nt / (ne + eps);
    const double lb = -20.0 * log10(eps + r) / (double)B;
    atomicAdd(loss, lb);
    // d lb / d r
    const double dr = -20.0 / (log(10.0) * (eps + r)) / (double)B;
    // r = nt/(ne+eps);  d nt/da = sign(alpha) |s| s/S ;  d ne/da = (e - s <e,s>/S)/ne,  e = a - alpha s
    const double sgn = alpha >= 0 ? 1.0 : -1.0;
    const double es = P - alpha * S2;                    // <e, s>
    const double k_nt = dr / (ne + eps) * sgn * sqrt(S2) / S;                    // * s
    const double k_ne = ne > 0 ? -dr * nt / ((ne + eps) * (ne + eps)) / ne : 0.0; // * (e - s es/S)
    // grad = k_nt s + k_ne (a - alpha s - s es/S) = k_ne a + (k_nt - k_ne (alpha + es/S)) s
    coef[b * 4 + 0] = (float)k_ne;
    coef[b * 4 + 1] = (float)(k_nt - k_ne * (alpha + es / S));
    coef[b * 4 + 2] = (float)mx;
    coef[b * 4 + 3] = (float)ms;
}

__global__ __launch_bounds__(256) void sisnr_grad_kernel(const float* x, const float* s, const float* coef, int B, int L,
                                                         float gscale, float* dx) {
    const int b = blockIdx.y;
    const float A = coef[b * 4 + 0] * gscale, C = coef[b * 4 + 1] * gscale, mx = coef[b * 4 + 2], ms = coef[b * 4 + 3];
    const long long o = (long long)b * L;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256)
        dx[o + i] = A * (x[o + i] - mx) + C * (s[o + i] - ms);
}

// est = mask * noisy spectrum on the first Fn bins, 0 above (utils/utils.py:418-420, R8)
__global__ __launch_bounds__(256) void mask_apply_kernel(const float* mask, const float* nre, const float* nim,
                                                         long long rows, int Fn, int Fs, float* ere, float* eim) {
    const long long n = rows * Fs;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / Fs;
        const int f = (int)(i - r * Fs);
        const float m = f < Fn ? mask[r * Fn + f] : 0.f;
        ere[i] = m * nre[i];
        eim[i] = m * nim[i];
    }
}
// dlogit = (dre*nre + dim*nim) * m (1 - m)   (dmask when `through_sigmoid` == 0)
__global__ __launch_bounds__(256) void mask_apply_bwd_kernel(const float* dre, const float* dim, const float* nre,
                                                             const float* nim, const float* mask, long long rows, int Fn,
                                                             int Fs, int through_sigmoid, float* dout) {
    const long long n = rows * Fn;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / Fn;
        const int f = (int)(i - r * Fn);
        const long long j = r * Fs + f;
        float d = dre[j] * nre[j] + dim[j] * nim[j];
        if (through_sigmoid) { const float m = mask[i]; d *= m * (1.f - m); }
        dout[i] = d;
    }
}

inline int blocks_for(long long n, int per) {
    long long g = (n + per - 1) / per;
    if (g < 1) g = 1;
    if (g > 4096) g = 4096;
    return (int)g;
}

}  // namespace

extern "C" int cruse_sisnr_fwd(const float* x, const float* s, int B, int L, float eps,
                               double* mom, double* loss, float* coef, void* stream) {
    CRUSE_REQUIRE(B > 0 && L > 0, CRUSE_E_SHAPE, "sisnr: bad shape B=%d L=%d", B, L);
    hipStream_t st = (hipStream_t)stream;
    { int rc = cruse_zero_async(mom, (size_t)B * 5 * sizeof(double), st, "sisnr"); if (rc) return rc; }
    { int rc = cruse_zero_async(loss, sizeof(double), st, "sisnr"); if (rc) return rc; }
    int chunks = (L + 256 * 16 - 1) / (256 * 16);
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL(sisnr_moments_kernel, dim3(chunks, B), dim3(256), 0, st, x, s, B, L, mom);
    CRUSE_LAUNCH_CHECK("sisnr_moments");
    hipLaunchKernelGGL(sisnr_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, st, mom, B, L, (double)eps, loss, coef);
    CRUSE_LAUNCH_CHECK("sisnr_finalize");
    return CRUSE_OK;
}

extern "C" int cruse_sisnr_bwd(const float* x, const float* s, const float* coef, int B, int L, float grad_scale,
                               float* dx, void* stream) {
    CRUSE_REQUIRE(B > 0 && L > 0, CRUSE_E_SHAPE, "sisnr_bwd: bad shape");
    int chunks = (L + 256 * 8 - 1) / (256 * 8);
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL(sisnr_grad_kernel, dim3(chunks, B), dim3(256), 0, (hipStream_t)stream, x, s, coef, B, L, grad_scale, dx);
    CRUSE_LAUNCH_CHECK("sisnr_grad");
    return CRUSE_OK;
}

extern "C" int cruse_mask_apply(const float* mask, const float* nre, const float* nim, long long rows, int Fn, int Fs,
                                float* est_re, float* est_im, void* stream) {
    CRUSE_REQUIRE(rows > 0 && Fn > 0 && Fs >= Fn, CRUSE_E_SHAPE, "mask_apply: bad shape");
    hipLaunchKernelGGL(mask_apply_kernel, dim3(blocks_for(rows * Fs, 1024)), dim3(256), 0, (hipStream_t)stream, mask, nre,
                       nim, rows, Fn, Fs, est_re, est_im);
    CRUSE_LAUNCH_CHECK("mask_apply");
    return CRUSE_OK;
}

extern "C" int cruse_mask_apply_bwd(const float* dre, const float* dim, const float* nre, const float* nim,
                                    const float* mask, long long rows, int Fn, int Fs, int through_sigmoid,
                                    float* dout, void* stream) {
    CRUSE_REQUIRE(rows > 0 && Fn > 0 && Fs >= Fn, CRUSE_E_SHAPE, "mask_apply_bwd: bad shape");
    hipLaunchKernelGGL(mask_apply_bwd_kernel, dim3(blocks_for(rows * Fn, 1024)), dim3(256), 0, (hipStream_t)stream, dre, dim,
                       nre, nim, mask, rows, Fn, Fs, through_sigmoid, dout);
    CRUSE_LAUNCH_CHECK("mask_apply_bwd");
    return CRUSE_OK;
}

// ---------------------------------------------------------------------------------------------------
// SNR-weighted speech-distortion loss (`sdnr`, loss_func/loss.py:151-175; vad == 1 because
// activity_detector_tf_frame is `pass`, utils/utils.py:217-219) with the mask as the gain est_g:
//   L_noise  = mean_{b,f} sum_{c,t} (noise * g)^2        L_speech = mean_{b,f} sum_{c,t} (clean - g*clean)^2
//   loss = alpha * L_speech + (1 - alpha) * L_noise,     alpha = 10^(snr/10) / (10^(snr/10) + 10^(beta/10))
// clean / noise spectra are [rows,Fs] re/im pairs (c = 2), the gain is the mask on the first Fn bins and 0 above.
// ---------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void sdnr_kernel(const float* mask, const float* cre, const float* cim, const float* nre,
                                                   const float* nim, long long rows, int Fn, int Fs, float alpha,
                                                   float inv_norm, double* loss_sum, float* dmask, float* dlogit) {
    __shared__ double sred[4];
    const long long n = rows * Fs;
    double acc = 0.0;
    float part = 0.f;
    int cnt = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / Fs;
        const int f = (int)(i - r * Fs);
        const float m = f < Fn ? mask[r * Fn + f] : 0.f;
        const float s2 = cre[i] * cre[i] + cim[i] * cim[i];                       // |clean|^2
        const float dr = nre[i] - cre[i], di = nim[i] - cim[i];
        const float v2 = dr * dr + di * di;                                       // |noise|^2 = |noisy - clean|^2
        part += alpha * (1.f - m) * (1.f - m) * s2 + (1.f - alpha) * m * m * v2;
        if (++cnt == 32) { acc += part; part = 0.f; cnt = 0; }
        if (f < Fn && (dmask || dlogit)) {
            const float dm = 2.f * (-alpha * (1.f - m) * s2 + (1.f - alpha) * m * v2) * inv_norm;
            if (dmask) dmask[r * Fn + f] = dm;
            if (dlogit) dlogit[r * Fn + f] = dm * m * (1.f - m);
        }
    }
    acc = wave_sum_d(acc + part);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, sred[0] + sred[1] + sred[2] + sred[3]);
}
}  // namespace

extern "C" int cruse_mask_sdnr_fwd(const float* mask, const float* cre, const float* cim, const float* nre, const float* nim,
                                   long long rows, int Fn, int Fs, int B, float snr_db, float beta_db,
                                   double* loss_sum, float* dmask, float* dlogit, void* stream) {
    CRUSE_REQUIRE(rows > 0 && Fn > 0 && Fs >= Fn && B > 0, CRUSE_E_SHAPE, "mask_sdnr: bad shape");
    const double st = pow(10.0, snr_db / 10.0), bt = pow(10.0, beta_db / 10.0);
    const float alpha = (float)(st / (st + bt));
    { int rc = cruse_zero_async(loss_sum, sizeof(double), (hipStream_t)stream, "mask_sdnr"); if (rc) return rc; }
    // loss = loss_sum / (B * Fs)  (mean over b and f of the (c,t)-summed squares)
    hipLaunchKernelGGL(sdnr_kernel, dim3(blocks_for(rows * Fs, 2048)), dim3(256), 0, (hipStream_t)stream, mask, cre, cim, nre,
                       nim, rows, Fn, Fs, alpha, 1.0f / ((float)B * (float)Fs), loss_sum, dmask, dlogit);
    CRUSE_LAUNCH_CHECK("mask_sdnr");
    return CRUSE_OK;
}

// ---------------------------------------------------------------------------------------------------
// Waveform L1 / MSE (`l1_loss` / `mse_loss` = torch.nn.L1Loss / MSELoss, train_base/loss.py:3-4, reachable through
// tools/train_stand.py:73-75) on the enhanced waveform est = iSTFT(mask * noisy spectrum): one streaming pass gives the loss
// sum (f64) and the gradient wrt est -- sign(est - ref) * gscale or 2 (est - ref) * gscale, gscale = 1 / numel for
// reduction = "mean".  HBM-bound: 8 B in + 4 B out per sample.
// ---------------------------------------------------------------------------------------------------
namespace {
template <bool MSE>
__global__ __launch_bounds__(256) void wave_l1mse_kernel(const float* x, const float* s, long long n, float gscale,
                                                         double* loss_sum, float* dx) {
    __shared__ double sred[4];
    double acc = 0.0;
    float part = 0.f;
    int cnt = 0;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(x)[i], b = reinterpret_cast<const float4*>(s)[i];
        const float d[4] = {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w};
        float g[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            part += MSE ? d[e] * d[e] : fabsf(d[e]);
            g[e] = MSE ? 2.f * d[e] * gscale : (d[e] > 0.f ? gscale : (d[e] < 0.f ? -gscale : 0.f));     // torch: sign(0) = 0
        }
        if (dx) reinterpret_cast<float4*>(dx)[i] = make_float4(g[0], g[1], g[2], g[3]);
        if (++cnt == 8) { acc += part; part = 0.f; cnt = 0; }
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {                       // tail (n % 4 samples)
        const long long i = (n4 << 2) + threadIdx.x;
        const float d = x[i] - s[i];
        part += MSE ? d * d : fabsf(d);
        if (dx) dx[i] = MSE ? 2.f * d * gscale : (d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f));
    }
    acc = wave_sum_d(acc + part);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, sred[0] + sred[1] + sred[2] + sred[3]);
}
}  // namespace

extern "C" int cruse_wave_l1_mse(const float* est, const float* ref, long long n, int mse, float grad_scale,
                                 double* loss_sum, float* dest, void* stream) {
    CRUSE_REQUIRE(n > 0 && est != nullptr && ref != nullptr && loss_sum != nullptr, CRUSE_E_SHAPE, "wave_l1_mse: bad arguments");
    CRUSE_REQUIRE((((uintptr_t)est | (uintptr_t)ref | (uintptr_t)dest) & 15) == 0, CRUSE_E_ALIGN, "wave_l1_mse: unaligned buffers");
    hipStream_t st = (hipStream_t)stream;
    { int rc = cruse_zero_async(loss_sum, sizeof(double), st, "wave_l1_mse"); if (rc) return rc; }
    const int grid = blocks_for(n >> 2, 2048);
    if (mse) hipLaunchKernelGGL(wave_l1mse_kernel<true>, dim3(grid), dim3(256), 0, st, est, ref, n, grad_scale, loss_sum, dest);
    else hipLaunchKernelGGL(wave_l1mse_kernel<false>, dim3(grid), dim3(256), 0, st, est, ref, n, grad_scale, loss_sum, dest);
    CRUSE_LAUNCH_CHECK("wave_l1_mse");
    return CRUSE_OK;
}
