// Elementwise and reduction kernels for the data formats either side of the hot path (SURVEY.md 8a rows a13, a14,
// a16; 8f item 3):
//   train_base/acoustics/mask.py:8-63        IRM / cIRM build, compress_cIRM / decompress_cIRM, complex_mul
//   loss_func/loss.py:48-118                  sisnr, rmse, c_rmse (values and gradients wrt the estimate)
//   model/mtfaa.py:130-163                    ComplexLinearProjection magnitude sqrt(r^2+i^2+1e-8) ** alpha
//   train_base/acoustics/feature.py:360-390   magnitude/phase <-> real/imag of CustomSTFT / CustomISTFT
//   dataset/dataset.py:236-264                SynDataset.snr_mix: peak-normalise, RMS-based SNR scaling, mix
// All streaming, HBM-bound; f64 accumulation for loss sums.
#include "common.h"

namespace {

constexpr float EPS32 = 1.1920928955078125e-07f;          // np.finfo(np.float32).eps = train_base/constant.py EPSILON

__device__ __forceinline__ float compress_cirm(float m, float K, float C) {      // mask.py:40-50
    m = m <= -100.f ? -100.f : m;
    const float e = expf(-C * m);
    return K * (1.f - e) / (1.f + e);
}
__device__ __forceinline__ float decompress_cirm(float m, float K, float limit) {   // mask.py:53-56
    m = m >= limit ? limit : (m <= -limit ? -limit : m);
    return -K * logf((K - m) / (K + m));
}

// mode 0: IRM = compress(clean_mag / (noisy_mag + EPS))                               (a = noisy_mag, c = clean_mag)
// mode 1: cIRM = compress([ (nr*cr + ni*ci)/den, (nr*ci - ni*cr)/den ]), den = nr^2 + ni^2 + EPS   -> out[n][2]
// mode 2: compress(a)          mode 3: decompress(a)
// mode 4: complex_mul (a + ib)(c + id) -> out = real, out2 = imag
// mode 5: element-wise pair product out = a*c, out2 = b*d   (PreProcess.masking "complex_mapping", utils/utils.py:421-423)
// mode 6: out = log(a)                                       (PreProcess.log_transform, utils/utils.py:414-415)
// adjoints (the reference's versions are plain torch and differentiable; they sit between the network and the loss):
// mode 7: out = c * d compress(a)/da      mode 8: out = c * d decompress(a)/da      (c = upstream gradient)
// mode 9: (a + ib) * conj(c + id) -> out = a*c + b*d, out2 = b*c - a*d   (both gradients of mode 4)
__global__ void mask_ops_kernel(int mode, const float* a, const float* b, const float* c, const float* d, long long n,
                                float K, float C, float limit, float* out, float* out2) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (mode == 0) {
            out[i] = compress_cirm(c[i] / (a[i] + EPS32), K, C);
        } else if (mode == 1) {
            const float nr = a[i], ni = b[i], cr = c[i], ci = d[i];
            const float den = nr * nr + ni * ni + EPS32;
            out[2 * i] = compress_cirm((nr * cr + ni * ci) / den, K, C);
            out[2 * i + 1] = compress_cirm((nr * ci - ni * cr) / den, K, C);
        } else if (mode == 2) {
            out[i] = compress_cirm(a[i], K, C);
        } else if (mode == 3) {
            out[i] = decompress_cirm(a[i], K, limit);
        } else if (mode == 4) {
            const float nr = a[i], ni = b[i], mr = c[i], mi = d[i];
            out[i] = nr * mr - ni * mi;
            out2[i] = nr * mi + ni * mr;
        } else if (mode == 5) {
            out[i] = a[i] * c[i];
            out2[i] = b[i] * d[i];
        } else if (mode == 6) {
            out[i] = logf(a[i]);
        } else if (mode == 7) {
            const float m = a[i];
            const float e = expf(-C * m);
            out[i] = m <= -100.f ? 0.f : c[i] * (2.f * K * C * e / ((1.f + e) * (1.f + e)));
        } else if (mode == 8) {
            const float m = a[i];
            out[i] = (m >= limit || m <= -limit) ? 0.f : c[i] * (2.f * K * K / (K * K - m * m));
        } else {
            const float gr = a[i], gi = b[i], mr = c[i], mi = d[i];
            out[i] = gr * mr + gi * mi;
            out2[i] = gi * mr - gr * mi;
        }
    }
}

// mode 0: (re, im) -> (mag = sqrt(re^2 + im^2 + eps) ** alpha, phase = atan2(im, re))     [phase may be NULL]
// mode 1: (mag, phase) -> (re = mag cos p, im = mag sin p)
// mode 2: backward of mode 0: magnitude (g, optional) dre = dm * alpha * amp^(alpha-2) * re, dim likewise; phase (g2,
//         optional) dre -= dp * im / (re^2 + im^2), dim += dp * re / (re^2 + im^2)
// mode 3: backward of mode 1 (a = mag, b = phase; g, g2 = d re, d im): o1 = d mag = g cos p + g2 sin p, o2 = d phase
__global__ void polar_kernel(int mode, const float* a, const float* b, const float* g, const float* g2, long long n,
                             float eps, float alpha, float* o1, float* o2) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (mode == 0) {
            const float r = a[i], im = b[i];
            const float amp = sqrtf(r * r + im * im + eps);
            o1[i] = alpha == 1.f ? amp : powf(amp, alpha);
            if (o2) o2[i] = atan2f(im, r);
        } else if (mode == 1) {
            float s, c;
            sincosf(b[i], &s, &c);
            o1[i] = a[i] * c; o2[i] = a[i] * s;
        } else if (mode == 2) {
            const float r = a[i], im = b[i];
            const float q = r * r + im * im + eps;
            const float k = g ? g[i] * alpha * powf(q, 0.5f * alpha - 1.f) : 0.f;
            float dr = k * r, di = k * im;
            if (g2) {
                const float q0 = r * r + im * im;
                const float w = q0 > 0.f ? g2[i] / q0 : 0.f;
                dr -= w * im; di += w * r;
            }
            o1[i] = dr; o2[i] = di;
        } else {
            float sn, cs;
            sincosf(b[i], &sn, &cs);
            o1[i] = g[i] * cs + g2[i] * sn;
            o2[i] = a[i] * (g2[i] * cs - g[i] * sn);
        }
    }
}

// rmse (loss_func/loss.py:59-78): sum |est - ref| / (B*T*F);  dest = sign(est - ref) * gscale
__global__ __launch_bounds__(256) void rmse_kernel(const float* ref, const float* est, long long n, float gscale,
                                                   double* loss_sum, float* dest) {
    __shared__ double red[4];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float e = est[i] - ref[i];
        acc += (double)sqrtf(e * e);
        if (dest) dest[i] = e > 0.f ? gscale : (e < 0.f ? -gscale : 0.f);
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, red[0] + red[1] + red[2] + red[3]);
}

// c_rmse (loss_func/loss.py:88-118) AS WRITTEN (its tmp3 / tmp4 mix mag_est^c with the reference phase and vice versa):
//   t1 = |est|^c, t2 = |ref|^c,  A = t1 cos(ph_ref) - t2 cos(ph_est),  Bq = t1 (sin(ph_ref) - sin(ph_est))
//   loss = (1-beta) sum (t2 - t1)^2 + beta sum (A^2 + Bq^2)          ref/est are [B,2,T,F]: plane 0 real, plane 1 imag
__global__ __launch_bounds__(256) void c_rmse_kernel(const float* ref, const float* est, int B, long long TF, float c,
                                                     float beta, double* loss_sum, float* dest) {
    __shared__ double red[4];
    double acc = 0.0;
    const long long n = (long long)B * TF;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long b = i / TF, j = i - b * TF;
        const long long o = b * 2 * TF + j;
        const float rr = ref[o], ri = ref[o + TF], er = est[o], ei = est[o + TF];
        const float mr = sqrtf(rr * rr + ri * ri), me = sqrtf(er * er + ei * ei);
        const float pr = atan2f(ri, rr), pe = atan2f(ei, er);
        const float t1 = powf(me, c), t2 = powf(mr, c);
        float spr, cpr, spe, cpe;
        sincosf(pr, &spr, &cpr); sincosf(pe, &spe, &cpe);
        const float A = t1 * cpr - t2 * cpe, Bq = t1 * (spr - spe);
        const float l1 = (t2 - t1) * (t2 - t1), l2 = A * A + Bq * Bq;
        acc += (double)((1.f - beta) * l1 + beta * l2);
        if (dest) {
            const float dt1 = -2.f * (1.f - beta) * (t2 - t1) + beta * (2.f * A * cpr + 2.f * Bq * (spr - spe));
            const float dme = dt1 * c * powf(me, c - 1.f);
            const float dpe = beta * (2.f * A * t2 * spe - 2.f * Bq * t1 * cpe);
            const float inv = 1.f / me, inv2 = inv * inv;
            dest[o] = dme * er * inv - dpe * ei * inv2;
            dest[o + TF] = dme * ei * inv + dpe * er * inv2;
        }
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, red[0] + red[1] + red[2] + red[3]);
}

// wo_male (loss_func/loss.py:121-148, repairs of SURVEY row a11) on EXPLICIT spectra [B,2,TF] (ref, est, unproc):
//   loss_sum = sum exp(alpha / (beta + |ref|/|unproc|)) * |log10(|est|+1) - log10(|ref|+1)|      (divide by B*T*F)
// (the training step uses the fused mask form, pointwise.hip mask_loss_kernel; this is loss_func.loss('WO_MALE'))
// element (b, j): real part at b*bstride + j, imaginary part pstride further ([B,2,TF]: 2TF, TF; [2,B,TF]: TF, B*TF)
__global__ __launch_bounds__(256) void wo_male_spec_kernel(const float* ref, const float* est, const float* unp, int B,
                                                           long long TF, long long bstride, long long pstride, float alpha,
                                                           float beta, float gscale, double* loss_sum, float* dest) {
    __shared__ double red[4];
    double acc = 0.0;
    const long long n = (long long)B * TF;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long b = i / TF, j = i - b * TF;
        const long long o = b * bstride + j, TFp = pstride;
        const float mr = sqrtf(ref[o] * ref[o] + ref[o + TFp] * ref[o + TFp]);
        const float er = est[o], ei = est[o + TFp];
        const float me = sqrtf(er * er + ei * ei);
        const float mu = sqrtf(unp[o] * unp[o] + unp[o + TFp] * unp[o + TFp]);
        const float w = expf(alpha / (beta + mr / mu));
        const float d = log10f(me + 1.f) - log10f(mr + 1.f);
        acc += (double)(w * fabsf(d));
        if (dest) {
            const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            const float k = me > 0.f ? gscale * w * sg * 0.4342944819032518f / ((me + 1.f) * me) : 0.f;
            dest[o] = k * er; dest[o + TFp] = k * ei;
        }
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, red[0] + red[1] + red[2] + red[3]);
}

// sisnr (loss_func/loss.py:48-56), NO mean removal: from the moments {sum x, s, x^2, s^2, xs} of cruse_sisnr_fwd's first pass
//   alpha = <x,s>/(<s,s>+eps); target = alpha^2 <s,s>; noise = <x,x> - 2 alpha <x,s> + alpha^2 <s,s>
//   snr_b = 10 log10(target / (noise + eps) + eps); value = mean_b snr_b.  coef[b] = {A, C, 0, 0}: d value / dx = A x + C s
__global__ void sisnr_plain_finalize_kernel(const double* mom, int B, double eps, double* value, float* coef) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double X2 = mom[b * 5 + 2], S2 = mom[b * 5 + 3], P = mom[b * 5 + 4];
    const double S = S2 + eps, al = P / S;
    const double tg = al * al * S2;
    const double nz = X2 - 2.0 * al * P + al * al * S2;
    const double q = tg / (nz + eps) + eps;
    atomicAdd(value, 10.0 * log10(q) / (double)B);
    const double dq = 10.0 / (log(10.0) * q) / (double)B;
    // d tg = 2 al S2 / S * s ;  d nz = 2 x - 2 al s - 2 P/S s + 2 al S2/S s
    const double dtg_s = 2.0 * al * S2 / S;
    const double dnz_x = 2.0, dnz_s = -2.0 * al - 2.0 * P / S + 2.0 * al * S2 / S;
    const double k_t = dq / (nz + eps), k_n = -dq * tg / ((nz + eps) * (nz + eps));
    coef[b * 4 + 0] = (float)(k_n * dnz_x);
    coef[b * 4 + 1] = (float)(k_t * dtg_s + k_n * dnz_s);
    coef[b * 4 + 2] = 0.f;
    coef[b * 4 + 3] = 0.f;
}

// ---- snr_mix (dataset/dataset.py:236-264) ----------------------------------------------------------------
// pass 1: per clip  st[b] = {max|clean| bits, max|noise| bits} (u32, atomicMax on the non-negative float's bit pattern),
//                   ss[b] = {sum clean^2, sum noise^2} (f64)
__global__ __launch_bounds__(256) void snr_mix_stats_kernel(const float* clean, const float* noise, int L, unsigned* st, double* ss) {
    __shared__ float rmax[2][4];
    __shared__ double rsum[2][4];
    const int b = blockIdx.y;
    const float* c = clean + (long long)b * L;
    const float* v = noise + (long long)b * L;
    float mc = 0.f, mv = 0.f;
    double sc = 0.0, sv = 0.0;
    float pc = 0.f, pv = 0.f;
    int cnt = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) {
        const float a = c[i], n = v[i];
        mc = fmaxf(mc, fabsf(a)); mv = fmaxf(mv, fabsf(n));
        pc += a * a; pv += n * n;
        if (++cnt == 16) { sc += pc; sv += pv; pc = pv = 0.f; cnt = 0; }
    }
    sc += pc; sv += pv;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mc = fmaxf(mc, __shfl_xor(mc, o, 64)); mv = fmaxf(mv, __shfl_xor(mv, o, 64)); }
    sc = wave_sum_d(sc); sv = wave_sum_d(sv);
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        rmax[0][w] = mc; rmax[1][w] = mv; rsum[0][w] = sc; rsum[1][w] = sv;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int k = threadIdx.x;
        const float m = fmaxf(fmaxf(rmax[k][0], rmax[k][1]), fmaxf(rmax[k][2], rmax[k][3]));
        atomicMax(&st[b * 2 + k], __float_as_uint(m));
        atomicAdd(&ss[b * 2 + k], rsum[k][0] + rsum[k][1] + rsum[k][2] + rsum[k][3]);
    }
}
// pass 2: clean_n = clean / (max|clean| + eps); noise_n = noise / (max|noise| + eps);
//         scalar = rms(clean_n) / 10^(snr/20) / (rms(noise_n) + eps); noisy = clean_n + scalar * noise_n
__global__ __launch_bounds__(256) void snr_mix_apply_kernel(const float* clean, const float* noise, const float* snr_db, int L,
                                                            const unsigned* st, const double* ss, float eps,
                                                            float* clean_out, float* noise_out, float* noisy) {
    const int b = blockIdx.y;
    const float ic = 1.f / (__uint_as_float(st[b * 2]) + eps), iv = 1.f / (__uint_as_float(st[b * 2 + 1]) + eps);
    const float rc = sqrtf((float)(ss[b * 2] / (double)L)) * ic;             // rms of the normalised clean
    const float rv = sqrtf((float)(ss[b * 2 + 1] / (double)L)) * iv;
    const float scalar = rc / powf(10.f, snr_db[b] / 20.f) / (rv + eps);
    const long long o = (long long)b * L;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) {
        const float c = clean[o + i] * ic;
        const float n = noise[o + i] * iv * scalar;
        if (clean_out) clean_out[o + i] = c;
        if (noise_out) noise_out[o + i] = n;
        noisy[o + i] = c + n;
    }
}

// y[b,n] = gain * (1-a) * sum_{k < taps} a^k x[b,n-k]   (one-pole low-pass as a truncated FIR: the synthetic "speech-like"
// spectral tilt of the bench / SyntheticPairs clips)
__global__ __launch_bounds__(256) void onepole_fir_kernel(const float* x, int L, float a, int taps, float gain, float* y) {
    __shared__ float xs[256 + 128];
    const int b = blockIdx.y, n0 = blockIdx.x * 256;
    const float* xb = x + (long long)b * L;
    for (int i = threadIdx.x; i < 256 + taps - 1; i += 256) {
        const int n = n0 - (taps - 1) + i;
        xs[i] = (n >= 0 && n < L) ? xb[n] : 0.f;
    }
    __syncthreads();
    const int n = n0 + threadIdx.x;
    if (n >= L) return;
    float acc = 0.f, w = (1.f - a) * gain;
    for (int k = 0; k < taps; ++k) { acc += w * xs[threadIdx.x + taps - 1 - k]; w *= a; }
    y[(long long)b * L + n] = acc;
}

// y[b,n] = sum_{k < R, k <= n} h[k] x[b,n-k]: scipy.signal.fftconvolve(x, h)[:L], the room-impulse-response step of
// SynDataset.snr_mix (dataset/dataset.py:245-248), as a direct causal FIR.  One block = 1024 outputs of one clip; the taps
// are walked in chunks of 256 with the matching input window (1279 samples) staged in LDS; a thread owns outputs
// n0 + t + 256 j, so the window reads of a wave are 64 consecutive floats (conflict-free) and the tap is a broadcast.
__global__ __launch_bounds__(256) void fir_causal_kernel(const float* x, const float* h, long long h_bstride, int L, int R, float* y) {
    constexpr int NO = 1024, KC = 256;
    __shared__ float xs[NO + KC];
    __shared__ float hs[KC];
    const int b = blockIdx.y, n0 = blockIdx.x * NO, t = threadIdx.x;
    const float* xb = x + (long long)b * L;
    const float* hb = h + (long long)b * h_bstride;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int kmax = min(R, n0 + NO);                        // taps beyond the last output's index only meet x[< 0]
    for (int kc = 0; kc < kmax; kc += KC) {
        __syncthreads();
        // xs[i] = x[n0 - kc - (KC - 1) + i],  i in [0, NO + KC - 1)
        for (int i = t; i < NO + KC - 1; i += 256) {
            const int n = n0 - kc - (KC - 1) + i;
            xs[i] = (n >= 0 && n < L) ? xb[n] : 0.f;
        }
        hs[t] = (kc + t < R) ? hb[kc + t] : 0.f;
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < KC; ++k) {
            const float w = hs[k];
            const int o = t + (KC - 1) - k;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += w * xs[o + 256 * j];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + t + 256 * j;
        if (n < L) y[(long long)b * L + n] = acc[j];
    }
}

inline int eblocks(long long n, int per = 1024, int cap = 4096) {
    long long g = (n + per - 1) / per;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace

#define ST(s) ((hipStream_t)(s))

extern "C" int cruse_mask_ops(int mode, const float* a, const float* b, const float* c, const float* d, long long n,
                              float K, float C, float limit, float* out, float* out2, void* stream) {
    CRUSE_REQUIRE(mode >= 0 && mode <= 9 && n > 0 && a && out, CRUSE_E_SHAPE, "mask_ops: bad arguments (mode %d, n %lld)", mode, n);
    CRUSE_REQUIRE(!((mode == 0 || mode == 7 || mode == 8) && !c) && !((mode == 1 || mode == 4 || mode == 5 || mode == 9) && !(b && c && d)) &&
                  !((mode == 4 || mode == 5 || mode == 9) && !out2), CRUSE_E_SHAPE,
                  "mask_ops: missing operand for mode %d", mode);
    hipLaunchKernelGGL(mask_ops_kernel, dim3(eblocks(n)), dim3(256), 0, ST(stream), mode, a, b, c, d, n, K, C, limit, out, out2);
    CRUSE_LAUNCH_CHECK("mask_ops");
    return CRUSE_OK;
}

extern "C" int cruse_polar(int mode, const float* a, const float* b, const float* g, const float* g2, long long n, float eps,
                           float alpha, float* o1, float* o2, void* stream) {
    CRUSE_REQUIRE(mode >= 0 && mode <= 3 && n > 0 && a && b && o1 && (mode == 0 || o2) && (mode != 2 || g || g2) &&
                  (mode != 3 || (g && g2)), CRUSE_E_SHAPE, "polar: bad arguments (mode %d)", mode);
    hipLaunchKernelGGL(polar_kernel, dim3(eblocks(n)), dim3(256), 0, ST(stream), mode, a, b, g, g2, n, eps, alpha, o1, o2);
    CRUSE_LAUNCH_CHECK("polar");
    return CRUSE_OK;
}

extern "C" int cruse_rmse(const float* ref, const float* est, long long n, float grad_scale, double* loss_sum, float* dest,
                          void* stream) {
    CRUSE_REQUIRE(n > 0, CRUSE_E_SHAPE, "rmse: n=%lld", n);
    { int rc = cruse_zero_async(loss_sum, sizeof(double), ST(stream), "rmse"); if (rc) return rc; }
    hipLaunchKernelGGL(rmse_kernel, dim3(eblocks(n, 2048, 1024)), dim3(256), 0, ST(stream), ref, est, n, grad_scale, loss_sum, dest);
    CRUSE_LAUNCH_CHECK("rmse");
    return CRUSE_OK;
}

extern "C" int cruse_c_rmse(const float* ref, const float* est, int B, long long TF, float c, float beta, double* loss_sum,
                            float* dest, void* stream) {
    CRUSE_REQUIRE(B > 0 && TF > 0, CRUSE_E_SHAPE, "c_rmse: bad shape");
    { int rc = cruse_zero_async(loss_sum, sizeof(double), ST(stream), "c_rmse"); if (rc) return rc; }
    hipLaunchKernelGGL(c_rmse_kernel, dim3(eblocks((long long)B * TF, 1024, 1024)), dim3(256), 0, ST(stream), ref, est, B, TF, c,
                       beta, loss_sum, dest);
    CRUSE_LAUNCH_CHECK("c_rmse");
    return CRUSE_OK;
}

extern "C" int cruse_wo_male_spec(const float* ref, const float* est, const float* unproc, int B, long long TF,
                                  long long bstride, long long pstride, float alpha, float beta, float grad_scale,
                                  double* loss_sum, float* dest, void* stream) {
    CRUSE_REQUIRE(B > 0 && TF > 0 && bstride >= TF && pstride >= TF, CRUSE_E_SHAPE, "wo_male_spec: bad shape");
    { int rc = cruse_zero_async(loss_sum, sizeof(double), ST(stream), "wo_male_spec"); if (rc) return rc; }
    hipLaunchKernelGGL(wo_male_spec_kernel, dim3(eblocks((long long)B * TF, 1024, 1024)), dim3(256), 0, ST(stream), ref, est,
                       unproc, B, TF, bstride, pstride, alpha, beta, grad_scale, loss_sum, dest);
    CRUSE_LAUNCH_CHECK("wo_male_spec");
    return CRUSE_OK;
}

extern "C" int cruse_sisnr_plain_finalize(const double* mom, int B, float eps, double* value, float* coef, void* stream) {
    CRUSE_REQUIRE(B > 0, CRUSE_E_SHAPE, "sisnr_plain: B=%d", B);
    { int rc = cruse_zero_async(value, sizeof(double), ST(stream), "sisnr_plain"); if (rc) return rc; }
    hipLaunchKernelGGL(sisnr_plain_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, ST(stream), mom, B, (double)eps, value, coef);
    CRUSE_LAUNCH_CHECK("sisnr_plain");
    return CRUSE_OK;
}

extern "C" int cruse_onepole_fir(const float* x, int B, int L, float a, int taps, float gain, float* y, void* stream) {
    CRUSE_REQUIRE(B > 0 && L > 0 && taps > 0 && taps <= 128 && x != y, CRUSE_E_SHAPE, "onepole_fir: bad arguments (taps <= 128, out of place)");
    hipLaunchKernelGGL(onepole_fir_kernel, dim3((L + 255) / 256, B), dim3(256), 0, ST(stream), x, L, a, taps, gain, y);
    CRUSE_LAUNCH_CHECK("onepole_fir");
    return CRUSE_OK;
}

extern "C" int cruse_fir_causal(const float* x, const float* h, long long h_bstride, int B, int L, int R, float* y, void* stream) {
    CRUSE_REQUIRE(B > 0 && L > 0 && R > 0 && x != y && (h_bstride == 0 || h_bstride >= R), CRUSE_E_SHAPE,
                  "fir_causal: bad arguments (B=%d L=%d R=%d; out of place; tap rows R apart or shared)", B, L, R);
    hipLaunchKernelGGL(fir_causal_kernel, dim3((L + 1023) / 1024, B), dim3(256), 0, ST(stream), x, h, h_bstride, L, R, y);
    CRUSE_LAUNCH_CHECK("fir_causal");
    return CRUSE_OK;
}

extern "C" int cruse_snr_mix(const float* clean, const float* noise, const float* snr_db, int B, int L, float eps,
                             void* scratch, float* clean_out, float* noise_out, float* noisy, void* stream) {
    CRUSE_REQUIRE(B > 0 && L > 0 && clean && noise && snr_db && noisy && scratch, CRUSE_E_SHAPE, "snr_mix: bad arguments");
    CRUSE_REQUIRE(((uintptr_t)scratch & 7) == 0, CRUSE_E_ALIGN, "snr_mix: scratch must be 8-byte aligned");
    double* ss = (double*)scratch;                               // [B][2] f64, then [B][2] u32
    unsigned* st = (unsigned*)(ss + 2 * (size_t)B);
    { int rc = cruse_zero_async(scratch, (size_t)B * 24, ST(stream), "snr_mix"); if (rc) return rc; }
    int chunks = (L + 4095) / 4096;
    if (chunks > 32) chunks = 32;
    hipLaunchKernelGGL(snr_mix_stats_kernel, dim3(chunks, B), dim3(256), 0, ST(stream), clean, noise, L, st, ss);
    CRUSE_LAUNCH_CHECK("snr_mix_stats");
    hipLaunchKernelGGL(snr_mix_apply_kernel, dim3(chunks, B), dim3(256), 0, ST(stream), clean, noise, snr_db, L, st, ss, eps,
                       clean_out, noise_out, noisy);
    CRUSE_LAUNCH_CHECK("snr_mix_apply");
    return CRUSE_OK;
}
