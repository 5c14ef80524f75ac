// STFT / iSTFT framing for 16 kHz CRUSE features on gfx950.
//
// Replaces torch.stft / torch.istft at train_base/acoustics/feature.py:22-30,53-61 and
// utils/utils.py:390-401,448-454 (periodic Hann(n_fft), center=True, reflect pad n_fft/2,
// one-sided spectrum, no normalisation; inverse divides by the window-square envelope).
//
// n_fft = 320 = 5 x 64: one 64-lane wavefront owns one frame.  Lane l holds samples
// l, l+64, ..., l+256; a radix-5 butterfly runs in registers, then five 64-point
// decimation-in-frequency FFTs run ACROSS the lanes with __shfl_xor butterflies
// (6 stages).  Samples of a 16-frame tile are staged once in LDS (each HBM sample is
// read once although frames overlap 50 %), spectra leave through LDS so the global
// stores are contiguous [frame][bin] rows.
#include "common.h"

namespace {

constexpr int NFFT = 320;
constexpr int NBIN = NFFT / 2 + 1;   // 161
constexpr int FPB = 16;              // frames per workgroup
constexpr int MAXSPAN = (FPB - 1) * NFFT + NFFT;

__device__ __forceinline__ int bitrev6(int l) { return (int)(__brev((unsigned)l) >> 26); }

// In-place 320-point complex DFT over one wavefront.
// in : x[j] = element 64*j + lane (j = 0..4)
// out: x[k1] = bin k1 + 5*bitrev6(lane)
// SIGN = -1: forward e^{-i...}; +1: inverse (unnormalised).  tw[m] = (cos, sin)(2*pi*m/320).
template <int SIGN>
__device__ __forceinline__ void fft320(float (&xr)[5], float (&xi)[5], const float2* tw, int lane) {
    const float c1 = 0.30901699437494745f, c2 = -0.8090169943749475f;
    const float s1 = 0.9510565162951535f, s2 = 0.5877852522924731f;
    // radix-5 (in registers)
    {
        const float ar = xr[1] + xr[4], ai = xi[1] + xi[4];
        const float br = xr[2] + xr[3], bi = xi[2] + xi[3];
        const float dr = xr[1] - xr[4], di = xi[1] - xi[4];
        const float er = xr[2] - xr[3], ei = xi[2] - xi[3];
        const float y0r = xr[0] + ar + br, y0i = xi[0] + ai + bi;
        const float p1r = xr[0] + c1 * ar + c2 * br, p1i = xi[0] + c1 * ai + c2 * bi;
        const float p2r = xr[0] + c2 * ar + c1 * br, p2i = xi[0] + c2 * ai + c1 * bi;
        // q = s1*d + s2*e ; r = s2*d - s1*e ; forward: Y1 = p1 - i q, Y4 = p1 + i q, Y2 = p2 - i r, Y3 = p2 + i r
        const float q1r = s1 * dr + s2 * er, q1i = s1 * di + s2 * ei;
        const float q2r = s2 * dr - s1 * er, q2i = s2 * di - s1 * ei;
        const float sg = (SIGN < 0) ? 1.f : -1.f;   // -i*q = (q_i, -q_r) for forward
        xr[0] = y0r; xi[0] = y0i;
        xr[1] = p1r + sg * q1i; xi[1] = p1i - sg * q1r;
        xr[4] = p1r - sg * q1i; xi[4] = p1i + sg * q1r;
        xr[2] = p2r + sg * q2i; xi[2] = p2i - sg * q2r;
        xr[3] = p2r - sg * q2i; xi[3] = p2i + sg * q2r;
    }
    // twiddle W320^{SIGN * lane * k1}
#pragma unroll
    for (int k1 = 1; k1 < 5; ++k1) {
        const float2 t = tw[lane * k1];
        const float c = t.x, s = (SIGN < 0) ? -t.y : t.y;   // multiply by (c + i s)
        const float r = xr[k1] * c - xi[k1] * s;
        const float i = xr[k1] * s + xi[k1] * c;
        xr[k1] = r; xi[k1] = i;
    }
    // five 64-point DIF FFTs across the lanes
#pragma unroll
    for (int h = 32; h >= 1; h >>= 1) {
        const bool upper = (lane & h) == 0;
        const float2 t = tw[(lane & (h - 1)) * (160 / h)];
        const float c = t.x, s = (SIGN < 0) ? -t.y : t.y;
#pragma unroll
        for (int k1 = 0; k1 < 5; ++k1) {
            const float orr = __shfl_xor(xr[k1], h, 64);
            const float oi = __shfl_xor(xi[k1], h, 64);
            if (upper) {
                xr[k1] += orr; xi[k1] += oi;
            } else {
                const float dr = orr - xr[k1], di = oi - xi[k1];
                xr[k1] = dr * c - di * s;
                xi[k1] = dr * s + di * c;
            }
        }
    }
}

__device__ __forceinline__ void init_tables(float* win, float2* tw, int tid, int nthreads) {
    for (int n = tid; n < NFFT; n += nthreads) {
        float s, c;
        sincospif((float)n / 160.0f, &s, &c);
        tw[n] = make_float2(c, s);
        win[n] = 0.5f - 0.5f * c;          // periodic Hann: 0.5 - 0.5 cos(2 pi n / N)
    }
}

// MODE 0: STFT (reflect padding).  MODE 1: adjoint of iSTFT (input = dwave, zero padding,
// pre-divided by the window-square envelope, outputs scaled by c_k / N).
template <int MODE>
__global__ __launch_bounds__(256) void stft320_kernel(const float* __restrict__ wave, int B, int L, int hop, int T,
                                                      float* re, float* im, float* mag, int mag_bins, float mag_eps) {
    __shared__ float s_x[MAXSPAN];
    __shared__ float s_win[NFFT];
    __shared__ float2 s_tw[NFFT];
    __shared__ float s_out[4][2][NBIN + 3];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ntile = (T + FPB - 1) / FPB;
    const int b = blockIdx.x / ntile;
    const int t0 = (blockIdx.x % ntile) * FPB;
    const int nf = min(FPB, T - t0);
    init_tables(s_win, s_tw, tid, 256);
    __syncthreads();
    const int span = (nf - 1) * hop + NFFT;
    const int p0 = t0 * hop;
    for (int i = tid; i < span; i += 256) {
        const int p = p0 + i;
        int o = p - NFFT / 2;
        float v;
        if (MODE == 0) {
            if (o < 0) o = -o;
            if (o >= L) o = 2 * (L - 1) - o;
            v = wave[(long long)b * L + o];
        } else {
            v = 0.f;
            if (o >= 0 && o < L) {
                float env = 0.f;
                const int tlo = max(0, (p - NFFT + hop) / hop), thi = min(T - 1, p / hop);
                for (int t = tlo; t <= thi; ++t) { const float w = s_win[p - t * hop]; env += w * w; }
                v = wave[(long long)b * L + o] / env;
            }
        }
        s_x[i] = v;
    }
    __syncthreads();
    for (int it = 0; it < FPB / 4; ++it) {
        const int f = it * 4 + wv;
        const bool valid = f < nf;
        float xr[5], xi[5];
        if (valid) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int n = lane + 64 * j;
                xr[j] = s_x[f * hop + n] * s_win[n];
                xi[j] = 0.f;
            }
            fft320<-1>(xr, xi, s_tw, lane);
            const int k2 = bitrev6(lane);
#pragma unroll
            for (int k1 = 0; k1 < 5; ++k1) {
                const int k = k1 + 5 * k2;
                if (k < NBIN) {
                    float sc = 1.f;
                    if (MODE == 1) sc = ((k == 0 || k == NFFT / 2) ? 1.f : 2.f) / (float)NFFT;
                    s_out[wv][0][k] = xr[k1] * sc;
                    s_out[wv][1][k] = (MODE == 1 && (k == 0 || k == NFFT / 2)) ? 0.f : xi[k1] * sc;
                }
            }
        }
        __syncthreads();
        if (valid) {
            const long long row = (long long)b * T + t0 + f;
            for (int k = lane; k < NBIN; k += 64) {
                const float r = s_out[wv][0][k], i = s_out[wv][1][k];
                if (re) re[row * NBIN + k] = r;
                if (im) im[row * NBIN + k] = i;
                if (mag && k < mag_bins) mag[row * mag_bins + k] = sqrtf(r * r + i * i + mag_eps);
            }
        }
        __syncthreads();
    }
}

// iSTFT: each workgroup produces TS = (FPB-1)*hop output samples (padded coordinates) and
// recomputes the (at most n_fft/hop) halo frames that overlap them.
__global__ __launch_bounds__(256) void istft320_kernel(const float* __restrict__ re, const float* __restrict__ im,
                                                       int B, int T, int hop, int L, float* wave) {
    __shared__ float s_y[MAXSPAN];
    __shared__ float s_w[MAXSPAN];
    __shared__ float s_win[NFFT];
    __shared__ float2 s_tw[NFFT];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int TS = (FPB - 1) * hop;
    const int total = NFFT / 2 + L;                       // padded coordinates needed: [n_fft/2, n_fft/2 + L)
    const int ntile = (L + TS - 1) / TS;
    const int b = blockIdx.x / ntile;
    const int s0 = NFFT / 2 + (blockIdx.x % ntile) * TS;  // first padded sample of this tile
    const int s1 = min(s0 + TS, total);
    init_tables(s_win, s_tw, tid, 256);
    for (int i = tid; i < TS; i += 256) { s_y[i] = 0.f; s_w[i] = 0.f; }
    __syncthreads();
    // frames t with [t*hop, t*hop+NFFT) intersecting [s0, s1)
    const int tlo = max(0, (s0 - NFFT + hop) / hop);      // ceil((s0-NFFT+1)/hop) for s0>=NFFT-1
    const int thi = min(T - 1, (s1 - 1) / hop);
    for (int t = tlo + wv; t <= thi; t += 4) {
        const long long row = (long long)b * T + t;
        float xr[5], xi[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int m = lane + 64 * j;
            const int k = m <= NFFT / 2 ? m : NFFT - m;
            float r = re[row * NBIN + k], i = im[row * NBIN + k];
            if (k == 0 || k == NFFT / 2) i = 0.f;        // c2r ignores these imaginary parts
            xr[j] = r;
            xi[j] = m <= NFFT / 2 ? i : -i;
        }
        fft320<1>(xr, xi, s_tw, lane);
        const int k2 = bitrev6(lane);
#pragma unroll
        for (int k1 = 0; k1 < 5; ++k1) {
            const int n = k1 + 5 * k2;
            const int p = t * hop + n;
            if (p >= s0 && p < s1) {
                const float w = s_win[n];
                atomicAdd(&s_y[p - s0], xr[k1] * (1.0f / NFFT) * w);
                atomicAdd(&s_w[p - s0], w * w);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < s1 - s0; i += 256) {
        const float env = s_w[i];
        wave[(long long)b * L + (s0 - NFFT / 2 + i)] = env > 1e-11f ? s_y[i] / env : 0.f;
    }
}

// generic fallback: direct DFT, one workgroup per frame
__global__ __launch_bounds__(256) void stft_dft_kernel(const float* __restrict__ wave, int B, int L, int n_fft, int hop,
                                                       int T, float* re, float* im, float* mag, int mag_bins,
                                                       float mag_eps) {
    extern __shared__ float sm[];
    float* xs = sm;                 // [n_fft] windowed samples
    float* cs = sm + n_fft;         // [n_fft] cos
    float* sn = cs + n_fft;         // [n_fft] sin
    const int b = blockIdx.x / T, t = blockIdx.x % T;
    const int nb = n_fft / 2 + 1;
    for (int n = threadIdx.x; n < n_fft; n += 256) {
        float s, c;
        sincospif(2.0f * (float)n / (float)n_fft, &s, &c);
        cs[n] = c; sn[n] = s;
        int o = t * hop + n - n_fft / 2;
        if (o < 0) o = -o;
        if (o >= L) o = 2 * (L - 1) - o;
        xs[n] = wave[(long long)b * L + o] * (0.5f - 0.5f * c);
    }
    __syncthreads();
    const long long row = (long long)b * T + t;
    for (int k = threadIdx.x; k < nb; k += 256) {
        float ar = 0.f, ai = 0.f;
        int idx = 0;
        for (int n = 0; n < n_fft; ++n) {
            ar += xs[n] * cs[idx];
            ai -= xs[n] * sn[idx];
            idx += k; if (idx >= n_fft) idx -= n_fft;
        }
        if (re) re[row * nb + k] = ar;
        if (im) im[row * nb + k] = ai;
        if (mag && k < mag_bins) mag[row * mag_bins + k] = sqrtf(ar * ar + ai * ai + mag_eps);
    }
}

}  // namespace

extern "C" int cruse_stft_fwd(const float* wave, int B, int L, int n_fft, int hop,
                              float* re, float* im, float* mag, int mag_bins, float mag_eps, void* stream) {
    CRUSE_REQUIRE(B > 0 && L > 0 && hop > 0, CRUSE_E_SHAPE, "stft: bad shape B=%d L=%d hop=%d", B, L, hop);
    CRUSE_REQUIRE(n_fft >= 2 && n_fft % 2 == 0 && n_fft <= 2048, CRUSE_E_SHAPE, "stft: n_fft=%d must be even and <= 2048", n_fft);
    CRUSE_REQUIRE(L > n_fft / 2, CRUSE_E_SHAPE, "stft: reflect padding needs L=%d > n_fft/2=%d", L, n_fft / 2);
    CRUSE_REQUIRE(mag == nullptr || (mag_bins > 0 && mag_bins <= n_fft / 2 + 1), CRUSE_E_SHAPE, "stft: mag_bins=%d", mag_bins);
    const int T = 1 + L / hop;
    if (n_fft == NFFT && hop <= NFFT) {
        hipLaunchKernelGGL(stft320_kernel<0>, dim3(B * cdiv(T, FPB)), dim3(256), 0, (hipStream_t)stream, wave, B, L,
                           hop, T, re, im, mag, mag_bins, mag_eps);
    } else {
        hipLaunchKernelGGL(stft_dft_kernel, dim3(B * T), dim3(256), 3 * n_fft * sizeof(float), (hipStream_t)stream,
                           wave, B, L, n_fft, hop, T, re, im, mag, mag_bins, mag_eps);
    }
    CRUSE_LAUNCH_CHECK("stft");
    return CRUSE_OK;
}

extern "C" int cruse_istft_fwd(const float* re, const float* im, int B, int T, int n_fft, int hop, int L,
                               float* wave, void* stream) {
    CRUSE_REQUIRE(B > 0 && T > 0 && L > 0, CRUSE_E_SHAPE, "istft: bad shape");
    CRUSE_REQUIRE(n_fft == NFFT && hop > 0 && hop <= NFFT, CRUSE_E_SHAPE, "istft: only n_fft=320, hop<=320 (got %d, %d)", n_fft, hop);
    CRUSE_REQUIRE(L <= n_fft + hop * (T - 1) - n_fft / 2, CRUSE_E_SHAPE, "istft: length %d exceeds the %d frames", L, T);
    const int TS = (FPB - 1) * hop;
    hipLaunchKernelGGL(istft320_kernel, dim3(B * cdiv(L, TS)), dim3(256), 0, (hipStream_t)stream, re, im, B, T, hop, L, wave);
    CRUSE_LAUNCH_CHECK("istft");
    return CRUSE_OK;
}

extern "C" int cruse_istft_bwd(const float* dwave, int B, int T, int n_fft, int hop, int L,
                               float* dre, float* dim, void* stream) {
    CRUSE_REQUIRE(B > 0 && T > 0 && L > 0, CRUSE_E_SHAPE, "istft_bwd: bad shape");
    CRUSE_REQUIRE(n_fft == NFFT && hop > 0 && hop <= NFFT, CRUSE_E_SHAPE, "istft_bwd: only n_fft=320, hop<=320");
    hipLaunchKernelGGL(stft320_kernel<1>, dim3(B * cdiv(T, FPB)), dim3(256), 0, (hipStream_t)stream, dwave, B, L, hop, T,
                       dre, dim, (float*)nullptr, 0, 0.f);
    CRUSE_LAUNCH_CHECK("istft_bwd");
    return CRUSE_OK;
}
