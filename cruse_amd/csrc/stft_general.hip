// Framed DFT with an arbitrary analysis window, and its overlap-add adjoint: the reference's OTHER STFT formulations
// (SURVEY.md 8a rows a5, a6, a16), all of which it writes as dense DFT-basis convolutions:
//   train_base/acoustics/feature.py:272-398   CustomSTFT / CustomISTFT: F.conv1d(x, K, stride=hop, padding=0) with
//        K = rfft(eye(fft)/S_)[:frame_len] * sqrt-Hann, S_ = 0.5*sqrt(fft^2/hop); inverse = conv_transpose1d with the same K
//   train_base/acoustics/conv_stft.py:8-129    STFT: Hamming-windowed full DFT basis, F.conv1d(stride=hop, padding=win-hop),
//        161 bins kept; inverse = conjugate-symmetric extension, conv_transpose1d with basis/win_size, / window sum
//   model/mtfaa.py:8-37                        STFT.transform: torch.stft(nfft, hop, win, hann|hamm window) (center, reflect)
// One kernel pair covers them:
//   X[b,t,f] = scale * sum_{n < win_len} w[n] * x_pad[b, t*hop + n + win_off] * exp(-2 pi i f (n + win_off) / n_fft)
//   x_pad = the clip with `pad` samples in front (pad_mode 0 zeros, 1 reflect)
//   y[b,m]  = post[m] * sum_{t,n : t*hop + n + win_off - pad == m} w[n] * scale * sum_f c_f (re cos - im sin)
//   c_f = 1 (plain adjoint: CustomISTFT) or the Hermitian weights 1,2,..,2,1 (full inverse DFT of a one-sided spectrum)
// The 320-point hot-path STFT keeps its wave-shuffle FFT (stft.hip); these variants are O(N^2) per frame by
// construction in the reference as well (a [2F x win] GEMM per frame), staged through LDS here.
#include "common.h"

namespace {

__device__ __forceinline__ float padded_sample(const float* xb, int L, int idx, int pad_mode) {
    if (idx < 0) {
        if (pad_mode == 0) return 0.f;
        idx = -idx;
    }
    if (idx >= L) {
        if (pad_mode == 0) return 0.f;
        idx = 2 * (L - 1) - idx;
        if (idx < 0) return 0.f;
    }
    return xb[idx];
}

__global__ __launch_bounds__(256) void stft_framed_kernel(const float* __restrict__ wave, const float* __restrict__ window,
                                                          int B, int L, int n_fft, int win_len, int win_off, int hop, int pad,
                                                          int pad_mode, int T, float scale, float* re, float* im) {
    extern __shared__ float sm[];
    float* xs = sm;                 // [win_len] windowed samples
    float* cs = sm + win_len;       // [n_fft] cos
    float* sn = cs + n_fft;         // [n_fft] sin
    const int b = blockIdx.x / T, t = blockIdx.x % T;
    const int nb = n_fft / 2 + 1;
    for (int n = threadIdx.x; n < n_fft; n += 256) {
        float s, c;
        sincospif(2.0f * (float)n / (float)n_fft, &s, &c);
        cs[n] = c; sn[n] = s;
    }
    for (int n = threadIdx.x; n < win_len; n += 256)
        xs[n] = padded_sample(wave + (long long)b * L, L, t * hop + n + win_off - pad, pad_mode) * window[n] * scale;
    __syncthreads();
    const long long row = (long long)b * T + t;
    for (int k = threadIdx.x; k < nb; k += 256) {
        float ar = 0.f, ai = 0.f;
        int idx = (int)(((long long)k * win_off) % n_fft);
        for (int n = 0; n < win_len; ++n) {
            ar += xs[n] * cs[idx];
            ai -= xs[n] * sn[idx];
            idx += k; if (idx >= n_fft) idx -= n_fft;
        }
        re[row * nb + k] = ar;
        im[row * nb + k] = ai;
    }
}

// one block per frame: inverse DFT of the frame into LDS-free registers, windowed, atomically overlap-added.
// out is zero-filled by the host wrapper; post (length L, per output sample) is applied by a second pass.
__global__ __launch_bounds__(256) void istft_framed_kernel(const float* __restrict__ re, const float* __restrict__ im,
                                                           const float* __restrict__ window, int B, int T, int n_fft,
                                                           int win_len, int win_off, int hop, int pad, int L, float scale,
                                                           int hermitian, float* out) {
    extern __shared__ float sm[];
    float* xr = sm;                 // [nb]
    float* xi = sm + (n_fft / 2 + 1);
    float* cs = xi + (n_fft / 2 + 1);
    float* sn = cs + n_fft;
    const int b = blockIdx.x / T, t = blockIdx.x % T;
    const int nb = n_fft / 2 + 1;
    const long long row = (long long)b * T + t;
    for (int n = threadIdx.x; n < n_fft; n += 256) {
        float s, c;
        sincospif(2.0f * (float)n / (float)n_fft, &s, &c);
        cs[n] = c; sn[n] = s;
    }
    for (int k = threadIdx.x; k < nb; k += 256) {
        const float wgt = (hermitian && k > 0 && 2 * k < n_fft) ? 2.f : 1.f;
        xr[k] = re[row * nb + k] * wgt;
        xi[k] = im[row * nb + k] * wgt;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < win_len; n += 256) {
        const int m = t * hop + n + win_off - pad;
        if (m < 0 || m >= L) continue;
        const int ph = n + win_off;
        float acc = 0.f;
        int idx = 0;
        for (int k = 0; k < nb; ++k) {
            acc += xr[k] * cs[idx] - xi[k] * sn[idx];
            idx += ph; if (idx >= n_fft) idx -= n_fft;
        }
        atomicAdd(&out[(long long)b * L + m], acc * window[n] * scale);
    }
}

__global__ void post_scale_kernel(float* out, const float* post, long long n, int L) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] *= post[(int)(i % L)];
}

}  // namespace

extern "C" int cruse_stft_framed(const float* wave, const float* window, int B, int L, int n_fft, int win_len, int win_off,
                                 int hop, int pad, int pad_mode, int T, float scale, float* re, float* im, void* stream) {
    CRUSE_REQUIRE(B > 0 && L > 0 && T > 0 && hop > 0 && n_fft >= 2 && n_fft % 2 == 0 && n_fft <= 4096, CRUSE_E_SHAPE,
                  "stft_framed: bad shape B=%d L=%d T=%d n_fft=%d hop=%d", B, L, T, n_fft, hop);
    CRUSE_REQUIRE(win_len > 0 && win_off >= 0 && win_len + win_off <= n_fft, CRUSE_E_SHAPE,
                  "stft_framed: window of %d at offset %d does not fit n_fft=%d", win_len, win_off, n_fft);
    CRUSE_REQUIRE(pad >= 0 && (pad_mode == 0 || pad_mode == 1) && (pad_mode == 0 || pad < L), CRUSE_E_SHAPE, "stft_framed: bad padding");
    const size_t lds = (size_t)(win_len + 2 * n_fft) * sizeof(float);
    { int rc = cruse_ensure_dyn_lds((const void*)stft_framed_kernel, lds, "stft_framed"); if (rc) return rc; }
    hipLaunchKernelGGL(stft_framed_kernel, dim3(B * T), dim3(256), lds, (hipStream_t)stream, wave, window, B, L, n_fft, win_len,
                       win_off, hop, pad, pad_mode, T, scale, re, im);
    CRUSE_LAUNCH_CHECK("stft_framed");
    return CRUSE_OK;
}

extern "C" int cruse_istft_framed(const float* re, const float* im, const float* window, const float* post, int B, int T,
                                  int n_fft, int win_len, int win_off, int hop, int pad, int L, float scale, int hermitian,
                                  float* out, void* stream) {
    CRUSE_REQUIRE(B > 0 && L > 0 && T > 0 && hop > 0 && n_fft >= 2 && n_fft % 2 == 0 && n_fft <= 4096, CRUSE_E_SHAPE,
                  "istft_framed: bad shape B=%d L=%d T=%d n_fft=%d hop=%d", B, L, T, n_fft, hop);
    CRUSE_REQUIRE(win_len > 0 && win_off >= 0 && win_len + win_off <= n_fft && pad >= 0, CRUSE_E_SHAPE, "istft_framed: bad window");
    hipStream_t st = (hipStream_t)stream;
    { int rc = cruse_zero_async(out, (size_t)B * L * sizeof(float), st, "istft_framed"); if (rc) return rc; }
    const size_t lds = (size_t)(2 * (n_fft / 2 + 1) + 2 * n_fft) * sizeof(float);
    { int rc = cruse_ensure_dyn_lds((const void*)istft_framed_kernel, lds, "istft_framed"); if (rc) return rc; }
    hipLaunchKernelGGL(istft_framed_kernel, dim3(B * T), dim3(256), lds, st, re, im, window, B, T, n_fft, win_len, win_off, hop,
                       pad, L, scale, hermitian, out);
    CRUSE_LAUNCH_CHECK("istft_framed");
    if (post) {
        const long long n = (long long)B * L;
        long long g = (n + 1023) / 1024; if (g > 4096) g = 4096;
        hipLaunchKernelGGL(post_scale_kernel, dim3((unsigned)g), dim3(256), 0, st, out, post, n, L);
        CRUSE_LAUNCH_CHECK("istft_framed post");
    }
    return CRUSE_OK;
}
