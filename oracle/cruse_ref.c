/* oracle/cruse_ref.c -- plain-C CPU twins of 33 entry points of include/cruse_hip.h ("every entry also has
 * a *_ref plain-C++ CPU twin ... used for ABI-level parity tests").
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT: only tests/ loads this library (tests/test_oracle.py pins it against torch's ops -- the library
 * calls the reference makes at the cited sites -- and against the committed golden vectors; tests/test_gpu_abi_ref.py compares
 * every HIP entry point with its twin through the C ABI).  Nothing under cruse_amd/ links, loads or calls it.
 *
 * A twin has the signature of its entry point with HOST pointers (the stream / scratch arguments are ignored), the same
 * layouts (frame-major [B,T,C,F] activations), the same "+=" conventions and the same return codes.  Arithmetic: the sums are
 * formed in double and rounded once -- the twins state WHAT an entry point computes, the tests state how far the f32 / bf16
 * kernels may be from it.  Each function cites the reference lines the entry point replaces.
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/libcruse_ref.so oracle/cruse_ref.c -lm   (__graft_entry__.build() does this)
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REF_OK 0
#define REF_E_SHAPE -1
#define REF_E_DTYPE -3
#define PREC_BF16 2
#define DT_F32 0

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

int cruse_ref_abi_version(void) { return 12; } /* the CRUSE_ABI_VERSION these twins were written against */

static uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u); /* NaN */
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_round(float f) {
    uint32_t u = (uint32_t)bf16_rne(f) << 16;
    float r;
    memcpy(&r, &u, 4);
    return r;
}

/* ---- acoustic front end ------------------------------------------------------------------------------------------------ */

/* torch.stft(y, n_fft, hop, win = n_fft, hann_window(n_fft), center = True (reflect), return_complex = True)
 * (train_base/acoustics/feature.py:22-30) + the [B,T,F] transposition and magnitude of utils/utils.py:397-400 */
int cruse_stft_fwd_ref(const float* wave, int B, int L, int n_fft, int hop, float* re, float* im, float* mag, int mag_bins,
                       float mag_eps, void* stream) {
    (void)stream;
    if (B <= 0 || L <= n_fft / 2 || n_fft <= 0 || (n_fft & 1) || hop <= 0) return REF_E_SHAPE;
    const int Fb = n_fft / 2 + 1, T = 1 + L / hop, half = n_fft / 2;
    double* win = (double*)malloc(sizeof(double) * n_fft);
    double* fr = (double*)malloc(sizeof(double) * n_fft);
    for (int n = 0; n < n_fft; ++n) win[n] = 0.5 - 0.5 * cos(2.0 * M_PI * n / n_fft); /* periodic Hann (torch.hann_window default) */
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) {
            for (int n = 0; n < n_fft; ++n) {
                long long i = (long long)t * hop + n - half;
                if (i < 0) i = -i;                       /* reflect (no edge repeat) */
                if (i >= L) i = 2ll * (L - 1) - i;
                fr[n] = (i >= 0 && i < L) ? win[n] * (double)wave[(long long)b * L + i] : 0.0;
            }
            for (int k = 0; k < Fb; ++k) {
                double sr = 0.0, si = 0.0;
                for (int n = 0; n < n_fft; ++n) {
                    const double ph = -2.0 * M_PI * (double)(((long long)k * n) % n_fft) / n_fft;
                    sr += fr[n] * cos(ph);
                    si += fr[n] * sin(ph);
                }
                const long long o = ((long long)b * T + t) * Fb + k;
                if (re) re[o] = (float)sr;
                if (im) im[o] = (float)si;
                if (mag && k < mag_bins) mag[((long long)b * T + t) * mag_bins + k] = (float)sqrt(sr * sr + si * si + (double)mag_eps);
            }
        }
    free(win);
    free(fr);
    return REF_OK;
}

/* torch.istft(X, n_fft, hop, win = n_fft, hann_window(n_fft), center = True, length = L) (feature.py:53-61, utils/utils.py:448-454) */
int cruse_istft_fwd_ref(const float* re, const float* im, int B, int T, int n_fft, int hop, int L, float* wave, void* stream) {
    (void)stream;
    if (B <= 0 || T <= 0 || n_fft <= 0 || (n_fft & 1) || hop <= 0 || L <= 0) return REF_E_SHAPE;
    const int Fb = n_fft / 2 + 1, half = n_fft / 2;
    const long long full = (long long)(T - 1) * hop + n_fft;
    double* win = (double*)malloc(sizeof(double) * n_fft);
    double* ola = (double*)malloc(sizeof(double) * full);
    double* env = (double*)malloc(sizeof(double) * full);
    for (int n = 0; n < n_fft; ++n) win[n] = 0.5 - 0.5 * cos(2.0 * M_PI * n / n_fft);
    for (int b = 0; b < B; ++b) {
        for (long long i = 0; i < full; ++i) { ola[i] = 0.0; env[i] = 0.0; }
        for (int t = 0; t < T; ++t) {
            const float* xr = re + ((long long)b * T + t) * Fb;
            const float* xi = im + ((long long)b * T + t) * Fb;
            for (int n = 0; n < n_fft; ++n) {
                /* irfft: the imaginary parts of bins 0 and n_fft/2 are ignored */
                double s = (double)xr[0] + ((n & 1) ? -1.0 : 1.0) * (double)xr[half];
                for (int k = 1; k < half; ++k) {
                    const double ph = 2.0 * M_PI * (double)(((long long)k * n) % n_fft) / n_fft;
                    s += 2.0 * ((double)xr[k] * cos(ph) - (double)xi[k] * sin(ph));
                }
                ola[(long long)t * hop + n] += s / n_fft * win[n];
                env[(long long)t * hop + n] += win[n] * win[n];
            }
        }
        for (int i = 0; i < L; ++i) {
            const long long j = (long long)i + half;
            wave[(long long)b * L + i] = (j < full && env[j] > 1e-11) ? (float)(ola[j] / env[j]) : 0.f;
        }
    }
    free(win);
    free(ola);
    free(env);
    return REF_OK;
}

/* ---- convolutions (nn.Conv2d / nn.ConvTranspose2d at model/cruse_net.py:138-143,149-164) ------------------------------------ */

static double act_of(double v, int act) { return act == 1 ? 1.0 / (1.0 + exp(-v)) : v; }

/* gather form (see cruse_conv_gather): Conv2d((KT,3), stride (1,S)) with causal time padding KT-1 and frequency padding pad */
int cruse_conv_gather_ref(const float* x, const float* w, const float* bias, float* y, int B, int T, int Cin, int Fin, int Cout,
                          int Fout, int KT, int S, int pad, int w_layout, int act, int accum, int prec, int x_dtype, int y_dtype,
                          void* stream) {
    (void)stream; (void)prec;
    if (x_dtype != DT_F32 || y_dtype != DT_F32) return REF_E_DTYPE;
    if (B <= 0 || T <= 0 || Cin <= 0 || Cout <= 0 || Fin <= 0 || Fout <= 0 || KT <= 0 || S <= 0) return REF_E_SHAPE;
    if (w_layout == 1 && (KT != 1 || S != 1)) return REF_E_SHAPE;
    if (accum && act) return REF_E_SHAPE;
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t)
            for (int co = 0; co < Cout; ++co)
                for (int fo = 0; fo < Fout; ++fo) {
                    double s = bias ? (double)bias[co] : 0.0;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int kt = 0; kt < KT; ++kt) {
                            const int ts = t - (KT - 1) + kt;
                            if (ts < 0) continue;
                            for (int kf = 0; kf < 3; ++kf) {
                                const int fi = fo * S - pad + kf;
                                if (fi < 0 || fi >= Fin) continue;
                                const double wv = w_layout == 0 ? (double)w[(((long long)co * Cin + ci) * KT + kt) * 3 + kf]
                                                                : (double)w[((long long)ci * Cout + co) * 3 + (2 - kf)];
                                s += wv * (double)x[(((long long)b * T + ts) * Cin + ci) * Fin + fi];
                            }
                        }
                    const long long o = (((long long)b * T + t) * Cout + co) * Fout + fo;
                    y[o] = accum ? (float)((double)y[o] + s) : (float)act_of(s, act);
                }
    return REF_OK;
}

/* scatter form with frequency stride 2 (see cruse_conv_scatter2): ConvTranspose2d((1,3), stride (1,2)) with the [..., :-1] crop
 * (cruse_net.py:161-164; KT = 1, pad = 0) and the backward-data of the (2,3)/(1,2) encoder conv (KT = 2, pad = 1) */
int cruse_conv_scatter2_ref(const float* g, const float* w, const float* bias, float* y, int B, int T, int Cs, int Fg, int Cout,
                            int Fout, int KT, int pad, int act, int accum, int prec, int x_dtype, int y_dtype, void* stream) {
    (void)stream; (void)prec;
    if (x_dtype != DT_F32 || y_dtype != DT_F32) return REF_E_DTYPE;
    if (B <= 0 || T <= 0 || Cs <= 0 || Cout <= 0 || Fg <= 0 || Fout <= 0 || KT <= 0) return REF_E_SHAPE;
    if (accum && act) return REF_E_SHAPE;
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t)
            for (int co = 0; co < Cout; ++co)
                for (int fo = 0; fo < Fout; ++fo) {
                    double s = bias ? (double)bias[co] : 0.0;
                    for (int cs = 0; cs < Cs; ++cs)
                        for (int kt = 0; kt < KT; ++kt) {
                            const int ts = t + (KT - 1) - kt;
                            if (ts < 0 || ts >= T) continue;
                            for (int kf = 0; kf < 3; ++kf) {
                                const int q = fo + pad - kf;
                                if (q < 0 || (q & 1) || q / 2 >= Fg) continue;
                                s += (double)w[(((long long)cs * Cout + co) * KT + kt) * 3 + kf] *
                                     (double)g[(((long long)b * T + ts) * Cs + cs) * Fg + q / 2];
                            }
                        }
                    const long long o = (((long long)b * T + t) * Cout + co) * Fout + fo;
                    y[o] = accum ? (float)((double)y[o] + s) : (float)act_of(s, act);
                }
    return REF_OK;
}

/* weight gradient of either form (see cruse_conv_wgrad): dw[ca][cb][kt][kf] += sum a[b,t,ca,fa] bt[b, t-(KT-1)+kt, cb, fa*S - pad + kf] */
int cruse_conv_wgrad_ref(const float* a, const float* bt, float* dw, int B, int T, int Ca, int Fa, int Cb, int Fb, int KT, int S,
                         int pad, int prec, int a_dtype, int bt_dtype, void* ws, void* stream) {
    (void)stream; (void)ws; (void)prec;
    if (a_dtype != DT_F32 || bt_dtype != DT_F32) return REF_E_DTYPE;
    if (B <= 0 || T <= 0 || Ca <= 0 || Cb <= 0 || Fa <= 0 || Fb <= 0 || KT <= 0 || S <= 0) return REF_E_SHAPE;
    for (int ca = 0; ca < Ca; ++ca)
        for (int cb = 0; cb < Cb; ++cb)
            for (int kt = 0; kt < KT; ++kt)
                for (int kf = 0; kf < 3; ++kf) {
                    double s = 0.0;
                    for (int b = 0; b < B; ++b)
                        for (int t = 0; t < T; ++t) {
                            const int ts = t - (KT - 1) + kt;
                            if (ts < 0) continue;
                            for (int fa = 0; fa < Fa; ++fa) {
                                const int fb = fa * S - pad + kf;
                                if (fb < 0 || fb >= Fb) continue;
                                s += (double)a[(((long long)b * T + t) * Ca + ca) * Fa + fa] *
                                     (double)bt[(((long long)b * T + ts) * Cb + cb) * Fb + fb];
                            }
                        }
                    float* d = dw + (((long long)ca * Cb + cb) * KT + kt) * 3 + kf;
                    *d = (float)((double)*d + s);
                }
    return REF_OK;
}

/* out[c] += sum_{rows,f} g[row,c,f] (bias gradients) */
int cruse_channel_sum_ref(const float* g, long long rows, int C, int F, float* out, void* stream) {
    (void)stream;
    if (rows <= 0 || C <= 0 || F <= 0) return REF_E_SHAPE;
    for (int c = 0; c < C; ++c) {
        double s = 0.0;
        for (long long r = 0; r < rows; ++r)
            for (int f = 0; f < F; ++f) s += (double)g[(r * C + c) * F + f];
        out[c] = (float)((double)out[c] + s);
    }
    return REF_OK;
}

/* ---- BatchNorm2d (+ReLU, + skip add) (cruse_net.py:141-142,149-152,161-163) -------------------------------------------------- */

int cruse_bn_stats_ref(const float* y, long long rows, int C, int F, double* sums, int zeroed, void* stream) {
    (void)stream;
    if (rows <= 0 || C <= 0 || F <= 0) return REF_E_SHAPE;
    if (!zeroed) memset(sums, 0, sizeof(double) * 2 * C);
    for (long long r = 0; r < rows; ++r)
        for (int c = 0; c < C; ++c)
            for (int f = 0; f < F; ++f) {
                const double v = (double)y[(r * C + c) * F + f];
                sums[c] += v;
                sums[C + c] += v * v;
            }
    return REF_OK;
}

/* training-mode statistics of torch.nn.BatchNorm2d: biased variance for the normalisation, unbiased for the running estimate */
int cruse_bn_finalize_ref(const double* sums, long long count, int C, float eps, float momentum, float* mean, float* rstd,
                          float* running_mean, float* running_var, void* stream) {
    (void)stream;
    if (count <= 0 || C <= 0) return REF_E_SHAPE;
    for (int c = 0; c < C; ++c) {
        const double m = sums[c] / (double)count;
        double var = sums[C + c] / (double)count - m * m;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean && running_var) {
            const double unb = count > 1 ? var * (double)count / (double)(count - 1) : var;
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * m);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
        }
    }
    return REF_OK;
}

int cruse_bn_act_fwd_ref(const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                         const float* skip, float* out, long long rows, int C, int F, int relu, void* stream) {
    (void)stream;
    if (rows <= 0 || C <= 0 || F <= 0) return REF_E_SHAPE;
    for (long long r = 0; r < rows; ++r)
        for (int c = 0; c < C; ++c)
            for (int f = 0; f < F; ++f) {
                const long long o = (r * C + c) * F + f;
                double v = ((double)y[o] - (double)mean[c]) * (double)rstd[c] * (double)gamma[c] + (double)beta[c];
                if (relu && v < 0.0) v = 0.0;
                if (skip) v += (double)skip[o];
                out[o] = (float)v;
            }
    return REF_OK;
}

/* sums[0..C) = sum g, sums[C..2C) = sum g * xhat with g = dout * [bn(y) > 0] */
int cruse_bn_act_bwd_reduce_ref(const float* dout, const float* y, const float* mean, const float* rstd, const float* gamma,
                                const float* beta, long long rows, int C, int F, int relu, double* sums, int zeroed, void* stream) {
    (void)stream;
    if (rows <= 0 || C <= 0 || F <= 0) return REF_E_SHAPE;
    if (!zeroed) memset(sums, 0, sizeof(double) * 2 * C);
    for (long long r = 0; r < rows; ++r)
        for (int c = 0; c < C; ++c)
            for (int f = 0; f < F; ++f) {
                const long long o = (r * C + c) * F + f;
                const double xh = ((double)y[o] - (double)mean[c]) * (double)rstd[c];
                const double z = xh * (double)gamma[c] + (double)beta[c];
                const double g = (relu && !(z > 0.0)) ? 0.0 : (double)dout[o];
                sums[c] += g;
                sums[C + c] += g * xh;
            }
    return REF_OK;
}

/* dy = gamma rstd (g - [training](sum_g + xhat sum_gx) / count); dgamma += sum_gx; dbeta += sum_g; dbias += sum dy (closed form) */
int cruse_bn_act_bwd_apply_ref(const float* dout, const float* y, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, const double* sums, int sum_replicas, long long rows, int C, int F, int relu,
                               int training, int dout_dtype, void* dy_, int dy_dtype, float* dgamma, float* dbeta, float* dbias,
                               void* stream) {
    (void)stream;
    if (dout_dtype != DT_F32 || dy_dtype != DT_F32) return REF_E_DTYPE;
    if (rows <= 0 || C <= 0 || F <= 0 || sum_replicas <= 0) return REF_E_SHAPE;
    float* dy = (float*)dy_;
    const double count = (double)rows * (double)F;
    for (int c = 0; c < C; ++c) {
        double sg = 0.0, sgx = 0.0;
        for (int q = 0; q < sum_replicas; ++q) { sg += sums[(size_t)q * 2 * C + c]; sgx += sums[(size_t)q * 2 * C + C + c]; }
        for (long long r = 0; r < rows; ++r)
            for (int f = 0; f < F; ++f) {
                const long long o = (r * C + c) * F + f;
                const double xh = ((double)y[o] - (double)mean[c]) * (double)rstd[c];
                const double z = xh * (double)gamma[c] + (double)beta[c];
                double g = (relu && !(z > 0.0)) ? 0.0 : (double)dout[o];
                if (training) g -= (sg + xh * sgx) / count;
                dy[o] = (float)((double)gamma[c] * (double)rstd[c] * g);
            }
        if (dgamma) dgamma[c] = (float)((double)dgamma[c] + sgx);
        if (dbeta) dbeta[c] = (float)((double)dbeta[c] + sg);
        if (dbias && !training) dbias[c] = (float)((double)dbias[c] + (double)gamma[c] * (double)rstd[c] * sg);
    }
    return REF_OK;
}

/* ---- LayerNorm (+ group interleave, + residual) (cruse_net.py:32-33,43-51,160) ------------------------------------------------ */

static int ln_perm(int c, int H, int g) { const int Hg = H / g; return (c % Hg) * g + c / Hg; } /* P(i*Hg + j) = j*g + i */

int cruse_ln_fwd_ref(const float* x, const float* gamma, const float* beta, const float* res, float* y, void* y_bf16, float* mean,
                     float* rstd, long long rows, int H, int interleave_g, float eps, int seg_len, long long seg_stride,
                     long long seg_off, void* stream) {
    (void)stream; (void)seg_stride; (void)seg_off;
    if (rows <= 0 || H <= 0 || interleave_g <= 0 || H % interleave_g) return REF_E_SHAPE;
    if (seg_len > 0) return REF_E_SHAPE; /* (the row-segment form is a scheduling device of the engine: not restated) */
    uint16_t* yb = (uint16_t*)y_bf16;
    for (long long r = 0; r < rows; ++r) {
        double m = 0.0, v = 0.0;
        for (int c = 0; c < H; ++c) m += (double)x[r * H + c];
        m /= H;
        for (int c = 0; c < H; ++c) { const double d = (double)x[r * H + c] - m; v += d * d; }
        const double rs = 1.0 / sqrt(v / H + (double)eps);
        if (mean) mean[r] = (float)m;
        if (rstd) rstd[r] = (float)rs;
        for (int c = 0; c < H; ++c) {
            const int p = ln_perm(c, H, interleave_g);
            double o = ((double)x[r * H + c] - m) * rs * (double)gamma[p] + (double)beta[p];
            if (res) o += (double)res[r * H + p];
            y[r * H + p] = (float)o;
            if (yb) yb[r * H + p] = bf16_rne((float)o);
        }
    }
    return REF_OK;
}

int cruse_ln_bwd_ref(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, long long rows, int H,
                     int interleave_g, float* dx, float* dgamma, float* dbeta, void* stream) {
    (void)stream;
    if (rows <= 0 || H <= 0 || interleave_g <= 0 || H % interleave_g) return REF_E_SHAPE;
    double* dg = (double*)calloc((size_t)2 * H, sizeof(double));
    double* db = dg + H;
    for (long long r = 0; r < rows; ++r) {
        const double m = (double)mean[r], rs = (double)rstd[r];
        double s1 = 0.0, s2 = 0.0;
        for (int c = 0; c < H; ++c) {
            const int p = ln_perm(c, H, interleave_g);
            const double xh = ((double)x[r * H + c] - m) * rs, d = (double)dy[r * H + p], gd = d * (double)gamma[p];
            dg[p] += d * xh;
            db[p] += d;
            s1 += gd;
            s2 += gd * xh;
        }
        s1 /= H;
        s2 /= H;
        for (int c = 0; c < H; ++c) {
            const int p = ln_perm(c, H, interleave_g);
            const double xh = ((double)x[r * H + c] - m) * rs, gd = (double)dy[r * H + p] * (double)gamma[p];
            dx[r * H + c] = (float)(rs * (gd - s1 - xh * s2));
        }
    }
    for (int c = 0; c < H; ++c) {
        if (dgamma) dgamma[c] = (float)((double)dgamma[c] + dg[c]);
        if (dbeta) dbeta[c] = (float)((double)dbeta[c] + db[c]);
    }
    free(dg);
    return REF_OK;
}

/* ---- GEMM (GRU gate projections nn.GRU at cruse_net.py:23-31; their dX / dW) -------------------------------------------------- */

/* C[M,N] (=|+=) op(A) op(B) (+ bias[n]); CRUSE_PREC_BF16: the operands are rounded to bf16 (RNE) first, as the MFMA kernels do */
int cruse_gemm_ref(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                   const float* bias, int accumulate, int splitk, int b_shift_T, int prec, void* stream) {
    (void)stream; (void)splitk;
    if (M <= 0 || N <= 0 || K <= 0) return REF_E_SHAPE;
    if (b_shift_T > 0 && transB) return REF_E_SHAPE;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double s = bias ? (double)bias[n] : 0.0;
            for (int k = 0; k < K; ++k) {
                float a = transA ? A[(long long)k * lda + m] : A[(long long)m * lda + k];
                float b;
                if (transB) b = B[(long long)n * ldb + k];
                else if (b_shift_T > 0) b = (k % b_shift_T == 0) ? 0.f : B[(long long)(k - 1) * ldb + n];
                else b = B[(long long)k * ldb + n];
                if (prec == PREC_BF16) { a = bf16_round(a); b = bf16_round(b); }
                s += (double)a * (double)b;
            }
            float* c = C + (long long)m * ldc + n;
            *c = accumulate ? (float)((double)*c + s) : (float)s;
        }
    return REF_OK;
}

/* ---- grouped-GRU recurrence (nn.GRU forward / backward at cruse_net.py:44,50; gate order r, z, n; h0 = 0) -------------------------- */

static void put_coef(void* coef, long long i, double v, int prec) {
    if (prec == PREC_BF16) ((uint16_t*)coef)[i] = bf16_rne((float)v);
    else ((float*)coef)[i] = (float)v;
}
static double get_coef(const void* coef, long long i, int prec) {
    if (prec == PREC_BF16) {
        const uint32_t u = (uint32_t)((const uint16_t*)coef)[i] << 16;
        float f;
        memcpy(&f, &u, 4);
        return (double)f;
    }
    return (double)((const float*)coef)[i];
}

/* gi [B,T,G,3*Hg] = x W_ih^T + b_ih; h [B,T,G*Hg]; coef [B,T,G,3*Hg], an / z [B,T,G*Hg] (all three or none):
 *   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r gh_n), h = (1 - z) n + z h_prev, gh = W_hh h_prev + b_hh
 *   a_n = (1 - z)(1 - n^2), c_r = a_n gh_n r (1 - r), c_z = (h_prev - n) z (1 - z), c_n = a_n r */
int cruse_gru_seq_fwd_ref(const float* gi, const float* const* w_hh, const float* const* b_hh, float* h, void* coef, float* an,
                          float* z, int B, int T, int G, int Hg, int prec, void* ws, void* stream) {
    (void)ws; (void)stream;
    if (B <= 0 || T <= 0 || G <= 0 || Hg <= 0) return REF_E_SHAPE;
    if ((coef != NULL) != (an != NULL) || (coef != NULL) != (z != NULL)) return REF_E_SHAPE;
    const int H = G * Hg;
    double* hp = (double*)calloc((size_t)Hg * 4, sizeof(double));
    double* gh = hp + Hg; /* 3*Hg */
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g) {
            for (int j = 0; j < Hg; ++j) hp[j] = 0.0;
            const float* W = w_hh[g];
            const float* bh = b_hh ? b_hh[g] : NULL;
            for (int t = 0; t < T; ++t) {
                for (int q = 0; q < 3 * Hg; ++q) {
                    double s = bh ? (double)bh[q] : 0.0;
                    for (int k = 0; k < Hg; ++k) s += (double)W[(long long)q * Hg + k] * hp[k];
                    gh[q] = s;
                }
                const long long gb = (((long long)b * T + t) * G + g) * 3 * Hg;
                const long long hb = ((long long)b * T + t) * H + (long long)g * Hg;
                for (int j = 0; j < Hg; ++j) {
                    const double r = 1.0 / (1.0 + exp(-((double)gi[gb + j] + gh[j])));
                    const double u = 1.0 / (1.0 + exp(-((double)gi[gb + Hg + j] + gh[Hg + j])));
                    const double n = tanh((double)gi[gb + 2 * Hg + j] + r * gh[2 * Hg + j]);
                    const double hn = (1.0 - u) * n + u * hp[j];
                    if (coef) {
                        const double a = (1.0 - u) * (1.0 - n * n);
                        put_coef(coef, gb + j, a * gh[2 * Hg + j] * r * (1.0 - r), prec);
                        put_coef(coef, gb + Hg + j, (hp[j] - n) * u * (1.0 - u), prec);
                        put_coef(coef, gb + 2 * Hg + j, a * r, prec);
                        an[hb + j] = (float)a;
                        z[hb + j] = (float)u;
                    }
                    h[hb + j] = (float)hn;
                    hp[j] = hn; /* (h_prev[j] is read by element j only once gh is formed) */
                }
            }
        }
    free(hp);
    return REF_OK;
}

/* dh_s = dout_s + z_{s+1} dh_{s+1} + (dh_{s+1} coef_{s+1}) W_hh: the total gradient reaching h_s */
int cruse_gru_seq_bwd_ref(const float* dout, const float* const* w_hh, const void* coef, const float* z, float* dh, int B, int T,
                          int G, int Hg, int prec, void* ws, void* stream) {
    (void)ws; (void)stream;
    if (B <= 0 || T <= 0 || G <= 0 || Hg <= 0) return REF_E_SHAPE;
    const int H = G * Hg;
    double* nxt = (double*)calloc((size_t)Hg * 5, sizeof(double));
    double* dgh = nxt + Hg; /* 3*Hg */
    double* cur = dgh + 3 * Hg;
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g) {
            const float* W = w_hh[g];
            for (int t = T - 1; t >= 0; --t) {
                const long long hb = ((long long)b * T + t) * H + (long long)g * Hg;
                for (int j = 0; j < Hg; ++j) cur[j] = (double)dout[hb + j];
                if (t + 1 < T) {
                    const long long gb1 = (((long long)b * T + t + 1) * G + g) * 3 * Hg;
                    const long long hb1 = ((long long)b * T + t + 1) * H + (long long)g * Hg;
                    for (int q = 0; q < 3 * Hg; ++q) dgh[q] = nxt[q % Hg] * get_coef(coef, gb1 + q, prec);
                    for (int k = 0; k < Hg; ++k) {
                        double s = (double)z[hb1 + k] * nxt[k];
                        for (int q = 0; q < 3 * Hg; ++q) s += dgh[q] * (double)W[(long long)q * Hg + k];
                        cur[k] += s;
                    }
                }
                for (int j = 0; j < Hg; ++j) { dh[hb + j] = (float)cur[j]; nxt[j] = cur[j]; }
            }
        }
    free(nxt);
    return REF_OK;
}

/* dgi = dh (c_r, c_z, a_n) (gradient wrt gi), dgh = dh (c_r, c_z, c_n) (gradient wrt W_hh h + b_hh), both [rows,G,3*Hg] */
int cruse_gru_gate_grads_ref(const float* dh, const void* coef, const float* an, float* dgi, float* dgh, long long rows, int G,
                             int Hg, int prec, void* stream) {
    (void)stream;
    if (rows <= 0 || G <= 0 || Hg <= 0) return REF_E_SHAPE;
    for (long long r = 0; r < rows; ++r)
        for (int g = 0; g < G; ++g)
            for (int j = 0; j < Hg; ++j) {
                const long long hb = (r * G + g) * Hg + j, gb = (r * G + g) * 3 * Hg;
                const double d = (double)dh[hb];
                const double cr = get_coef(coef, gb + j, prec), cz = get_coef(coef, gb + Hg + j, prec),
                             cn = get_coef(coef, gb + 2 * Hg + j, prec);
                if (dgi) { dgi[gb + j] = (float)(d * cr); dgi[gb + Hg + j] = (float)(d * cz); dgi[gb + 2 * Hg + j] = (float)(d * (double)an[hb]); }
                if (dgh) { dgh[gb + j] = (float)(d * cr); dgh[gb + Hg + j] = (float)(d * cz); dgh[gb + 2 * Hg + j] = (float)(d * cn); }
            }
    return REF_OK;
}

/* ---- mask application + weighted spectral loss -------------------------------------------------------------------------------- */

/* PreProcess.masking "mag_mapping" (utils/utils.py:418-420) + WO-MALE (loss_func/loss.py:121-148, gamma = 1) and its gradient */
int cruse_mask_loss_fwd_ref(const float* mask, const float* nre, const float* nim, const float* cmag, long long rows, int Fn, int Fs,
                            float alpha, float beta, double* loss_sum, float* dmask, float* dlogit, float* est_re, float* est_im,
                            void* stream) {
    (void)stream;
    if (rows <= 0 || Fn <= 0 || Fs < Fn) return REF_E_SHAPE;
    const double n = (double)rows * (double)Fs, ln10 = log(10.0);
    double acc = 0.0;
    for (long long r = 0; r < rows; ++r)
        for (int f = 0; f < Fs; ++f) {
            const long long i = r * Fs + f;
            const double re = (double)nre[i], im = (double)nim[i];
            const double m = f < Fn ? (double)mask[r * Fn + f] : 0.0;
            const double mag_unp = sqrt(re * re + im * im), mag_est = fabs(m) * mag_unp, mag_ref = (double)cmag[i];
            const double w = exp((double)alpha / ((double)beta + mag_ref / mag_unp));
            const double d = log10(mag_est + 1.0) - log10(mag_ref + 1.0);
            acc += w * fabs(d);
            if (est_re) est_re[i] = (float)(m * re);
            if (est_im) est_im[i] = (float)(m * im);
            if (f < Fn && (dmask || dlogit)) {
                const double sgn = d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0);
                const double dm = w * sgn / ln10 / (mag_est + 1.0) * mag_unp / n;
                if (dmask) dmask[r * Fn + f] = (float)dm;
                if (dlogit) dlogit[r * Fn + f] = (float)(dm * m * (1.0 - m));
            }
        }
    loss_sum[0] = acc;
    return REF_OK;
}

int cruse_mask_apply_ref(const float* mask, const float* nre, const float* nim, long long rows, int Fn, int Fs, float* est_re,
                         float* est_im, void* stream) {
    (void)stream;
    if (rows <= 0 || Fn <= 0 || Fs < Fn) return REF_E_SHAPE;
    for (long long r = 0; r < rows; ++r)
        for (int f = 0; f < Fs; ++f) {
            const float m = f < Fn ? mask[r * Fn + f] : 0.f;
            est_re[r * Fs + f] = m * nre[r * Fs + f];
            est_im[r * Fs + f] = m * nim[r * Fs + f];
        }
    return REF_OK;
}

/* backward of nn.Sigmoid (cruse_net.py:164) */
int cruse_sigmoid_bwd_ref(const float* dmask, const float* mask, float* dlogit, long long n, void* stream) {
    (void)stream;
    if (n <= 0) return REF_E_SHAPE;
    for (long long i = 0; i < n; ++i) dlogit[i] = dmask[i] * mask[i] * (1.f - mask[i]);
    return REF_OK;
}

/* ---- DeepFilter head (model/deep_filter.py:15-41, BASELINE config 4): box sum of X * H, imaginary part xr hi + xi hr */
int cruse_deepfilter_fwd_ref(const float* xr, const float* xi, const float* hr, const float* hi, int B, int F, int T, int f_dim,
                             int t_dim, float* out_r, float* out_i, void* stream) {
    (void)stream;
    if (B <= 0 || F <= 0 || T <= 0 || f_dim < 0 || t_dim < 0) return REF_E_SHAPE;
    for (int b = 0; b < B; ++b)
        for (int f = 0; f < F; ++f)
            for (int t = 0; t < T; ++t) {
                double sr = 0.0, si = 0.0;
                for (int df = -f_dim; df <= f_dim; ++df)
                    for (int dt = -t_dim; dt <= t_dim; ++dt) {
                        const int ff = f + df, tt = t + dt;
                        if (ff < 0 || ff >= F || tt < 0 || tt >= T) continue;
                        const long long o = ((long long)b * F + ff) * T + tt;
                        sr += (double)xr[o] * (double)hr[o] - (double)xi[o] * (double)hi[o];
                        si += (double)xr[o] * (double)hi[o] + (double)xi[o] * (double)hr[o];
                    }
                out_r[((long long)b * F + f) * T + t] = (float)sr;
                out_i[((long long)b * F + f) * T + t] = (float)si;
            }
    return REF_OK;
}

/* ---- optimizer (torch.optim.Adam at tools/train_stand.py:68-71; no amsgrad, L2 weight decay) -------------------------------------- */
int cruse_adam_step_ref(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float grad_scale, void* stream) {
    (void)stream;
    if (n <= 0 || step < 1) return REF_E_SHAPE;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = sqrt(1.0 - pow((double)beta2, (double)step));
    for (long long i = 0; i < n; ++i) {
        double gi = (double)g[i] * (double)grad_scale;
        if (weight_decay != 0.f) gi += (double)weight_decay * (double)p[i];
        const double mi = (double)beta1 * (double)m[i] + (1.0 - (double)beta1) * gi;
        const double vi = (double)beta2 * (double)v[i] + (1.0 - (double)beta2) * gi * gi;
        m[i] = (float)mi;
        v[i] = (float)vi;
        p[i] = (float)((double)p[i] - (double)lr / bc1 * (mi / (sqrt(vi) / bc2 + (double)eps)));
    }
    return REF_OK;
}

/* ---- further twins (round 5): adjoints, time-domain losses, small layout / bookkeeping entry points --------------------------------- */

/* adjoint of cruse_istft_fwd_ref (the backward of torch.istft): dwave [B,L] -> dre, dim [B,T,n_fft/2+1]; the imaginary parts of bins 0 and
 * n_fft/2 do not reach the output of an inverse real FFT, so their gradients are 0 */
int cruse_istft_bwd_ref(const float* dwave, int B, int T, int n_fft, int hop, int L, float* dre, float* dim, void* stream) {
    (void)stream;
    if (B <= 0 || T <= 0 || n_fft <= 0 || (n_fft & 1) || hop <= 0 || L <= 0) return REF_E_SHAPE;
    const int Fb = n_fft / 2 + 1, half = n_fft / 2;
    const long long full = (long long)(T - 1) * hop + n_fft;
    double* win = (double*)malloc(sizeof(double) * n_fft);
    double* env = (double*)calloc((size_t)full, sizeof(double));
    double* dfr = (double*)malloc(sizeof(double) * n_fft);
    for (int n = 0; n < n_fft; ++n) win[n] = 0.5 - 0.5 * cos(2.0 * M_PI * n / n_fft);
    for (int t = 0; t < T; ++t)
        for (int n = 0; n < n_fft; ++n) env[(long long)t * hop + n] += win[n] * win[n];
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) {
            for (int n = 0; n < n_fft; ++n) {
                const long long j = (long long)t * hop + n, i = j - half;
                dfr[n] = (i >= 0 && i < L && env[j] > 1e-11) ? (double)dwave[(long long)b * L + i] / env[j] * win[n] / n_fft : 0.0;
            }
            for (int k = 0; k < Fb; ++k) {
                const double c = (k == 0 || k == half) ? 1.0 : 2.0;
                double sr = 0.0, si = 0.0;
                for (int n = 0; n < n_fft; ++n) {
                    const double ph = 2.0 * M_PI * (double)(((long long)k * n) % n_fft) / n_fft;
                    sr += dfr[n] * cos(ph);
                    si -= dfr[n] * sin(ph);
                }
                const long long o = ((long long)b * T + t) * Fb + k;
                dre[o] = (float)(c * sr);
                dim[o] = (k == 0 || k == half) ? 0.f : (float)(c * si);
            }
        }
    free(win);
    free(env);
    free(dfr);
    return REF_OK;
}

/* backward of cruse_mask_apply: dout[rows,Fn] = (dre nre + dim nim) [* mask (1 - mask) through the sigmoid] */
int cruse_mask_apply_bwd_ref(const float* dre, const float* dim, const float* nre, const float* nim, const float* mask, long long rows,
                             int Fn, int Fs, int through_sigmoid, float* dout, void* stream) {
    (void)stream;
    if (rows <= 0 || Fn <= 0 || Fs < Fn) return REF_E_SHAPE;
    for (long long r = 0; r < rows; ++r)
        for (int f = 0; f < Fn; ++f) {
            const long long j = r * Fs + f;
            double d = (double)dre[j] * (double)nre[j] + (double)dim[j] * (double)nim[j];
            if (through_sigmoid) d *= (double)mask[r * Fn + f] * (1.0 - (double)mask[r * Fn + f]);
            dout[r * Fn + f] = (float)d;
        }
    return REF_OK;
}

/* si_snr_loss (train_base/loss.py:7-25): loss = -mean_b 20 log10(eps + |t| / (|x_zm - t| + eps)), t = <x_zm, s_zm> s_zm / (|s_zm|^2 + eps);
 * mom [B,5] = (sum x, sum s, sum x^2, sum s^2, sum x s); coef [B,4] = (A, C, mean_x, mean_s) with d loss / d x = A (x - mean_x) + C (s - mean_s) */
int cruse_sisnr_fwd_ref(const float* x, const float* s, int B, int L, float eps_, double* mom, double* loss, float* coef, void* stream) {
    (void)stream;
    if (B <= 0 || L <= 0) return REF_E_SHAPE;
    const double eps = (double)eps_, n = (double)L;
    loss[0] = 0.0;
    for (int b = 0; b < B; ++b) {
        double m[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < L; ++i) {
            const double a = (double)x[(long long)b * L + i], c = (double)s[(long long)b * L + i];
            m[0] += a; m[1] += c; m[2] += a * a; m[3] += c * c; m[4] += a * c;
        }
        for (int k = 0; k < 5; ++k) mom[b * 5 + k] = m[k];
        const double mx = m[0] / n, ms = m[1] / n;
        const double P = m[4] - n * mx * ms;
        double S2 = m[3] - n * ms * ms; if (S2 < 0) S2 = 0;
        double X2 = m[2] - n * mx * mx; if (X2 < 0) X2 = 0;
        const double S = S2 + eps, alpha = P / S, nt = fabs(alpha) * sqrt(S2);
        double e2 = X2 - 2.0 * alpha * P + alpha * alpha * S2; if (e2 < 0) e2 = 0;
        const double ne = sqrt(e2), r = nt / (ne + eps);
        loss[0] += -20.0 * log10(eps + r) / (double)B;
        const double dr = -20.0 / (log(10.0) * (eps + r)) / (double)B;
        const double sgn = alpha >= 0 ? 1.0 : -1.0, es = P - alpha * S2;
        const double k_nt = dr / (ne + eps) * sgn * sqrt(S2) / S;
        const double k_ne = ne > 0 ? -dr * nt / ((ne + eps) * (ne + eps)) / ne : 0.0;
        coef[b * 4 + 0] = (float)k_ne;
        coef[b * 4 + 1] = (float)(k_nt - k_ne * (alpha + es / S));
        coef[b * 4 + 2] = (float)mx;
        coef[b * 4 + 3] = (float)ms;
    }
    return REF_OK;
}

int cruse_sisnr_bwd_ref(const float* x, const float* s, const float* coef, int B, int L, float grad_scale, float* dx, void* stream) {
    (void)stream;
    if (B <= 0 || L <= 0) return REF_E_SHAPE;
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < L; ++i) {
            const long long o = (long long)b * L + i;
            dx[o] = (float)((double)grad_scale * ((double)coef[b * 4] * ((double)x[o] - (double)coef[b * 4 + 2]) +
                                                 (double)coef[b * 4 + 1] * ((double)s[o] - (double)coef[b * 4 + 3])));
        }
    return REF_OK;
}

/* torch.nn.L1Loss / MSELoss on waveforms (train_base/loss.py:3-4): the SUM and, optionally, grad_scale * d sum / d est */
int cruse_wave_l1_mse_ref(const float* est, const float* ref, long long n, int mse, float grad_scale, double* loss_sum, float* dest,
                          void* stream) {
    (void)stream;
    if (n <= 0) return REF_E_SHAPE;
    double acc = 0.0;
    for (long long i = 0; i < n; ++i) {
        const double d = (double)est[i] - (double)ref[i];
        acc += mse ? d * d : fabs(d);
        if (dest) dest[i] = (float)((double)grad_scale * (mse ? 2.0 * d : (d > 0 ? 1.0 : (d < 0 ? -1.0 : 0.0))));
    }
    loss_sum[0] = acc;
    return REF_OK;
}

/* backward of cruse_deepfilter_fwd: (g_r, g_i) = box(dout) (the box sum is its own adjoint), then the four operand gradients of X * H */
int cruse_deepfilter_bwd_ref(const float* dout_r, const float* dout_i, const float* xr, const float* xi, const float* hr, const float* hi,
                             int B, int F, int T, int f_dim, int t_dim, float* dxr, float* dxi, float* dhr, float* dhi, void* stream) {
    (void)stream;
    if (B <= 0 || F <= 0 || T <= 0 || f_dim < 0 || t_dim < 0) return REF_E_SHAPE;
    for (int b = 0; b < B; ++b)
        for (int f = 0; f < F; ++f)
            for (int t = 0; t < T; ++t) {
                double gr = 0.0, gi = 0.0;
                for (int df = -f_dim; df <= f_dim; ++df)
                    for (int dt = -t_dim; dt <= t_dim; ++dt) {
                        const int ff = f + df, tt = t + dt;
                        if (ff < 0 || ff >= F || tt < 0 || tt >= T) continue;
                        gr += (double)dout_r[((long long)b * F + ff) * T + tt];
                        gi += (double)dout_i[((long long)b * F + ff) * T + tt];
                    }
                const long long o = ((long long)b * F + f) * T + t;
                dxr[o] = (float)(gr * (double)hr[o] + gi * (double)hi[o]);
                dxi[o] = (float)(-gr * (double)hi[o] + gi * (double)hr[o]);
                dhr[o] = (float)(gr * (double)xr[o] + gi * (double)xi[o]);
                dhi[o] = (float)(-gr * (double)xi[o] + gi * (double)xr[o]);
            }
    return REF_OK;
}

/* eval-mode BatchNorm2d: mean = running_mean, rstd = 1 / sqrt(running_var + eps) */
int cruse_bn_eval_stats_ref(const float* running_mean, const float* running_var, int C, float eps, float* mean, float* rstd, void* stream) {
    (void)stream;
    if (C <= 0) return REF_E_SHAPE;
    for (int c = 0; c < C; ++c) { mean[c] = running_mean[c]; rstd[c] = (float)(1.0 / sqrt((double)running_var[c] + (double)eps)); }
    return REF_OK;
}

/* cruse_bn_finalize + cruse_bn_act_fwd as one call; sums is [sum_replicas][2*C], the statistic the sum over the replicas; out_bf16: a bf16 copy */
int cruse_bn_finalize_act_fwd_ref(const float* y, const double* sums, int sum_replicas, long long count, float eps, float momentum,
                                  const float* gamma, const float* beta, const float* skip, float* out, void* out_bf16, float* mean,
                                  float* rstd, float* running_mean, float* running_var, long long rows, int C, int F, int relu, void* stream) {
    if (sum_replicas <= 0 || C <= 0) return REF_E_SHAPE;
    double* tot = (double*)calloc((size_t)2 * C, sizeof(double));
    for (int q = 0; q < sum_replicas; ++q)
        for (int c = 0; c < 2 * C; ++c) tot[c] += sums[(size_t)q * 2 * C + c];
    int rc = cruse_bn_finalize_ref(tot, count, C, eps, momentum, mean, rstd, running_mean, running_var, stream);
    free(tot);
    if (rc) return rc;
    rc = cruse_bn_act_fwd_ref(y, mean, rstd, gamma, beta, skip, out, rows, C, F, relu, stream);
    if (rc) return rc;
    if (out_bf16)
        for (long long i = 0; i < rows * C * F; ++i) ((uint16_t*)out_bf16)[i] = bf16_rne(out[i]);
    return REF_OK;
}

/* out[j] += sum_rows g[row*ld + j], j < ncol (GRU bias gradients) */
int cruse_col_sum_ref(const float* g, long long rows, int ncol, int ld, float* out, void* stream) {
    (void)stream;
    if (rows <= 0 || ncol <= 0 || ld < ncol) return REF_E_SHAPE;
    for (int j = 0; j < ncol; ++j) {
        double s = 0.0;
        for (long long r = 0; r < rows; ++r) s += (double)g[r * ld + j];
        out[j] = (float)((double)out[j] + s);
    }
    return REF_OK;
}

/* out = a x + b y (y may be NULL) */
int cruse_axpby_ref(float* out, const float* x, const float* y, float a, float b, long long n, void* stream) {
    (void)stream;
    if (n <= 0) return REF_E_SHAPE;
    for (long long i = 0; i < n; ++i) out[i] = a * x[i] + (y ? b * y[i] : 0.f);
    return REF_OK;
}

/* y = bf16(x), round to nearest even */
int cruse_cast_bf16_ref(const float* x, void* y, long long n, void* stream) {
    (void)stream;
    if (n <= 0) return REF_E_SHAPE;
    for (long long i = 0; i < n; ++i) ((uint16_t*)y)[i] = bf16_rne(x[i]);
    return REF_OK;
}
