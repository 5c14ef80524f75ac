/* libcruse_hip.so -- C ABI of the MI355X (gfx950) CRUSE hot path.
 *
 * The reference (Okrio/CRUSE) is pure Python and has no FFI; its extension point
 * is dotted-path module loading (train_base/utils.py:68-100).  Every entry point
 * below therefore replaces a *PyTorch library op at one of the reference's call
 * sites* (SURVEY.md section 2b); the call site is cited on each declaration.
 * Host code (cruse_amd/) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - extern "C"; every function returns int: 0 = ok, <0 = CRUSE_E_*;
 *     cruse_last_error() returns a thread-local message.
 *   - all buffers are caller-allocated DEVICE pointers (f32 unless noted); the
 *     library never frees or retains them beyond the call.
 *   - last argument is the hipStream_t (passed as void*); calls are asynchronous
 *     and re-entrant per stream; no global mutable state.
 *   - activation layout is "frame-major": [B, T, C, F] contiguous, i.e. one row of
 *     C*F floats per spectrogram frame.  For C == 1 this is the reference's
 *     [B,1,T,F]; GGRU's [B,T,C*F] view (model/cruse_net.py:39-40) is this layout
 *     with no copy.
 *   - "+=" outputs accumulate into the caller's buffer (gradient buffers).
 */
#ifndef CRUSE_HIP_H
#define CRUSE_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRUSE_ABI_VERSION 13

enum {
    CRUSE_OK = 0,
    CRUSE_E_SHAPE = -1,
    CRUSE_E_ALIGN = -2,
    CRUSE_E_DTYPE = -3,
    CRUSE_E_HIP = -4,
    CRUSE_E_TIMEOUT = -5
};

/* precision of the MFMA contractions (GEMMs and GRU recurrence) */
enum {
    CRUSE_PREC_F32 = 0,    /* v_mfma_f32_16x16x4_f32: exact f32                         */
    CRUSE_PREC_BF16X3 = 1, /* split bf16 hi+lo, 3 bf16 MFMAs: ~2^-17 relative           */
    CRUSE_PREC_BF16 = 2,   /* operands rounded to bf16, f32 accumulate                  */
    CRUSE_PREC_F16 = 3     /* v_mfma_f32_16x16x32_f16: operands rounded to f16, f32 accumulate (cruse_gemm; BASELINE config 5) */
};

/* storage type of the activation tensors of the general [B,C,H,W] blocks (cruse_conv2d_nchw, cruse_bn_nchw_*, ...):
 * BASELINE config 5 ("MTFAA ... fp16") keeps them in f16 -- the blocks are HBM-bound, so the bytes are what fp16 buys --
 * with f32 parameters, parameter gradients, accumulation and statistics; its pointwise convolutions run on
 * v_mfma_f32_16x16x32_f16. */
enum {
    CRUSE_DT_F32 = 0,
    CRUSE_DT_F16 = 1,
    CRUSE_DT_BF16 = 2     /* backward-only tensors of the bf16 mode (x_dtype / a_dtype / bt_dtype / dy_dtype below) */
};

int cruse_abi_version(void);
const char* cruse_last_error(void);
/* Library options -- kernel-selection A/B switches and profiling aids, set EXPLICITLY by the host (the library reads no
 * environment variables): "gru_bwd_rs" / "gru_fwd_lean" 0 = the generic recurrence kernels in bf16 mode; "gru_wlo" 0 / 1 =
 * W_hh low plane of the lean forward recurrence off / on for every Hg (unset: on for Hg <= 320); "gru_bg" 16; "gru_dbg",
 * "wg_dbg" phase-skip profiling; "cm_grid", "cm_nw", "gb_deep_min", "gb_deep", "lnb_grid", "wg_tfw", "wg_grid" launch
 * geometry overrides; "pw_valu" 1 = VALU pointwise convolutions in f16 storage.  unset != 0 restores the default. */
int cruse_set_option(const char* name, int value, int unset);
int cruse_get_option(const char* name, int* value, int* is_set);

/* ---- acoustic front end ---------------------------------------------------- */

/* torch.stft(y, n_fft, hop, win=n_fft, window=hann_window(n_fft), center=True,
 * return_complex=True) at train_base/acoustics/feature.py:22-30, fused with the
 * [B,T,F] transposition and magnitude of utils/utils.py:397-400.
 * wave [B,L] -> re, im [B,T,n_fft/2+1] (either may be NULL),
 *               mag [B,T,mag_bins] = sqrt(re^2+im^2+mag_eps) for the first mag_bins bins (may be NULL).
 * T = 1 + L/hop.  Reflect padding of n_fft/2.  n_fft == 320 runs the radix-5 x 64
 * wavefront-shuffle FFT; any other even n_fft <= 2048 a direct DFT. */
int cruse_stft_fwd(const float* wave, int B, int L, int n_fft, int hop,
                   float* re, float* im, float* mag, int mag_bins, float mag_eps, void* stream);

/* torch.istft(X, n_fft, hop, win=n_fft, window=hann_window(n_fft), center=True, length=L)
 * at feature.py:53-61 / utils/utils.py:448-454.  re, im [B,T,n_fft/2+1] -> wave [B,L]. */
int cruse_istft_fwd(const float* re, const float* im, int B, int T, int n_fft, int hop, int L,
                    float* wave, void* stream);
/* adjoint of cruse_istft_fwd: dwave [B,L] -> dre, dim [B,T,n_fft/2+1] */
int cruse_istft_bwd(const float* dwave, int B, int T, int n_fft, int hop, int L,
                    float* dre, float* dim, void* stream);

/* ---- convolutions (nn.Conv2d / nn.ConvTranspose2d at model/cruse_net.py:138-143,149-164) */

/* gather form:
 *   y[b,t,co,fo] (+)= act(bias[co] + sum_{ci,kt,kf} W(co,ci,kt,kf) * x[b, t-(KT-1)+kt, ci, fo*S - pad + kf])
 * zero outside the clip.  kf in 0..2.  w_layout 0: W = w[co][ci][kt][kf] (Conv2d weight, or a
 * ConvTranspose2d weight read as [out=Cin_t][in=Cout_t] for its backward-data).
 * w_layout 1 (KT == 1, S == 1): W(co,ci,0,kf) = w[ci][co][0][2-kf] (backward-data of a stride-1 conv).
 * act 0 none, 1 sigmoid.  accum != 0: y += result (act must be 0). bias may be NULL.
 * prec: CRUSE_PREC_* selects the MFMA implicit-GEMM kernel (Cin % 8 == 0, 8 <= Cout <= 64);
 * prec < 0, or an ineligible shape (Cin == 1, Cout == 1), runs the exact-f32 VALU kernel. */
int cruse_conv_gather(const float* x, const float* w, const float* bias, float* y,
                      int B, int T, int Cin, int Fin, int Cout, int Fout,
                      int KT, int S, int pad, int w_layout, int act, int accum, int prec, int x_dtype, int y_dtype, void* stream);

/* scatter form, frequency stride 2:
 *   y[b,t,co,fo] (+)= act(bias[co] + sum_{cs,kt,kf : (fo+pad-kf) even} w[cs][co][kt][kf] *
 *                                     g[b, t+(KT-1)-kt, cs, (fo+pad-kf)/2])
 * ConvTranspose2d((1,3), stride (1,2)) forward with the [..., :-1] crop (cruse_net.py:161-164):
 * KT=1, pad=0, Fout=2*Fg.  Backward-data of the (2,3)/(1,2) encoder conv: KT=2, pad=1. */
int cruse_conv_scatter2(const float* g, const float* w, const float* bias, float* y,
                        int B, int T, int Cs, int Fg, int Cout, int Fout,
                        int KT, int pad, int act, int accum, int prec, int x_dtype, int y_dtype, void* stream);

/* Both forms with the batch statistics of the BatchNorm2d that follows every encoder / decoder conv
 * (cruse_net.py:139,150) accumulated by the conv's own epilogue: y = conv (no activation, no accumulate), and what
 * cruse_bn_stats(y) would add -- sum_{b,t,f} y and sum y^2 per channel -- without the extra pass over y.
 * sums is [CRUSE_BN_STAT_REPLICAS][2*Cout] doubles: a workgroup adds its partial sums to replica (block id mod
 * replicas), so the f64 atomics of ~1000 workgroups do not queue on 2*Cout addresses (that queue cost 15-30 us per
 * layer); the statistic is the sum over the replicas -- cruse_bn_finalize_act_fwd(sum_replicas) folds them.  The
 * partial sums are f32-valued, so the f64 additions are exact and the result does not depend on their order.
 * zeroed != 0: the caller has cleared sums (else cleared here, on `stream`). */
#define CRUSE_BN_STAT_REPLICAS 16
int cruse_conv_gather_bnstats(const float* x, const float* w, const float* bias, float* y,
                              int B, int T, int Cin, int Fin, int Cout, int Fout,
                              int KT, int S, int pad, int prec, double* sums, int zeroed, void* stream);
int cruse_conv_scatter2_bnstats(const float* g, const float* w, const float* bias, float* y,
                                int B, int T, int Cs, int Fg, int Cout, int Fout,
                                int KT, int pad, int prec, double* sums, int zeroed, void* stream);

/* FORWARD convolutions that apply the BatchNorm2d(train) + ReLU of the layer BELOW to their input while staging it (ABI 7; the
 * BN -> ReLU -> conv chains of cruse_net.py:149-152,161-163): x_pre / g_pre is the PRE-BatchNorm tensor [B,T,Cin,Fin]; every workgroup
 * derives mean / rstd from that layer's batch sums in_sums [in_replicas][2*Cin] (in_count = rows*Fin elements per channel) and stages
 * e = relu((x - mean)*rstd*gamma + beta) [+ in_add, the decoder's skip tensor] -- cruse_bn_finalize_act_fwd's arithmetic, so the
 * normalised tensor never exists in HBM.  in_mean / in_rstd (nullable pair): block 0 publishes the statistics (the backward pass
 * reads them) and, with in_running_mean / in_running_var, updates the running statistics -- ONE consumer of a layer passes them.
 * in_copy_bf16 (nullable): a bf16 copy of the transformed rows, the operand the weight gradients of the bf16 mode read.
 * out_sums (nullable): the batch sums of y as cruse_conv_*_bnstats.  MFMA kernel only: prec == CRUSE_PREC_BF16X3 (the forward
 * convs of the bf16 mode), Cin a power of two in 8..64, 8 <= Cout <= 64; anything else is refused (run the unfused passes). */
int cruse_conv_gather_bnin(const float* x_pre, const double* in_sums, int in_replicas, long long in_count, float eps, float momentum,
                           const float* in_gamma, const float* in_beta, float* in_mean, float* in_rstd, float* in_running_mean,
                           float* in_running_var, const float* in_add, void* in_copy_bf16,
                           const float* w, const float* bias, float* y, int B, int T, int Cin, int Fin, int Cout, int Fout,
                           int KT, int S, int pad, int prec, double* out_sums, int zeroed, void* stream);
int cruse_conv_scatter2_bnin(const float* g_pre, const double* in_sums, int in_replicas, long long in_count, float eps, float momentum,
                             const float* in_gamma, const float* in_beta, float* in_mean, float* in_rstd, float* in_running_mean,
                             float* in_running_var, const float* in_add, void* in_copy_bf16,
                             const float* w, const float* bias, float* y, int B, int T, int Cs, int Fg, int Cout, int Fout,
                             int KT, int pad, int prec, double* out_sums, int zeroed, void* stream);

/* x_dtype / a_dtype / bt_dtype / dy_dtype (ABI 7; CRUSE_DT_F32 or CRUSE_DT_BF16): BACKWARD-ONLY tensors of the bf16 mode -- the
 * BatchNorm-backward output dy, which its two consumers (the data-gradient conv and the weight gradient) round to bf16 operands
 * anyway -- may be stored as bf16 in the same [B,T,C,F] layout: half the bytes written once and read twice, the same bits into
 * the MFMAs.  bf16 inputs need the MFMA kernels (CRUSE_PREC_BF16 for the weight gradient); the VALU fall-backs refuse them.
 * y_dtype / dout_dtype: the data gradients themselves (de, du: conv output -> BatchNorm-backward input, skip-path leaves) as bf16
 * too -- sums and accumulation stay f32, the stored value is rounded once per pass; a bf16 output goes with a bf16 input and
 * the swapped-role (vector-store) forms of the MFMA conv. */
/* DATA-GRADIENT convolutions whose output y is the gradient wrt the OUTPUT of a BatchNorm2d(+ReLU) (the backward pass of
 * cruse_net.py:139-142,149-152 walks conv -> BN -> ReLU stacks, so every data gradient but the first feeds a BatchNorm
 * backward): the conv's epilogue also accumulates what cruse_bn_act_bwd_reduce(dout = y, bn_y, ...) would -- sum g and
 * sum g*xhat per channel, g = y masked by the ReLU, xhat from the pre-BN tensor bn_y [B,T,Cout,Fout] -- into
 * sums [CRUSE_BN_STAT_REPLICAS][2*Cout] (cruse_bn_act_bwd_apply(sum_replicas) folds them): that pass over (y, bn_y), 132 MB
 * and ~47 us per level at the bench shape, is not needed.  No bias, no activation; accum != 0: y += conv (the skip path's
 * gradient is already there) and the statistics are those of the sum.  Shapes without the MFMA kernel run the conv and
 * then the reduce pass (into replica 0).  The per-workgroup partial sums are f32, added in f64: independent of order. */
int cruse_conv_gather_bnbwd(const float* x, const float* w, float* y, int B, int T, int Cin, int Fin, int Cout, int Fout,
                            int KT, int S, int pad, int w_layout, int accum, int prec,
                            const float* bn_y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                            int relu, double* sums, int zeroed, int x_dtype, int y_dtype, void* stream);
int cruse_conv_scatter2_bnbwd(const float* g, const float* w, float* y, int B, int T, int Cs, int Fg, int Cout, int Fout,
                              int KT, int pad, int accum, int prec,
                              const float* bn_y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                              int relu, double* sums, int zeroed, int x_dtype, int y_dtype, void* stream);

/* (ABI 8) DATA-GRADIENT convolutions that also apply the BatchNorm2d(+ReLU) BACKWARD of their INPUT: dout is the gradient wrt the OUTPUT of
 * the BatchNorm above the convolution being differentiated (cruse_net.py:139-142,149-152: conv -> BN -> ReLU), in_y that BatchNorm's pre-BN
 * tensor [B,T,Cin,Fin], in_sums its backward batch sums [in_replicas][2*Cin] (sum g, sum g*xhat -- from cruse_conv_*_bnbwd or
 * cruse_bn_act_bwd_reduce).  One call = cruse_bn_act_bwd_apply(dout -> dy_bf16; in_dgamma / in_dbeta / in_dbias += ...) followed by
 * cruse_conv_gather / _scatter2 [_bnbwd] (dy_bf16 -> y): in the bf16 data-gradient mode with a bf16 dout the MFMA kernel forms dy while it
 * stages its tiles (the arithmetic of cruse_bn_act_bwd_apply, rounded to bf16 once) and writes dy_bf16 [B,T,Cin,Fin] -- the operand of the weight
 * gradient -- on the way: the separate pass over (dout, in_y), 132 MB and 40-95 us per level in the step, is gone.  Any other shape / dtype runs
 * the two calls.  bn_y / mean / ... / sums: the OUTPUT-side statistics of cruse_conv_*_bnbwd (bn_y == NULL and sums == NULL: none). */
int cruse_conv_gather_bnbwd_in(const void* dout, int dout_dtype, const float* in_y, const float* in_mean, const float* in_rstd,
                               const float* in_gamma, const float* in_beta, const double* in_sums, int in_replicas, int in_relu,
                               int in_training, void* dy_bf16, float* in_dgamma, float* in_dbeta, float* in_dbias,
                               const float* w, void* y, int B, int T, int Cin, int Fin, int Cout, int Fout, int KT, int S, int pad,
                               int w_layout, int accum, int prec,
                               const float* bn_y, const float* mean, const float* rstd, const float* gamma, const float* beta, int relu,
                               double* sums, int zeroed, int y_dtype, void* stream);
int cruse_conv_scatter2_bnbwd_in(const void* dout, int dout_dtype, const float* in_y, const float* in_mean, const float* in_rstd,
                                 const float* in_gamma, const float* in_beta, const double* in_sums, int in_replicas, int in_relu,
                                 int in_training, void* dy_bf16, float* in_dgamma, float* in_dbeta, float* in_dbias,
                                 const float* w, void* y, int B, int T, int Cs, int Fg, int Cout, int Fout, int KT, int pad, int accum,
                                 int prec,
                                 const float* bn_y, const float* mean, const float* rstd, const float* gamma, const float* beta, int relu,
                                 double* sums, int zeroed, int y_dtype, void* stream);

/* profiling aid: with cruse_set_option("cm_dbg", 1) workgroup 0 of an MFMA convolution launch stamps s_memtime at its phase
 * boundaries; this copies the sums of the LAST such launch to out8 (host memory, 8 values: prologue, tile staging incl. the wait
 * for the prefetch, k-loops, epilogues, tiles, N-tiles of wave 0, total; cycles) -- tools/conv_probe.py */
int cruse_conv_mfma_stamps(unsigned long long* out8);

/* weight gradient of either form:
 *   dw[ca][cb][kt][kf] += sum_{b,t,fa} a[b,t,ca,fa] * bt[b, t-(KT-1)+kt, cb, fa*S - pad + kf]
 * ws: scratch of cruse_conv_wgrad_ws_bytes() bytes.  prec: CRUSE_PREC_* selects an MFMA kernel -- CRUSE_PREC_BF16 with rows of >= 8
 * (even) positions and Fb == S * Fa: fragments loaded straight from the two tensors (wgrad_rd.hip); otherwise the patch matrix is
 * materialised in LDS (wgrad_mfma.hip); prec < 0 or an ineligible shape runs the f32 VALU kernel. */
size_t cruse_conv_wgrad_ws_bytes(int Ca, int Cb, int KT);
int cruse_conv_wgrad(const float* a, const float* bt, float* dw,
                     int B, int T, int Ca, int Fa, int Cb, int Fb,
                     int KT, int S, int pad, int prec, int a_dtype, int bt_dtype, void* ws, void* stream);

/* out[c] += sum_{rows,f} g[row,c,f]   (bias gradients; F=1 gives a column sum) */
int cruse_channel_sum(const float* g, long long rows, int C, int F, float* out, void* stream);
/* out[j] += sum_rows g[row*ld + j], j < ncol  (GRU bias gradients: a column slice of dgi / dgh) */
int cruse_col_sum(const float* g, long long rows, int ncol, int ld, float* out, void* stream);

/* ---- BatchNorm2d (+ReLU, + skip add) (cruse_net.py:141-142,149-152,161-163) ---- */

/* sums[0..C) = sum y, sums[C..2C) = sum y^2 over rows x F (f64).  zeroed != 0: the caller hands over accumulators that are
 * already zero (the step's scratch arena, cleared by ONE cruse_zero per step) -- otherwise the callee clears them first. */
int cruse_bn_stats(const float* y, long long rows, int C, int F, double* sums, int zeroed, void* stream);
/* training: mean/rstd from sums (biased var), running stats updated with momentum and
 * unbiased var when running_mean != NULL (torch BatchNorm2d semantics). */
int cruse_bn_finalize(const double* sums, long long count, int C, float eps, float momentum,
                      float* mean, float* rstd, float* running_mean, float* running_var, void* stream);
/* eval: mean = running_mean, rstd = 1/sqrt(running_var+eps) */
int cruse_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps,
                        float* mean, float* rstd, void* stream);
/* out = [relu]((y-mean)*rstd*gamma+beta) [+ skip] */
int cruse_bn_act_fwd(const float* y, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, const float* skip, float* out,
                     long long rows, int C, int F, int relu, void* stream);
/* out_bf16 (nullable): also a bf16 copy of out, [rows, C*F] -- the operand of the gate GEMM that reads the encoder's last
 * level, saving a cast pass.
 * cruse_bn_finalize + cruse_bn_act_fwd as one launch (training): mean / rstd come from the batch sums inside the kernel,
 * are also written out (the backward pass needs them) and the running statistics are updated.  sums is
 * [sum_replicas][2*C] and the statistic the sum over the replicas: 1 after cruse_bn_stats, CRUSE_BN_STAT_REPLICAS after
 * cruse_conv_*_bnstats. */
int cruse_bn_finalize_act_fwd(const float* y, const double* sums, int sum_replicas, long long count, float eps, float momentum,
                              const float* gamma, const float* beta, const float* skip, float* out, void* out_bf16,
                              float* mean, float* rstd, float* running_mean, float* running_var,
                              long long rows, int C, int F, int relu, void* stream);
/* sums[0..C) = sum g, sums[C..2C) = sum g*xhat with g = dout * [bn(y) > 0]; `zeroed` as for cruse_bn_stats */
/* (ABI 9) the same with the element type of the 2-byte operand copy chosen: copy_dtype = CRUSE_DT_BF16 (the function above) or
 * CRUSE_DT_F16 -- the operand of the single-pass f16 gate projection cruse_gemm_f16_nt. */
int cruse_bn_finalize_act_fwd_c(const float* y, const double* sums, int sum_replicas, long long count, float eps, float momentum,
                                const float* gamma, const float* beta, const float* skip, float* out, void* out_copy, int copy_dtype,
                                float* mean, float* rstd, float* running_mean, float* running_var,
                                long long rows, int C, int F, int relu, void* stream);
int cruse_bn_act_bwd_reduce(const float* dout, const float* y, const float* mean, const float* rstd,
                            const float* gamma, const float* beta, long long rows, int C, int F,
                            int relu, double* sums, int zeroed, void* stream);
/* dy = gamma*rstd*(g - [training](sum_g + xhat*sum_gx)/count); dgamma += sum_gx; dbeta += sum_g;
 * dbias (nullable) += per-channel sum of dy -- the gradient of the bias of the conv that feeds this BN -- in closed
 * form from the sums: gamma*rstd*sum_g with running statistics, and exactly 0 with batch statistics (sum xhat = 0:
 * the bias is cancelled by the mean subtraction; autograd leaves rounding noise there)
 * sums is [sum_replicas][2*C] (1 after cruse_bn_act_bwd_reduce, CRUSE_BN_STAT_REPLICAS after cruse_conv_*_bnbwd). */
int cruse_bn_act_bwd_apply(const float* dout, const float* y, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, const double* sums, int sum_replicas,
                           long long rows, int C, int F, int relu, int training, int dout_dtype,
                           void* dy, int dy_dtype, float* dgamma, float* dbeta, float* dbias, void* stream);

/* ---- LayerNorm (+ group interleave, + residual) (cruse_net.py:32-33,43-51,160) -- */

/* y[row, P(c)] = (x[row,c]-mean)*rstd*gamma[P(c)] + beta[P(c)] (+ res[row,P(c)])
 * P(i*Hg + j) = j*g + i with Hg = H/g is the stack(dim=-1)+flatten of cruse_net.py:43-45
 * (interleave_g = g; 1 = identity / the plain cat of :49-50). eps = 1e-5.
 * y_bf16 (nullable): also a bf16 copy of y [rows, H] (the next gate GEMM's operand).
 * seg_len > 0 (interleave_g == 1): ROW SEGMENTS -- logical row r of every row-indexed array is physical row
 * (r / seg_len) * seg_stride + seg_off + r % seg_len: frames [seg_off, seg_off + seg_len) of every clip of [B, T = seg_stride]
 * tensors, i.e. one TIME CHUNK of the batch (run beside the recurrence of the next chunk); rows = B * seg_len. */
int cruse_ln_fwd(const float* x, const float* gamma, const float* beta, const float* res,
                 float* y, void* y_bf16, float* mean, float* rstd, long long rows, int H, int interleave_g,
                 float eps, int seg_len, long long seg_stride, long long seg_off, void* stream);
/* (ABI 9) cruse_ln_fwd with the element type of the operand copy chosen (CRUSE_DT_BF16 or CRUSE_DT_F16; f16 needs interleave_g == 1) */
int cruse_ln_fwd_c(const float* x, const float* gamma, const float* beta, const float* res,
                   float* y, void* y_copy, int copy_dtype, float* mean, float* rstd, long long rows, int H, int interleave_g,
                   float eps, int seg_len, long long seg_stride, long long seg_off, void* stream);
int cruse_ln_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                 const float* gamma, long long rows, int H, int interleave_g,
                 float* dx, float* dgamma, float* dbeta, void* stream);

/* ---- MFMA GEMM (GRU gate projections nn.GRU at cruse_net.py:23-31; their dX / dW) -- */

/* C[M,N] (=|+=) op(A) * op(B) (+ bias[n]);  op(A) is [M,K]: transA==0 -> A[m*lda+k], else A[k*lda+m];
 * op(B) is [K,N]: transB==0 -> B[k*ldb+n], else B[n*ldb+k].
 * accumulate != 0: C += ...; splitk > 1 splits K over blockIdx.z and adds atomically (implies +=).
 * b_shift_T > 0 (transB==0 only): row k of B is read from row k-1 and is zero when k % b_shift_T == 0
 * (the h_{t-1} operand of dW_hh). prec: CRUSE_PREC_*. */
int cruse_gemm(int transA, int transB, int M, int N, int K,
               const float* A, int lda, const float* B, int ldb, float* C, int ldc,
               const float* bias, int accumulate, int splitk, int b_shift_T, int prec, void* stream);

/* bf16-operand form of the same products for CRUSE_PREC_BF16: C[M,N] (=|+=) A[M,K] . B[N,K]^T (+ bias[n]), C f32,
 * K % 64 == 0.  Operand element (m, k) is A[(k/64)*a_kstride + m*lda + k%64]: a_kstride == 64 is plain row-major
 * (lda >= K); the K-TILED time-major layout of cruse_transpose_bf16 / cruse_gru_gate_grads_bf16 has lda == 64 and
 * a_kstride == 64 * (number of lines).  Same for B.  splitk > 1 adds atomically and needs accumulate != 0;
 * splitk < -1 is |splitk| k-slices with slice z computed entirely on XCD z % 8 (all output tiles of a slice share that
 * XCD's L2: the long-K weight gradients then read every operand byte from HBM about once).
 * Operand values are the RNE roundings cruse_gemm forms internally. */
int cruse_gemm_bf16_nt(int M, int N, int K, const void* A, long long lda, long long a_kstride,
                       const void* B, long long ldb, long long b_kstride,
                       float* C, long long ldc, const float* bias, int accumulate, int splitk, void* stream);
/* The split-K weight-gradient form without atomics: C[M,N] += A . B^T with |splitk| k-slices (splitk < -1: pinned to XCDs as
 * above), every slice STORING its partial sums to its own [M][N] slab of `scratch` (cruse_gemm_bf16_slab_bytes(M, N, splitk)
 * bytes, 16-byte aligned) and one kernel adding the slabs to C in slice order.  The f32 atomics of the splitk form from the 8 XCDs
 * meet at the memory side (9.8 M per gate weight gradient: 0.14 ms of the training step), and their order made dW differ from run
 * to run in the last bits; this form is reproducible.  N % 4 == 0. */
size_t cruse_gemm_bf16_slab_bytes(int M, int N, int splitk);
int cruse_gemm_bf16_nt_slabs(int M, int N, int K, const void* A, long long lda, long long a_kstride,
                             const void* B, long long ldb, long long b_kstride,
                             float* C, long long ldc, int splitk, void* scratch, size_t scratch_bytes, void* stream);
/* (ABI 9) Up to three products that share N, K, the operand strides and the A tensor, concatenated along M into ONE launch and ONE output:
 *   C[sum M_i, N] += cat_i( A[a_rows[i] .. a_rows[i] + M_i) . B_i^T )      slab form as above (a product whose M_i is not a multiple of 128 still starts on a
 *   row tile of its own; the output rows stay contiguous).
 * The three weight-gradient products of an nn.GRU layer (cruse_net.py:23-31: dW_ih += (r, z, n_i)^T x, dW_hh += (r, z)^T h_{t-1} and
 * n_h^T h_{t-1}) are such a set: their A rows are slabs of the one time-major gate-gradient tensor, and dW_ih / dW_hh lie back to back in
 * the flat gradient buffer.  Ms, a_rows, Bs: HOST arrays of nprob (1..3) entries. */
int cruse_gemm_bf16_nt_slabs_cat(int nprob, const int* Ms, int N, int K, const void* A, const long long* a_rows, long long lda,
                                 long long a_kstride, const void* const* Bs, long long ldb, long long b_kstride,
                                 float* C, long long ldc, int splitk, void* scratch, size_t scratch_bytes, void* stream);
/* (ABI 9) G products of the same shape in ONE launch, side by side along N -- the GRU groups of one layer (GGRU with rnn_groups > 1,
 * cruse_net.py:14-55):  C[:, q * c_gstep + (0..N)] (+)= A[:, q * a_gstep + (0..K)] . B_q^T + bias_q,  B_q = B + q * b_gstep, bias_q = bias + q * bias_gstep.
 * A row-major (planes A_hi / A_lo nullable as in cruse_gemm_bf16x3_nt), B as in cruse_gemm_bf16_nt (B_lo nullable, same group stride).  Used for
 * the forward gate projections and the input gradients of the grouped configurations: G launches of 2-4 column tiles each become one. */
int cruse_gemm_bf16_nt_groups(int M, int N, int K, int G, const void* A_hi, const void* A_lo, long long lda, long long a_gstep,
                              const void* B_hi, const void* B_lo, long long ldb, long long b_kstride, long long b_gstep,
                              float* C, long long ldc, long long c_gstep, const float* bias, long long bias_gstep, int accumulate,
                              void* stream);
/* cruse_gemm_bf16_nt / cruse_gemm_bf16x3_nt (A_lo, B_lo nullable: plain bf16) on a TIME CHUNK of the batch: logical row m of A
 * (row-major, lda) and of C is physical row (m / seg_len) * seg_stride + seg_off + m % seg_len; M = B * seg_len.  The gate
 * projection gi = x W_ih^T of frames [seg_off, seg_off + seg_len) of every clip then runs beside the recurrence of the
 * frames before them (model/cruse_net.py:44,50 chunked in time; results identical to the unchunked call). */
int cruse_gemm_bf16_nt_seg(int M, int N, int K, const void* A_hi, const void* A_lo, long long lda,
                           const void* B_hi, const void* B_lo, long long ldb, long long b_kstride,
                           float* C, long long ldc, const float* bias, int accumulate,
                           int seg_len, long long seg_stride, long long seg_off, void* stream);
/* y[i] = bf16(x[i]), n % 4 == 0 */
int cruse_cast_bf16(const float* x, void* y, long long n, void* stream);
/* K-tiling without transposition: y[(k/64)*rows*64 + n*64 + k%64] = bf16(x[n*ld + k]), zero for cols <= k < ceil64(cols):
 * a K-contiguous operand (W_ih [3Hg, Hg]) whose K is not a multiple of 64.  y_lo (nullable): the low plane
 * bf16(x - y) of the split-bf16 x3 form, same layout. */
int cruse_ktile_bf16(const float* x, int rows, int cols, long long ld, void* y, void* y_lo, void* stream);
/* (ABI 9) the same K-tiled layout with IEEE-f16 elements: the B operand of cruse_gemm_f16_nt */
int cruse_ktile_f16(const float* x, int rows, int cols, long long ld, void* y, void* stream);
/* (ABI 12) ... with the low plane y_lo = f16(x - f16(x)), same layout: the B_lo operand of cruse_gemm_f16x2_nt */
int cruse_ktile_f16_split(const float* x, int rows, int cols, long long ld, void* y, void* y_lo, void* stream);
/* (ABI 13; first in ABI 9-10) C[M,N] (+)= A[M,K] . B[N,K]^T with A read from its TIME-MAJOR K-tiled image -- element (m, k) at
 * A_T[(m / 64) * a_mb_stride + k * 64 + m % 64], n_mb 64-row blocks present -- the layout of the gate-gradient tensor dgT the weight-gradient GEMMs
 * consume: the input gradient dX = dgi . W_ih of nn.GRU's backward (model/cruse_net.py:23-31,44,50) straight from it, so that the row-major copy dgi
 * (98 MB per layer at the bench shape) is never written.  B as cruse_gemm_bf16_nt; plain bf16, one pass, K % 64 == 0.  Bit-identical to the row-major form. */
int cruse_gemm_bf16_nt_atr(int M, int N, int K, const void* A_T, long long a_mb_stride, int n_mb,
                           const void* B, long long ldb, long long b_kstride,
                           float* C, long long ldc, int accumulate, void* stream);
/* (ABI 13) The forward gate projections with a 2-BYTE RESULT: C[M,N] = (A_hi + A_lo) . (B_hi + B_lo)^T + bias stored as IEEE f16 (out_dtype =
 * CRUSE_DT_F16) or bf16 (CRUSE_DT_BF16) rows [M, ldc] -- gi is the largest tensor of the forward pass (197 MB in f32 at the bench shape), written once
 * here and read once by cruse_gru_seq_fwd_gi16.  operands_f16 = 0: bf16 operand planes in the layouts of cruse_gemm_bf16_nt / cruse_gemm_bf16x3_nt
 * (A_lo, B_lo nullable; A_lo needs B_lo); 1: IEEE-f16 planes (cruse_gemm_f16_nt / cruse_gemm_f16x2_nt; no A_lo).  f32 accumulation, ONE rounding at
 * the store.  Replaces the output side of nn.GRU's input projection (model/cruse_net.py:23-31,44,50). */
int cruse_gemm_nt_out16(int M, int N, int K, const void* A_hi, const void* A_lo, long long lda, long long a_kstride,
                        const void* B_hi, const void* B_lo, long long ldb, long long b_kstride,
                        void* C, long long ldc, const float* bias, int operands_f16, int out_dtype, void* stream);
/* (ABI 9) C[M,N] = A[M,K] . B[N,K]^T + bias[n] with IEEE-f16 operands in the layouts of cruse_gemm_bf16_nt, f32 accumulation
 * (v_mfma_f32_16x16x32_f16): the FORWARD gate projection gi = x W_ih^T (cruse_net.py:23-31,44,50) in ONE pass -- 11 significant bits
 * on both operands, where the split-bf16 form above spends a second pass to correct W_ih only and keeps x at 8 bits.  The operands
 * are O(1) activations (BatchNorm + ReLU / LayerNorm outputs) and |W_ih| <= 1 / sqrt(Hg): inside f16's range. */
int cruse_gemm_f16_nt(int M, int N, int K, const void* A, long long lda, long long a_kstride,
                      const void* B, long long ldb, long long b_kstride,
                      float* C, long long ldc, const float* bias, void* stream);
/* (ABI 12) C = A . (B_hi + B_lo)^T + bias[n]: f16 activations against TWO f16 planes of the weights in one launch (the k-range is walked twice on the
 * same accumulators) -- the forward gate projection of GGRU layer 1 (nn.GRU input projection, cruse_net.py:23-31): 11 bits on x, ~20 on W_ih. */
int cruse_gemm_f16x2_nt(int M, int N, int K, const void* A, long long lda, long long a_kstride,
                        const void* B_hi, const void* B_lo, long long ldb, long long b_kstride,
                        float* C, long long ldc, const float* bias, void* stream);
/* y = bf16(x) and y_lo = bf16(x - y) (nullable) */
int cruse_cast_bf16_split(const float* x, void* y, void* y_lo, long long n, void* stream);
/* Split-bf16 x3 form of cruse_gemm_bf16_nt: A = A_hi + A_lo, B = B_hi + B_lo (bf16 planes, same layout each);
 * C = A_hi.B_hi + A_hi.B_lo [+ A_lo.B_hi] accumulated in one pass.  A_lo may be NULL: two passes, only B's rounding
 * is corrected.  Used for the FORWARD gate projection gi = x W_ih^T: the bf16 rounding of W_ih dominates the
 * forward error of CRUSE_PREC_BF16 (enhanced spectrum 1.25e-3; 5.1e-4 with W split, 4.8e-4 with both).  No split-K. */
int cruse_gemm_bf16x3_nt(int M, int N, int K, const void* A_hi, const void* A_lo, long long lda, long long a_kstride,
                         const void* B_hi, const void* B_lo, long long ldb, long long b_kstride,
                         float* C, long long ldc, const float* bias, int accumulate, void* stream);
/* K-tiled time-major transpose: yT[(r/64)*cols*64 + c*64 + r%64] = bf16(x[(r - s)*ld + c]) with s = 0, or s = 1
 * when shift_T > 0 (then 0 where r % shift_T == 0: the h_{t-1} operand of dW_hh); frames rows <= r < ldT are
 * zero-filled (ldT % 64 == 0). */
int cruse_transpose_bf16(const float* x, long long rows, int cols, long long ld, void* yT, long long ldT,
                         int shift_T, void* stream);

/* ---- grouped-GRU recurrence (nn.GRU forward/backward at cruse_net.py:44,50) ----- */

/* Persistent recurrence.  gi [B,T,G,3*Hg] = x W_ih^T + b_ih (gate order r,z,n) from cruse_gemm;
 * w_hh[g] -> [3*Hg,Hg], b_hh[g] -> [3*Hg] (HOST arrays of G device pointers); h0 = 0.
 * Output h [B,T,G*Hg] in "cat" layout (feature = g*Hg + j).  For backward (all three or none):
 *   coef [B,T,G,3*Hg] = d(W_hh h + b_hh)-gradient coefficients (c_r, c_z, c_n): dgh_t = dh_t * coef_t
 *                       (bf16 elements when prec == CRUSE_PREC_BF16, f32 otherwise),
 *   an   [B,T,G*Hg]   = dgi_n coefficient ((1-z)(1-n^2)),   z [B,T,G*Hg] = update gate.
 * Hg % 32 == 0, Hg <= 1024 (the BACKWARD recurrence keeps a [16][3*Hg] operand image in LDS: Hg <= 800 in the f32 and split-bf16
 * modes, 1024 in CRUSE_PREC_BF16; larger is refused with CRUSE_E_SHAPE).  ws: cruse_gru_ws_bytes() bytes of device scratch.  Its first 256 bytes are a STICKY
 * header the CALLER zeroes once when it allocates the buffer: word 0 becomes non-zero if a hand-off ever timed out
 * (the launch then completes unsynchronised and its outputs are garbage) and is never cleared by the library, so it
 * can be polled once per epoch and handed to cruse_adam_step_guarded as skip_flag.  The hand-off panels behind the
 * header are zeroed by the callee on every call. */
size_t cruse_gru_ws_bytes(int B, int G, int Hg);
/* The launch plan cruse_gru_seq_fwd (fwd != 0) / cruse_gru_seq_bwd take for this shape and precision with h0 = 0 and no explicit chain width:
 * out[0] clips per chain (8, or 16: the wide-chain kernels of gru_w16.hip), out[1] chains per group, out[2] chains per group and launch,
 * out[3] launches, out[4] workgroups per chain (Hg / 32), out[5] 1 = wide-chain kernels.  A query: no device work.  Replaces nothing in the
 * reference (nn.GRU has no launch plan, model/cruse_net.py:23-31); it lets tests and bench.py state which kernels a batch size runs on. */
int cruse_gru_plan(int B, int G, int Hg, int prec, int fwd, int* out);
int cruse_gru_seq_fwd(const float* gi, const float* const* w_hh, const float* const* b_hh,
                      float* h, void* coef, float* an, float* z,
                      int B, int T, int G, int Hg, int prec, void* ws, void* stream);
/* dout = dL/dh [B,T,G*Hg] -> dh [B,T,G*Hg] = total gradient reaching h_t (incl. the recurrent path):
 * dh_s = dout_s + z_{s+1}*dh_{s+1} + (dh_{s+1}*coef_{s+1}) W_hh. */
int cruse_gru_seq_bwd(const float* dout, const float* const* w_hh, const void* coef, const float* z,
                      float* dh, int B, int T, int G, int Hg, int prec, void* ws, void* stream);
/* The same two launches for CONCURRENT recurrences (half batches on two streams, so that the projections / layer norms
 * of one half run beside the recurrence of the other): the workspace is given as a per-launch panel scratch of
 * cruse_gru_ws_bytes(B,G,Hg) - 256 bytes (zeroed here) plus a caller-owned sticky status word that several launches may
 * share, and chain c of the launch runs on XCD (c + xcd_rot) % 8 instead of c % 8.  A recurrence workgroup owns its CU;
 * two launches of <= 4 chains with xcd_rot 0 and 4 occupy disjoint XCDs and are co-resident, with the same xcd_rot they
 * would queue for the same CUs.  cruse_gru_seq_fwd/bwd(ws) == _on(ws + 256, (unsigned*)ws, 0).
 * _bwd_on: dgi != NULL (CRUSE_PREC_BF16, with the a_n rows `an`): also writes the bf16 gate gradients
 * dgi [B,T,G,3*Hg] = dh * (c_r, c_z, a_n) -- the dgi output of cruse_gru_gate_grads_bf16 -- from inside the recurrence, so
 * that dX = dgi W_ih can start right behind it; cruse_gru_gate_grads_bf16(dgi = NULL) then only makes dgT / the bias sums. */
int cruse_gru_seq_fwd_on(const float* gi, const float* const* w_hh, const float* const* b_hh,
                         float* h, void* coef, float* an, float* z,
                         int B, int T, int G, int Hg, int prec, void* panels, unsigned* status, int xcd_rot, void* stream);
int cruse_gru_seq_bwd_on(const float* dout, const float* const* w_hh, const void* coef, const float* z,
                         float* dh, const float* an, void* dgi, int B, int T, int G, int Hg, int prec,
                         void* panels, unsigned* status, int xcd_rot, void* stream);
/* SUB-SEQUENCES and INITIAL STATE (cust_conv.py:305-325 passes h0 explicitly; GroupGRU.forward(input, state) :392-416).
 * T steps of tensors whose clips are TS >= T frames apart; every pointer is already advanced to the first frame of the run.
 *   _fwd_ex: h0 [B][G*Hg] ("cat" feature layout, clips h0_bstride floats apart, 16-byte aligned) or NULL (= 0).  A long
 *            sequence can be run as consecutive time chunks: chunk [t0, t0+n) is (gi + t0*G*3*Hg, h + t0*G*Hg, ..., h0 =
 *            h + (t0-1)*G*Hg with h0_bstride = TS*G*Hg, T = n) -- the results are those of the single launch.
 *   _bwd_ex: carry != 0: the first iteration takes dh of the run's LAST frame from the dh buffer (written by the run that
 *            follows it in time) instead of forming it from dout: chunk [t0, t0+n) of a longer sequence is run as the
 *            n + 1 frames [t0, t0+n] with carry = 1 (the last chunk: n frames, carry = 0).  dgi on a sub-sequence is
 *            supported by the reduce-scatter kernels only (CRUSE_PREC_BF16, Hg <= 640).
 *            dg_slabs: 3 = dgi rows [G][3][Hg] (r, z, n_i); 4 = [G][4][Hg] with slab 3 = dh * c_n, the n gate of
 *            dgh = dh * (c_r, c_z, c_n) (reduce-scatter kernel): one row-major tensor that is the A operand of
 *            dX = dgi W_ih (K = 3*Hg of every 4*Hg) (the row-major TN weight-gradient products that also read it are gone: measured slower, r3).
 *   panels_zeroed: the caller has cleared the panel scratch (cruse_gru_ws_bytes() - 256 bytes at `panels`) since its last use, on
 *            this stream or ordered before it; 0: the call clears it itself (one memset launch in front of the recurrence).  A
 *            training step clears the scratches of its four recurrences with one launch at its top.
 *   chain_clips: clips served by one team of Hg/32 workgroups.  0 = the library's plan: chains of 8 while the batch's chains fit
 *            the CUs; beyond that (B > 96 at Hg = 640) WIDE chains of 16 (half the workgroups per clip, the full 16 columns of the
 *            MFMA; CRUSE_PREC_BF16, Hg % 128 == 0, Hg <= 640, h0 == NULL) where that needs fewer launches.
 *            8 / 16 force the width: with 16 a batch of 64 at Hg = 640 takes 80 CUs, so the recurrences of BOTH GGRU layers are
 *            co-resident (measured slower than one after the other on chains of 8: DESIGN.md section 8).  The forward
 *            results do not depend on the width (same sums in the same order); the backward ones up to the f32 summation order.
 *            chain_clips = 16 WITH h0: the caller vouches for |h0| < 1 (the wide kernels' hand-off keeps the epoch bit in the top
 *            exponent bit of every exchanged bf16) -- true for the continuation of a sequence that started from h0 = 0. */
int cruse_gru_seq_fwd_ex(const float* gi, const float* const* w_hh, const float* const* b_hh,
                         float* h, void* coef, float* an, float* z, const float* h0, long long h0_bstride,
                         int B, int T, int TS, int G, int Hg, int prec, int chain_clips, void* panels,
                         int panels_zeroed, unsigned* status, int xcd_rot, void* stream);
int cruse_gru_seq_bwd_ex(const float* dout, const float* const* w_hh, const void* coef, const float* z,
                         float* dh, const float* an, void* dgi, int dg_slabs, int carry, int B, int T, int TS, int G,
                         int Hg, int prec, int chain_clips, void* panels, int panels_zeroed, unsigned* status, int xcd_rot,
                         void* stream);
/* (ABI 13) gi rows stored as IEEE f16 ([B,T,G,3*Hg] halves, written by cruse_gemm_nt_out16): half the bytes of the largest tensor of the forward
 * pass.  gi_dtype: CRUSE_DT_F16, or CRUSE_DT_F32 (then exactly cruse_gru_seq_fwd_ex with h0 = NULL, chain_clips = 0).  f16 rows are served where the
 * bench step runs -- CRUSE_PREC_BF16, Hg = 640, chains of 8 clips (cruse_gru_plan: B <= 96) -- and refused with CRUSE_E_SHAPE elsewhere.
 * Replaces the recurrence of nn.GRU (model/cruse_net.py:23-31, :41-51) like cruse_gru_seq_fwd_ex. */
int cruse_gru_seq_fwd_gi16(const void* gi, int gi_dtype, const float* const* w_hh, const float* const* b_hh,
                           float* h, void* coef, float* an, float* z,
                           int B, int T, int TS, int G, int Hg, int prec, void* panels,
                           int panels_zeroed, unsigned* status, int xcd_rot, void* stream);
/* dgi = dh*(c_r,c_z,a_n) (gradient wrt gi), dgh = dh*(c_r,c_z,c_n) (gradient wrt W_hh h + b_hh), both
 * [rows,G,3*Hg]; dW_ih, dW_hh, dX and the bias gradients follow from cruse_gemm / cruse_col_sum. */
int cruse_gru_gate_grads(const float* dh, const void* coef, const float* an, float* dgi, float* dgh,
                         long long rows, int G, int Hg, int prec, void* stream);
/* CRUSE_PREC_BF16 form feeding cruse_gemm_bf16_nt (coef bf16).  dgi and dgh share the r and z gates, so four
 * slabs (r, z, n_i, n_h) carry both: dgi [rows,G,3,Hg] bf16 = (r,z,n_i) row-major; dgT [ldT/64,G,4,Hg,64] bf16 =
 * the K-tiled time-major transpose (ldT = rows rounded up to 64, zero padded): dW_ih uses slabs 0-2, dW_hh 0,1,3.
 * db_ih / db_hh: HOST arrays of G device pointers (or NULL) to [3*Hg] bias gradients, ACCUMULATED from f32. */
/* dgi or dgT may be NULL (not both): only the other output (and the bias sums) is produced */
int cruse_gru_gate_grads_bf16(const float* dh, const void* coef, const float* an, void* dgi, void* dgT,
                              long long ldT, float* const* db_ih, float* const* db_hh,
                              long long rows, int G, int Hg, void* stream);

/* ---- mask application + weighted spectral loss ------------------------------- */

/* PreProcess.masking "mag_mapping" (utils/utils.py:418-420) fused with WO-MALE
 * (loss_func/loss.py:121-148, alpha/(beta+iam), gamma = 1) and its gradient.
 * mask [rows,Fn]; nre, nim [rows,Fs] noisy spectrum; cmag [rows,Fs] = |clean|; bins Fn..Fs-1 of the
 * estimate are zero.  loss_sum[0] = sum W*|log10(|est|+1) - log10(|ref|+1)| (f64, zeroed by callee;
 * the loss is loss_sum/(rows*Fs)).  Optional outputs: dmask [rows,Fn] = d(loss)/d(mask),
 * dlogit [rows,Fn] = dmask*mask*(1-mask), est_re/est_im [rows,Fs]. */
int cruse_mask_loss_fwd(const float* mask, const float* nre, const float* nim, const float* cmag,
                        long long rows, int Fn, int Fs, float alpha, float beta,
                        double* loss_sum, float* dmask, float* dlogit, float* est_re, float* est_im,
                        void* stream);

/* ---- time-domain loss in the loop ---------------------------- */

/* PreProcess.masking "mag_mapping" alone (utils/utils.py:418-420): est = mask * noisy spectrum on the first Fn
 * bins, zero above.  mask [rows,Fn]; nre, nim, est_re, est_im [rows,Fs]. */
int cruse_mask_apply(const float* mask, const float* nre, const float* nim, long long rows, int Fn, int Fs,
                     float* est_re, float* est_im, void* stream);
/* its backward: dout[rows,Fn] = (dre*nre + dim*nim) [* mask*(1-mask) when through_sigmoid != 0] */
int cruse_mask_apply_bwd(const float* dre, const float* dim, const float* nre, const float* nim,
                         const float* mask, long long rows, int Fn, int Fs, int through_sigmoid,
                         float* dout, void* stream);
/* SNR-weighted speech-distortion loss `sdnr` (loss_func/loss.py:151-175, vad == 1) with the mask as gain:
 * loss = loss_sum[0] / (B*Fs) = alpha*L_speech + (1-alpha)*L_noise, alpha = 10^(snr/10)/(10^(snr/10)+10^(beta/10)).
 * cre/cim clean, nre/nim noisy spectra [rows,Fs] (noise = noisy - clean); optional dmask / dlogit [rows,Fn]. */
int cruse_mask_sdnr_fwd(const float* mask, const float* cre, const float* cim, const float* nre, const float* nim,
                        long long rows, int Fn, int Fs, int B, float snr_db, float beta_db,
                        double* loss_sum, float* dmask, float* dlogit, void* stream);
/* si_snr_loss of train_base/loss.py:7-25 on waveforms x (estimate), s (target) [B,L]:
 * loss[0] = -mean_b 20 log10(eps + |t|/(|x_zm - t| + eps)) (f64; zeroed by the callee).
 * mom [B,5] f64 scratch; coef [B,4] f32 receives the per-clip gradient scalars for cruse_sisnr_bwd. */
int cruse_sisnr_fwd(const float* x, const float* s, int B, int L, float eps,
                    double* mom, double* loss, float* coef, void* stream);
/* dx = grad_scale * d loss / d x */
int cruse_sisnr_bwd(const float* x, const float* s, const float* coef, int B, int L, float grad_scale,
                    float* dx, void* stream);
/* l1_loss / mse_loss = torch.nn.L1Loss / MSELoss (train_base/loss.py:3-4; selected by tools/train_stand.py:73-75) on waveforms:
 * loss_sum[0] = sum |est - ref| (mse = 0) or sum (est - ref)^2 (mse = 1) over n samples (f64; zeroed by the callee);
 * dest (optional, n floats) = grad_scale * sign(est - ref)  resp.  grad_scale * 2 (est - ref): pass 1/n for reduction "mean". */
int cruse_wave_l1_mse(const float* est, const float* ref, long long n, int mse, float grad_scale,
                      double* loss_sum, float* dest, void* stream);

/* ---- DeepFilter head (model/deep_filter.py:15-41, BASELINE config 4) --------------- */
/* out[b,f,t] = sum over the (2*f_dim+1) x (2*t_dim+1) neighbourhood of X*H (complex, zero outside); all
 * tensors [B,F,T] in the reference's layout.  Imaginary part = xr*hi + xi*hr (decision: SURVEY 8a a15). */
int cruse_deepfilter_fwd(const float* xr, const float* xi, const float* hr, const float* hi,
                         int B, int F, int T, int f_dim, int t_dim, float* out_r, float* out_i, void* stream);
int cruse_deepfilter_bwd(const float* dout_r, const float* dout_i, const float* xr, const float* xi,
                         const float* hr, const float* hi, int B, int F, int T, int f_dim, int t_dim,
                         float* dxr, float* dxi, float* dhr, float* dhi, void* stream);

/* backward of nn.Sigmoid (cruse_net.py:164): dlogit = dmask * mask * (1 - mask) */
int cruse_sigmoid_bwd(const float* dmask, const float* mask, float* dlogit, long long n, void* stream);

/* out = a*x + b*y elementwise (x or y may alias out) */
int cruse_axpby(float* out, const float* x, const float* y, float a, float b, long long n, void* stream);

/* ---- general NCHW blocks: the reference's other conv-recurrent modules (cust_conv.py, mtfaa.py) ---------------- */

/* nn.Conv2d / nn.ConvTranspose2d with arbitrary kernel, stride, dilation, groups and zero padding on [B,C,H,W]
 * (model/based_model/cust_conv.py:47-55,94-104,150-166; model/mtfaa.py:60-73,170-183), optional nearest
 * FreqUpsample folded into the gather index (cust_conv.py:163-166,177-184) and a fused ReLU / PReLU.
 *   transposed 0: y[b,co,ho,wo] = bias[co] + sum w[co][ci_l][kh][kw] * X[b,ci, ho*sh-pt+kh*dh, wo*sw-pl+kw*dw],
 *                 X = zero-padded x (W index divided by up_w when up_w > 1; Win is the size before upsampling)
 *   transposed 1: gather form of ConvTranspose2d(padding=(pt,pl)), weight [Cin][Cout/g][KH][KW]
 * The data gradient of either form is the other form with the same weight tensor.  act 0 none, 1 ReLU, 2 PReLU(slope[co]).
 * dtype (CRUSE_DT_*): storage type of x / y (and of S, Bg, dy, dx in the functions below); weights, bias, dw stay f32. */
int cruse_conv2d_nchw(const void* x, const float* w, const float* bias, void* y,
                      int B, int Cin, int Hin, int Win, int Cout, int Hout, int Wout,
                      int KH, int KW, int sh, int sw, int dh, int dw, int pt, int pl,
                      int groups, int up_w, int transposed, int act, const float* slope, int accumulate, int dtype,
                      void* stream);
/* dw[ca][cb_l][kh][kw] += sum_{n,h,w} S[n,ca,h,w] * Bg[n, g*CB/groups + cb_l, h*sh-pt+kh*dh, (w*sw-pl+kw*dw_)/up_w]
 * Conv2d: S = dy, Bg = x.  ConvTranspose2d: S = x, Bg = dy. */
int cruse_conv2d_nchw_wgrad(const void* S, const void* Bg, float* dw,
                            int N, int CA, int HS, int WS, int CB, int HB, int WB,
                            int KH, int KW, int sh, int sw, int dh, int dw_, int pt, int pl,
                            int groups, int up_w, int dtype, void* stream);
/* (round 5, ABI 11) the same for nn.Conv2d (S = dy) with the bias gradient db[ca] += sum_{n,h,w} dy[n,ca,h,w] in the same call: inside the
 * pointwise MFMA kernel (the constant 1 rides a spare column of the last column tile), else by cruse_nchw_channel_sum.  db nullable. */
int cruse_conv2d_nchw_wgrad_ex(const void* S, const void* Bg, float* dw, float* db,
                               int N, int CA, int HS, int WS, int CB, int HB, int WB,
                               int KH, int KW, int sh, int sw, int dh, int dw_, int pt, int pl,
                               int groups, int up_w, int dtype, void* stream);
/* out[c] += sum_{n,hw} x[n,c,hw] (conv bias gradient) */
int cruse_nchw_channel_sum(const void* x, int N, int C, int HW, float* out, int dtype, void* stream);
/* gradient of the nearest FreqUpsample: dx[..,w] = sum_{j<up} dxu[.., w*up + j] */
int cruse_downsum_w(const void* dxu, long long rows, int W, int up, void* dx, int dtype, void* stream);
/* nearest FreqUpsample (cust_conv.py:177-184) of a tensor whose last axis is W: xu[.., w*up + j] = x[.., w]; the frame-major
   upsample decoder (model/cruse.py:14, convkxf mode="upsample") materialises it for its (1,3) conv and that conv's weight gradient */
int cruse_upsample_w(const void* x, long long rows, int W, int up, void* xu, int dtype, void* stream);
/* nn.BatchNorm2d (+ nn.ReLU / nn.PReLU(C) / nn.Sigmoid: act 1 / 2 / 3) on [N,C,HW]: batch sums for cruse_bn_finalize; y = act(gamma*(x-mean)*rstd+beta)
 * (mean == NULL: activation only); backward with the PReLU slope gradient.  scratch: 3*C doubles. */
int cruse_bn_nchw_stats(const void* x, int N, int C, int HW, double* sums, int dtype, void* stream);
int cruse_bn_nchw_fwd(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                      const float* slope, int act, int N, int C, int HW, void* y, int dtype, void* stream);
int cruse_bn_nchw_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, const float* slope, int act, int training, int N, int C, int HW,
                      double* scratch, void* dx, float* dgamma, float* dbeta, float* dslope, int dtype, void* stream);

/* Round 5 (ABI 11): the same two passes with their small companions folded in.
 * cruse_bn_nchw_fwd_train: training-mode nn.BatchNorm2d (+ act) from the batch sums of cruse_bn_nchw_stats -- mean / rstd are formed inside the
 *   kernel and published to mean_out / rstd_out, running_mean / running_var (nullable pair: momentum, unbiased variance) and
 *   num_batches_tracked (nullable, += 1) are updated: cruse_bn_finalize + cruse_counters_add + cruse_bn_nchw_fwd in one launch
 *   (nn.Conv2d -> nn.BatchNorm2d -> nn.PReLU of TFCM_Block, mtfaa.py:166-193; Conv2dNormAct, cust_conv.py:15-111).
 *   sums may be sum_replicas x [2C] (cruse_conv2d_nchw_ex): the statistic is the sum over the replicas.
 * cruse_conv2d_nchw_ex: cruse_conv2d_nchw (no accumulation) with what follows a convolution in the reference's blocks folded in, both optional:
 *   residual != NULL: y = conv(x) + residual (TFCM_Block, mtfaa.py:191) in the epilogue of the f16 LDS-transposed pointwise kernel (else by
 *   cruse_add_nchw); bn_sums != NULL: the batch sums of the output into bn_sums [bn_nrep][2*Cout] f64 (cleared by the caller) from the
 *   epilogue of the f16 pointwise-MFMA / LDS-depthwise kernels (else the statistics pass over y into replica 0).  Not both.
 * cruse_bn_nchw_bwd_ex: cruse_bn_nchw_bwd (scratch: 4*C doubles) whose apply pass also adds the parameter gradients and, when dx_sum != NULL,
 *   dx_sum[c] += sum over the channel of dx -- the bias gradient of the convolution in front of the BatchNorm -- in closed form from the
 *   reduce pass's sums (training: -gamma rstd mean(d xhat) sum(xhat), the rounding of the batch mean; eval: gamma rstd sum(d)) instead of a
 *   channel-sum pass over dx. */
int cruse_bn_nchw_fwd_train(const void* x, const double* sums, int sum_replicas, float eps, float momentum, const float* gamma, const float* beta,
                            const float* slope, int act, int N, int C, int HW, void* y, float* mean_out, float* rstd_out,
                            float* running_mean, float* running_var, long long* num_batches_tracked, int dtype, void* stream);
int cruse_conv2d_nchw_ex(const void* x, const float* w, const float* bias, const void* residual, void* y,
                         int B, int Cin, int Hin, int Win, int Cout, int Hout, int Wout,
                         int KH, int KW, int sh, int sw, int dh, int dw, int pt, int pl,
                         int groups, int up_w, int transposed, int act, const float* slope,
                         double* bn_sums, int bn_nrep, int dtype, void* stream);
int cruse_bn_nchw_bwd_ex(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                         const float* beta, const float* slope, int act, int training, int N, int C, int HW,
                         double* scratch, int scratch_zeroed, int sums_replicas, void* dx, float* dgamma, float* dbeta, float* dslope, float* dx_sum,
                         int dtype, void* stream);
/* The data gradient of a convolution whose INPUT was a BatchNorm2d (+ act) output: y = conv(x) (no bias / act / accumulation) is the gradient wrt that
 * output; the f16 pointwise-MFMA / LDS-depthwise kernels also accumulate the BatchNorm's backward sums of the stored y against bn_x (its input) into
 * r [r_nrep][4][Cout] f64 (cleared by the caller): cruse_bn_nchw_bwd_ex(scratch = r, sums_replicas = r_nrep) then skips its reduce pass.
 * *delivered = 0: the form that ran has no such epilogue -- run the plain backward (sums_replicas = 0).
 * Replaces autograd's conv_backward(input) + the reduction half of batch_norm_backward for nn.Conv2d after nn.BatchNorm2d + nn.PReLU
 * (TFCM_Block, mtfaa.py:170-183; Conv2dNormAct, cust_conv.py:15-111). */
int cruse_conv2d_nchw_bnbwd(const void* x, const float* w, void* y,
                            int B, int Cin, int Hin, int Win, int Cout, int Hout, int Wout,
                            int KH, int KW, int sh, int sw, int dh, int dw, int pt, int pl, int groups, int transposed,
                            const void* bn_x, const float* bn_mean, const float* bn_rstd, const float* bn_gamma, const float* bn_beta,
                            const float* bn_slope, int bn_act, double* r, int r_nrep, int* delivered, int dtype, void* stream);
/* cruse_bn_nchw_stats into sums the caller cleared itself (zeroed != 0; like scratch_zeroed above: one fill launch per pool chunk of
 * accumulators instead of one per call) */
int cruse_bn_nchw_stats_ex(const void* x, int N, int C, int HW, double* sums, int zeroed, int dtype, void* stream);

/* out = a + b on n elements of storage type dtype (TFCM_Block residual, mtfaa.py:191; GroupGRU add_outputs, cust_conv.py:411-412) */
int cruse_add_nchw(const void* a, const void* b, void* out, long long n, int dtype, void* stream);
/* f32 -> f16 (to_f16 != 0) or f16 -> f32 copy of n elements: where the fp16 part of a model (BASELINE config 5) begins / ends */
int cruse_cast_f16(const void* src, void* dst, long long n, int to_f16, void* stream);

/* ---- other STFT formulations (feature.py:272-398 CustomSTFT/CustomISTFT, conv_stft.py:8-129, mtfaa.py:8-37) ------- */
/* X[b,t,f] = scale * sum_{n<win_len} window[n] * x_pad[b, t*hop + n + win_off - pad] * exp(-2 pi i f (n + win_off)/n_fft),
 * f = 0..n_fft/2; x_pad: pad_mode 0 zeros / 1 reflect outside the clip.  re, im [B,T,n_fft/2+1]. */
int cruse_stft_framed(const float* wave, const float* window, int B, int L, int n_fft, int win_len, int win_off,
                      int hop, int pad, int pad_mode, int T, float scale, float* re, float* im, void* stream);
/* overlap-add adjoint: out[b,m] = post[m] * sum_{t,n: t*hop+n+win_off-pad == m} window[n]*scale*sum_f c_f (re cos - im sin);
 * hermitian 0: c_f = 1 (conv_transpose1d with the analysis kernel, feature.py:395); 1: c_f = 1,2,..,2,1 (inverse DFT of a
 * one-sided spectrum, conv_stft.py:106-122).  post: L per-sample factors (1 / window-overlap envelope) or NULL. */
int cruse_istft_framed(const float* re, const float* im, const float* window, const float* post, int B, int T,
                       int n_fft, int win_len, int win_off, int hop, int pad, int L, float scale, int hermitian,
                       float* out, void* stream);

/* ---- masks, further losses, data synthesis ------------------------------------------------------------------------ */
/* train_base/acoustics/mask.py:8-63.  mode 0 IRM (a = noisy_mag, c = clean_mag), 1 cIRM (a,b = noisy re,im; c,d = clean
 * re,im; out [n,2]), 2 compress_cIRM(a), 3 decompress_cIRM(a), 4 complex_mul (a + ib)(c + id) -> out, out2;
 * PreProcess (utils/utils.py:414-423): 5 pair product out = a*c, out2 = b*d ("complex_mapping"), 6 out = log(a).
 * Adjoints (the reference's ops are plain torch, hence differentiable): 7 out = c * compress'(a), 8 out = c * decompress'(a)
 * (c = upstream gradient), 9 (a + ib) conj(c + id) -> out, out2 (both gradients of mode 4). */
int cruse_mask_ops(int mode, const float* a, const float* b, const float* c, const float* d, long long n,
                   float K, float C, float limit, float* out, float* out2, void* stream);
/* mode 0: (re, im) -> (sqrt(re^2+im^2+eps)**alpha, atan2(im, re)) (feature.py:363-364, mtfaa.py:136-137,162);
 * mode 1: (mag, phase) -> (mag cos, mag sin) (feature.py:386-387); mode 2: gradient of mode 0 (g = d mag, g2 = d phase,
 * either may be NULL) -> (d re, d im); mode 3: gradient of mode 1 (a, b = mag, phase; g, g2 = d re, d im) -> (d mag, d phase). */
int cruse_polar(int mode, const float* a, const float* b, const float* g, const float* g2, long long n, float eps, float alpha,
                float* o1, float* o2, void* stream);
/* rmse (loss_func/loss.py:59-78): loss_sum = sum |est - ref| (divide by B*T*F); dest = sign(est-ref)*grad_scale (may be NULL) */
int cruse_rmse(const float* ref, const float* est, long long n, float grad_scale, double* loss_sum, float* dest, void* stream);
/* c_rmse (loss_func/loss.py:88-118) as written; ref, est [B,2,TF]; dest = d loss / d est (may be NULL) */
int cruse_c_rmse(const float* ref, const float* est, int B, long long TF, float c, float beta, double* loss_sum, float* dest,
                 void* stream);
/* wo_male (loss_func/loss.py:121-148) on explicit spectra ref, est, unproc: element (b, j) has its real part at
 * b*bstride + j and its imaginary part pstride further ([B,2,TF]: 2*TF, TF; [2,B,TF]: TF, B*TF).  loss_sum (divide by
 * B*T*F) and dest = grad_scale * d loss_sum / d est (may be NULL).  The mask-only step uses the fused cruse_mask_loss_fwd;
 * the DeepFilter step (BASELINE config 4) uses this form on the filtered spectrum. */
int cruse_wo_male_spec(const float* ref, const float* est, const float* unproc, int B, long long TF, long long bstride,
                       long long pstride, float alpha, float beta, float grad_scale, double* loss_sum, float* dest, void* stream);
/* sisnr (loss_func/loss.py:48-56, no mean removal) from the moments cruse_sisnr_fwd's first pass leaves in `mom`:
 * value = mean_b 10 log10(.), coef [B,4] for cruse_sisnr_bwd (gradient of the VALUE; the loss is its negative) */
int cruse_sisnr_plain_finalize(const double* mom, int B, float eps, double* value, float* coef, void* stream);
/* synthetic clips: y[b,n] = gain*(1-a) * sum_{k<taps} a^k x[b,n-k], a one-pole low-pass as a truncated FIR */
int cruse_onepole_fir(const float* x, int B, int L, float a, int taps, float gain, float* y, void* stream);
/* y[b,n] = sum_{k < R, k <= n} h[b or 0][k] x[b,n-k] = scipy.signal.fftconvolve(x, h)[:L]: the room-impulse-response
 * convolution at the top of SynDataset.snr_mix (dataset/dataset.py:245-248).  h_bstride = 0: one tap row for all clips,
 * else tap rows h_bstride >= R floats apart.  Out of place. */
int cruse_fir_causal(const float* x, const float* h, long long h_bstride, int B, int L, int R, float* y, void* stream);
/* SynDataset.snr_mix (dataset/dataset.py:236-264) for B clips at once: peak-normalise both, scale the noise to snr_db[b]
 * from the RMS ratio, mix.  scratch: 24*B bytes.  clean_out / noise_out may be NULL. */
int cruse_snr_mix(const float* clean, const float* noise, const float* snr_db, int B, int L, float eps,
                  void* scratch, float* clean_out, float* noise_out, float* noisy, void* stream);

/* ---- scheduling aids ------------------------------------------------------------------------------------------------ */
/* a HIP stream restricted to the CUs whose bit is set in mask[0..nwords) (hipExtStreamCreateWithCUMask) */
int cruse_stream_create_masked(void** stream_out, const unsigned* mask, int nwords);
/* out[2*b], out[2*b+1] = HW_REG_XCC_ID, HW_REG_HW_ID of block b (each block idles `spin` clock ticks): placement census */
int cruse_cu_census(unsigned* out, int nblocks, unsigned spin, void* stream);
/* test / probe rig: nblocks workgroups that each HOLD a CU (128 KB of LDS) for `ticks` shader clocks -- what a collective's
 * channels or another tenant do to the persistent recurrences, whose teams must be co-resident (tests/test_gpu_ddp.py) */
int cruse_cu_hog(int nblocks, unsigned long long ticks, void* stream);

/* ---- stream-ordered bookkeeping (keeps the training step free of library kernels) ---- */
/* zero-fill `bytes` (multiple of 4) with a KERNEL node (hipMemsetAsync nodes raced inside captured graphs) --
 * optimizer.zero_grad() at the top of the step */
int cruse_zero(void* p, size_t bytes, void* stream);
/* acc[i] += x[i] (f64): running loss sum of an epoch without a host synchronisation per step */
int cruse_accum_f64(double* acc, const double* x, int n, void* stream);
/* *counters[i] += v for a HOST array of n <= 32 device int64 pointers: BatchNorm2d.num_batches_tracked */
int cruse_counters_add(long long* const* counters, int n, long long v, void* stream);

/* ---- optimizer (torch.optim.Adam at tools/train_stand.py:68-71) -------------- */
/* One fused Adam step over a flat parameter buffer; g is multiplied by grad_scale first
 * (1/world_size after a sum all-reduce). step >= 1. */
int cruse_adam_step(float* p, const float* g, float* m, float* v, long long n,
                    float lr, float beta1, float beta2, float eps, float weight_decay,
                    int step, float grad_scale, void* stream);
/* The same step behind device-side guards, so that one bad batch cannot poison the parameters and no host
 * synchronisation is needed to decide (all pointers optional):
 *   skip_flag  : n_skip_words u32 words; any non-zero word skips the step (the sticky GRU hand-off status word of
 *                cruse_gru_seq_fwd, or the -- possibly all-reduced -- health words of cruse_step_health);
 *   loss_check : a non-finite *loss_check skips the step (the reference has no such check: a 0/0 bin in wo_male,
 *                loss_func/loss.py:141, would turn every parameter into NaN);
 *   gsumsq     : sum of squares of g (cruse_sumsq); a non-finite norm skips the step; with max_norm > 0 the gradient is
 *                scaled by min(1, max_norm / (sqrt(gsumsq)*grad_scale + 1e-6)) = torch.nn.utils.clip_grad_norm_
 *                (train_base/trainer/base_trainer.py:75 `clip_grad_norm_value`);
 *   skipped    : skipped[0] += 1 for every skipped step; with n_skip_words > 1 also skipped[1 + i] += 1 for each
 *                non-zero word i (per-reason counters).  The bias corrections use step - skipped[0] (the value BEFORE this
 *                call): a skipped step does not advance Adam's step, as if torch.optim.Adam.step() had not been called;
 *   loss_sum / loss_acc (ABI 7): when the step is APPLIED, loss_acc[0] += *loss_sum * loss_scale and loss_acc[1] += 1 --
 *                the running epoch loss counts exactly the steps that reached the parameters (a timed-out step's loss is
 *                garbage even when finite), each with its own normalisation. */
int cruse_adam_step_guarded(float* p, const float* g, float* m, float* v, long long n,
                            float lr, float beta1, float beta2, float eps, float weight_decay,
                            int step, float grad_scale, float max_norm, const double* gsumsq,
                            const unsigned* skip_flag, int n_skip_words, const double* loss_check, unsigned* skipped,
                            const double* loss_sum, double loss_scale, double* loss_acc, void* stream);
/* Per-step health of a training step, decided on the device (no reference counterpart: train/trainer_casual.py is empty
 * and base_trainer.py has no guard).  health[0] = the GRU status word was set -- and CLEARS it, so one transient
 * hand-off time-out costs the step it happened in, not every later one; health[1] = *loss_sum is not finite.
 * loss_acc (optional) += *loss_sum * loss_scale when finite (the running epoch loss, each step with its own norm).
 * Data-parallel callers all-reduce health[0..1] with MAX before handing it to cruse_adam_step_guarded, so that every
 * rank takes the same skip decision.  gru_status may be NULL. */
int cruse_step_health(unsigned* gru_status, const double* loss_sum, unsigned* health, double* loss_acc,
                      double loss_scale, void* stream);
/* out (+)= sum x[i]^2 in f64 (total gradient norm for clip_grad_norm_); x 16-byte aligned */
int cruse_sumsq(const float* x, long long n, double* out, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CRUSE_HIP_H */
