// General NCHW building blocks for the reference's OTHER conv-recurrent blocks -- the callers either side of the
// unet_2 hot path (8f item 2):
//   model/based_model/cust_conv.py:15-174  Conv2dNormAct / ConvTranspose2dNormAct / convkxf (normal, transposed,
//                                           upsample = nearest FreqUpsample + Conv2d; depthwise + 1x1)
//   model/mtfaa.py:39-193                   ComplexConv2d, PhaseEncoder, TFCM_Block (1x1 -> depthwise dilated causal 3x3 -> 1x1)
// These take arbitrary kernel sizes, dilations, groups and pads, so they do not fit the 640-floats-per-frame
// frame-major kernels of the hot path (conv_mfma / conv.hip); they run here on the reference's own [B,C,H,W] layout
// (H,W = T,F for cust_conv; F,T for mtfaa) as direct VALU convolutions, coalesced along W.  All HBM-bound at these
// channel counts (8..64); one thread per output element, weights through the scalar/L1 path.
#include "common.h"
#include <stdlib.h>

namespace {

// STORAGE TYPE of the activation tensors (x, y, dy, dx): float, or _Float16 for BASELINE config 5 ("MTFAA ... fp16"):
// these blocks are HBM-bound, so halving the bytes of every tensor is what the fp16 mode buys; parameters, their
// gradients, accumulators and BatchNorm statistics stay f32 / f64 in either mode.
typedef _Float16 f16;

template <typename T> struct GConv {
    const T* x; const float* w; const float* bias; T* y;
    int B, Cin, Hin, Win, Cout, Hout, Wout;
    int KH, KW, sh, sw, dh, dw, pt, pl;
    int groups, up_w, transposed, act, accumulate;
    const float* slope;
    // optional (f16 MFMA-pointwise / LDS-depthwise kernels): the BatchNorm batch sums of the STORED outputs, bn_sums[rep][2 Cout] f64 with
    // rep = block % bn_nrep (sum, then sum of squares) -- what cruse_bn_nchw_stats would read back from y
    double* bn_sums; int bn_nrep;
    const T* res;                          // optional (f16 LDS-transposed pointwise kernel): y = T(T(conv) + res), the residual add of a block
    // optional (same two f16 kernels): y is the gradient wrt the OUTPUT of a BatchNorm (+ act) whose input is bb_x -- the epilogue accumulates that
    // BatchNorm's backward sums of the stored y, bb_r[rep][4][Cout] f64: sum d, sum d xhat, sum d z [z < 0] (PReLU slope), sum xhat with
    // d = y act'(z), z = gamma xhat + beta -- what cruse_bn_nchw_bwd's reduce pass would read y and bb_x again for
    const T* bb_x; const float* bb_mean; const float* bb_rstd; const float* bb_gamma; const float* bb_beta; const float* bb_slope;
    int bb_act; double* bb_r; int bb_nrep;
};

// transposed == 0 (nn.Conv2d, weight [Cout][Cin/g][KH][KW]; also the data gradient of a ConvTranspose2d):
//   y[b,co,ho,wo] = bias[co] + sum_{ci in group, kh, kw} w[co][ci_l][kh][kw] * X[b, ci, ho*sh - pt + kh*dh, wo*sw - pl + kw*dw]
//   X = x with zero padding; with up_w > 1, X[.., wi] = x[.., wi / up_w] (nearest FreqUpsample, cust_conv.py:177-184,
//   folded into the gather index: the upsampled tensor is never materialised) and Win is the size BEFORE upsampling.
// transposed == 1 (nn.ConvTranspose2d, weight [Cin][Cout/g][KH][KW]; also the data gradient of a Conv2d):
//   y[b,co,ho,wo] = bias[co] + sum_{ci, kh, kw : (ho + pt - kh*dh) % sh == 0, ...} w[ci][co_l][kh][kw] * x[b, ci, (ho + pt - kh*dh)/sh, (wo + pl - kw*dw)/sw]
template <typename T>
__global__ __launch_bounds__(256) void gconv_kernel(GConv<T> a) {
    const long long total = (long long)a.B * a.Cout * a.Hout * a.Wout;
    const int cin_g = a.Cin / a.groups, cout_g = a.Cout / a.groups;
    const int Wup = a.Win * a.up_w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int wo = (int)(i % a.Wout);
        long long r = i / a.Wout;
        const int ho = (int)(r % a.Hout); r /= a.Hout;
        const int co = (int)(r % a.Cout);
        const int b = (int)(r / a.Cout);
        const int g = co / cout_g, co_l = co - g * cout_g;
        float acc = a.bias ? a.bias[co] : 0.f;
        const T* xb = a.x + ((long long)b * a.Cin + (long long)g * cin_g) * a.Hin * a.Win;
        for (int kh = 0; kh < a.KH; ++kh) {
            int hi;
            if (!a.transposed) {
                hi = ho * a.sh - a.pt + kh * a.dh;
            } else {
                const int num = ho + a.pt - kh * a.dh;
                if (num < 0 || num % a.sh) continue;
                hi = num / a.sh;
            }
            if (hi < 0 || hi >= a.Hin) continue;
            for (int kw = 0; kw < a.KW; ++kw) {
                int wi;
                if (!a.transposed) {
                    wi = wo * a.sw - a.pl + kw * a.dw;
                    if (wi < 0 || wi >= Wup) continue;
                    wi /= a.up_w;
                } else {
                    const int num = wo + a.pl - kw * a.dw;
                    if (num < 0 || num % a.sw) continue;
                    wi = num / a.sw;
                    if (wi >= a.Win) continue;
                }
                const T* xp = xb + (long long)hi * a.Win + wi;
                if (!a.transposed) {
                    const float* wp = a.w + (((long long)co * cin_g) * a.KH + kh) * a.KW + kw;
                    for (int ci = 0; ci < cin_g; ++ci)
                        acc += wp[(long long)ci * a.KH * a.KW] * (float)xp[(long long)ci * a.Hin * a.Win];
                } else {
                    const float* wp = a.w + ((((long long)g * cin_g) * cout_g + co_l) * a.KH + kh) * a.KW + kw;
                    for (int ci = 0; ci < cin_g; ++ci)
                        acc += wp[(long long)ci * cout_g * a.KH * a.KW] * (float)xp[(long long)ci * a.Hin * a.Win];
                }
            }
        }
        if (a.act == 1) acc = fmaxf(acc, 0.f);
        else if (a.act == 2) acc = acc >= 0.f ? acc : a.slope[co] * acc;
        if (a.accumulate) a.y[i] = (T)((float)a.y[i] + acc); else a.y[i] = (T)acc;
    }
}

// 1x1, stride 1, groups 1 (the pointwise convolutions of TFCM_Block / the depthwise-separable blocks, either form): one
// thread per POSITION computes all Cout outputs from Cin coalesced loads -- the general kernel above issues Cin loads per
// OUTPUT (Cout x more).  Weights sit in LDS as [Cout][Cin] (transposed == 1: read from the [Cin][Cout] tensor).
template <typename T, int MAXCO>
__global__ __launch_bounds__(256) void gconv_pointwise_kernel(GConv<T> a) {
    extern __shared__ float wl[];                    // [Cout][Cin]
    for (int i = threadIdx.x; i < a.Cout * a.Cin; i += 256) {
        const int co = i / a.Cin, ci = i - co * a.Cin;
        wl[i] = a.transposed ? a.w[(long long)ci * a.Cout + co] : a.w[i];
    }
    __syncthreads();
    const long long hw = (long long)a.Hin * a.Win, total = (long long)a.B * hw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / hw, p = i - b * hw;
        const T* xp = a.x + b * a.Cin * hw + p;
        float acc[MAXCO];
#pragma unroll
        for (int co = 0; co < MAXCO; ++co) acc[co] = (co < a.Cout && a.bias) ? a.bias[co] : 0.f;
        for (int ci = 0; ci < a.Cin; ++ci) {
            const float v = (float)xp[(long long)ci * hw];
#pragma unroll
            for (int co = 0; co < MAXCO; ++co)
                if (co < a.Cout) acc[co] += wl[co * a.Cin + ci] * v;
        }
        T* yp = a.y + b * a.Cout * hw + p;
#pragma unroll
        for (int co = 0; co < MAXCO; ++co) {
            if (co < a.Cout) {
                float v = acc[co];
                if (a.act == 1) v = fmaxf(v, 0.f);
                else if (a.act == 2) v = v >= 0.f ? v : a.slope[co] * v;
                if (a.accumulate) yp[(long long)co * hw] = (T)((float)yp[(long long)co * hw] + v); else yp[(long long)co * hw] = (T)v;
            }
        }
    }
}

// The same pointwise convolution on the MATRIX CORES for f16 storage (BASELINE config 5; model/mtfaa.py:170-183 -- the two
// 1x1 convolutions of every TFCM_Block -- and the separable blocks of cust_conv.py:56-57,111-112), both forms:
//   D[co][p] = sum_ci W[co][ci] x[b][ci][p]   as   v_mfma_f32_16x16x32_f16, M = Cout (MT tiles), K = Cin (KS steps), N = positions.
// The weights live in registers as A fragments for the life of the block.  NCHW keeps p contiguous and ci strided, so the B
// fragment (lane = position l & 15, eight consecutive ci from (l >> 4) * 8) is gathered straight from global memory with
// 2-byte loads: the 16 lanes of a row read one 32-byte run, four tiles are in flight per wave, and a whole input line is
// consumed by four neighbouring tiles of the same wave (L1).  The VALU form above issues Cin * Cout FMAs and as many LDS
// weight reads per position (576 + 576 for the 24-channel TFCM layers); this one issues KS * 8 loads + MT * KS MFMAs per 16.
// (Measured alternative, round 4: a register-blocked f32 FMA form -- 4 positions x all channels per thread, 8-byte row loads -- is
// scheduled by the compiler into SGPR / VGPR spills whichever way the C x C weights are supplied and ran at 47 us vs 30 us here.)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
// eight independently loaded f16 (as 16-bit integers, one 32-bit register each so that no load waits for another) -> one fragment
__device__ __forceinline__ f16x8 pack_f16x8(const unsigned (&h)[8]) {
    u32x4 u = {h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
    return __builtin_bit_cast(f16x8, u);
}
__device__ __forceinline__ unsigned ld_u16(const f16* p) { return (unsigned)*reinterpret_cast<const unsigned short*>(p); }

template <int MT, int KS>
__global__ __launch_bounds__(256) void gconv_pointwise_mfma_f16_kernel(GConv<f16> a) {
    constexpr int TPI = 4;                                    // position tiles in flight per wave
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = lane & 15, kg = lane >> 4;
    const long long hw = (long long)a.Hin * a.Win;
    const int tiles_img = (int)((hw + 15) >> 4);
    const long long ntile = (long long)a.B * tiles_img;
    // resident weight fragments (f32 master weights rounded to f16 here): A[mt][ks][e] = W[co = mt*16 + m][ci = ks*32 + kg*8 + e]
    f16x8 wf[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int co = mt * 16 + m, ci = ks * 32 + kg * 8 + e;
                float w = 0.f;
                if (co < a.Cout && ci < a.Cin) w = a.transposed ? a.w[(long long)ci * a.Cout + co] : a.w[(long long)co * a.Cin + ci];
                wf[mt][ks][e] = (f16)w;
            }
    float bs[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = mt * 16 + kg * 4 + j;
            bs[mt][j] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
        }
    float sl[MT][4];                                          // PReLU slopes of the lane's output channels
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) sl[mt][j] = a.act == 2 ? a.slope[min(mt * 16 + kg * 4 + j, a.Cout - 1)] : 0.f;
    const long long wstride = (long long)gridDim.x * 4 * TPI;
    for (long long t0 = ((long long)blockIdx.x * 4 + wv) * TPI; t0 < ntile; t0 += wstride) {
        f16x8 fb[TPI][KS];
#pragma unroll
        for (int q = 0; q < TPI; ++q) {
            const long long t = t0 + q;
            const int b = (int)(t / tiles_img);
            const long long pp = (long long)(t - (long long)b * tiles_img) * 16 + m;
            const int ok = (t < ntile ? 1 : 0) & (pp < hw ? 1 : 0);
            // (unconditional loads from clamped addresses, masked afterwards: a branch per element serialises the round trips)
            const f16* xp = a.x + (long long)(t < ntile ? b : 0) * a.Cin * hw + (ok ? pp : 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                unsigned hv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ci = ks * 32 + kg * 8 + e;
                    const unsigned v = ld_u16(xp + (long long)min(ci, a.Cin - 1) * hw);
                    hv[e] = (ok & (ci < a.Cin ? 1 : 0)) ? v : 0u;
                }
                fb[q][ks] = pack_f16x8(hv);
            }
        }
#pragma unroll
        for (int q = 0; q < TPI; ++q) {
            const long long t = t0 + q;
            const int b = (int)(t / tiles_img);
            const long long pp = (long long)(t - (long long)b * tiles_img) * 16 + m;
            const bool ok = t < ntile && pp < hw;
            f16* yp = a.y + (long long)b * a.Cout * hw + pp;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 acc = {bs[mt][0], bs[mt][1], bs[mt][2], bs[mt][3]};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mt][ks], fb[q][ks], acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int co = mt * 16 + kg * 4 + j;
                    float v = acc[j];
                    v = a.act == 1 ? fmaxf(v, 0.f) : (a.act == 2 ? (v >= 0.f ? v : sl[mt][j] * v) : v);
                    if ((ok ? 1 : 0) & (co < a.Cout ? 1 : 0)) {
                        if (a.accumulate) v += (float)yp[(long long)co * hw];
                        yp[(long long)co * hw] = (f16)v;
                    }
                }
            }
        }
    }
}

// The same product with the operand TRANSPOSED THROUGH LDS (round 5).  The gather above issues eight 2-byte loads per fragment and eight
// 2-byte stores per tile and lane -- 16 memory instructions of 128 bytes per 16 positions: 30 us for 50 MB at the config-5 shape.  Here a
// wave moves 64 positions of all channels per step with 16-byte accesses: rows [ci][64 positions] are copied into a wave-private LDS image,
// the MFMA operand of a 16-position tile (lane = position, eight consecutive ci) is two TRANSPOSING reads (ds_read_b64_tr_b16: a 16-lane
// group reads a [4 ci][16 positions] block, lane c gets the four ci of position c); the product runs with the roles swapped
// (D^T[p][co] = x^T W^T: a lane then owns FOUR CONSECUTIVE positions of one output channel = one 8-byte LDS write), and the output rows
// [co][64 positions] leave through 16-byte stores.  No block barrier: the four waves of a block share nothing.
template <int MT, int KS, int EPI>      // EPI: 0 plain, 1 BatchNorm batch sums of the output (a.bn_sums), 2 BatchNorm-backward sums (a.bb_r)
__global__ __launch_bounds__(256) void gconv_pointwise_tr_f16_kernel(GConv<f16> a) {
    constexpr int ROWB = 144;                                 // bytes per LDS row: 64 positions + 16 (rows 4 apart land on different banks)
    constexpr int KR = KS * 32, MR = MT * 16;
    typedef __attribute__((ext_vector_type(4))) short s16x4_;
    typedef __attribute__((address_space(3))) s16x4_ lds_s16x4;
    extern __shared__ __align__(16) unsigned char dsm_raw[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = lane & 15, kg = lane >> 4;
    unsigned char* in_img = dsm_raw + (size_t)wv * (KR + MR) * ROWB;
    unsigned char* out_img = in_img + KR * ROWB;
    const long long hw = (long long)a.Hin * a.Win;
    const int nch = (int)((hw + 63) >> 6);
    const long long total = (long long)a.B * nch;
    // resident weight fragments, now the B operand: B[mt][ks][e] = W[co = mt*16 + m][ci = ks*32 + kg*8 + e]
    f16x8 wf[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int co = mt * 16 + m, ci = ks * 32 + kg * 8 + e;
                float w = 0.f;
                if (co < a.Cout && ci < a.Cin) w = a.transposed ? a.w[(long long)ci * a.Cout + co] : a.w[(long long)co * a.Cin + ci];
                wf[mt][ks][e] = (f16)w;
            }
    float bs[MT], sl[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int co = mt * 16 + m;
        bs[mt] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
        sl[mt] = a.act == 2 ? a.slope[min(co, a.Cout - 1)] : 0.f;
    }
    const int rr = lane >> 3, ck = lane & 7;                   // staging: row rr (+ 8 per pass), 16-byte chunk ck of the 64 positions
    const int tr_r = m >> 2, tr_s = m & 3;                     // transposing read: this lane addresses row tr_r, positions 4 tr_s .. + 4 of the block
    // the rows of a chunk, loaded one chunk AHEAD (a wave walks several chunks: the next one's loads fly under this one's LDS / MFMA / store phases)
    f16x8 pre[KR / 8];
    auto fetch = [&](long long ch) {
        const int b = (int)(ch / nch);
        const long long pos = (ch - (long long)b * nch) * 64 + ck * 8;
        const int valid = (int)min<long long>(max<long long>(hw - pos, 0), 8);
        const f16* xb = a.x + (long long)b * a.Cin * hw + pos;
#pragma unroll
        for (int it = 0; it < KR / 8; ++it) {
            const int ci = it * 8 + rr;
            f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ci < a.Cin && valid > 0) {
                const f16* q = xb + (long long)ci * hw;
                if (valid == 8) __builtin_memcpy(&v, q, 16);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = e < valid ? q[e] : (f16)0.f;
                }
            }
            pre[it] = v;
        }
    };
    float st1[MR / 8], st2[MR / 8];                            // BatchNorm sums of this lane's output rows (co = it * 8 + rr)
    float bq[MR / 8][4];                                       // ... / BatchNorm-backward sums (a.bb_r)
#pragma unroll
    for (int it = 0; it < MR / 8; ++it) { st1[it] = 0.f; st2[it] = 0.f; bq[it][0] = bq[it][1] = bq[it][2] = bq[it][3] = 0.f; }
    const long long ch0 = (long long)blockIdx.x * 4 + wv, chs = (long long)gridDim.x * 4;
    if (ch0 < total) fetch(ch0);
    for (long long ch = ch0; ch < total; ch += chs) {
        const int b = (int)(ch / nch);
        const long long p0 = (ch - (long long)b * nch) * 64;
        const long long pos = p0 + ck * 8;
        const int valid = (int)min<long long>(max<long long>(hw - pos, 0), 8);
        // ---- rows [ci][64 positions] -> LDS
#pragma unroll
        for (int it = 0; it < KR / 8; ++it) *reinterpret_cast<f16x8*>(in_img + (it * 8 + rr) * ROWB + ck * 16) = pre[it];
        if (ch + chs < total) fetch(ch + chs);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- four tiles of 16 positions
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f16x8 fa[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const unsigned char* bp = in_img + (ks * 32 + kg * 8 + tr_r) * ROWB + (q * 16 + 4 * tr_s) * 2;
                const s16x4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)bp);
                const s16x4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(bp + 4 * ROWB));
                typedef __attribute__((ext_vector_type(8))) short s16x8_;
                const s16x8_ both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                fa[ks] = __builtin_bit_cast(f16x8, both);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 acc = {bs[mt], bs[mt], bs[mt], bs[mt]};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ks], wf[mt][ks], acc, 0, 0, 0);
                typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_;
                f16x4_ o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[j];                           // D^T[position kg*4 + j][co = mt*16 + m]
                    v = a.act == 1 ? fmaxf(v, 0.f) : (a.act == 2 ? (v >= 0.f ? v : sl[mt] * v) : v);
                    o[j] = (f16)v;
                }
                *reinterpret_cast<f16x4_*>(out_img + (mt * 16 + m) * ROWB + (q * 16 + kg * 4) * 2) = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- rows [co][64 positions] -> global
        f16* yb = a.y + (long long)b * a.Cout * hw + pos;
#pragma unroll
        for (int it = 0; it < MR / 8; ++it) {
            const int co = it * 8 + rr;
            if (co < a.Cout && valid > 0) {
                f16x8 v = *reinterpret_cast<const f16x8*>(out_img + co * ROWB + ck * 16);
                if (a.res != nullptr) {                        // (two roundings, as the convolution followed by cruse_add_nchw)
                    const f16* rq = a.res + (long long)b * a.Cout * hw + pos + (long long)co * hw;
                    f16x8 r8 = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (valid == 8) __builtin_memcpy(&r8, rq, 16);
                    else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (e < valid) r8[e] = rq[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] + (float)r8[e]);
                }
                if constexpr (EPI == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e < valid) { const float u = (float)v[e]; st1[it] += u; st2[it] += u * u; }
                }
                if constexpr (EPI == 2) {
                    const f16* xq = a.bb_x + (long long)b * a.Cout * hw + pos + (long long)co * hw;
                    f16x8 x8 = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (valid == 8) __builtin_memcpy(&x8, xq, 16);
                    else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (e < valid) x8[e] = xq[e];
                    }
                    const float m_ = a.bb_mean[co], rs_ = a.bb_rstd[co], ga_ = a.bb_gamma[co], be_ = a.bb_beta[co];
                    const float sl_ = a.bb_act == 2 ? a.bb_slope[co] : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (e < valid) {
                            const float xh = ((float)x8[e] - m_) * rs_;
                            const float z = xh * ga_ + be_;
                            float d = (float)v[e];
                            if (a.bb_act == 1) d = z > 0.f ? d : 0.f;
                            else if (a.bb_act == 2) { if (z < 0.f) { bq[it][2] += d * z; d *= sl_; } }
                            else if (a.bb_act == 3) { const float sg = 1.f / (1.f + expf(-z)); d *= sg * (1.f - sg); }
                            bq[it][0] += d; bq[it][1] += d * xh; bq[it][3] += xh;
                        }
                    }
                }
                f16* q = yb + (long long)co * hw;
                if (valid == 8) __builtin_memcpy(q, &v, 16);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e < valid) q[e] = v[e];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (EPI == 1) {                                // 8 lanes share a row; 4 waves share the block: one f64 atomic per channel, sum and block
        __shared__ float bred[4][2][MR];
#pragma unroll
        for (int it = 0; it < MR / 8; ++it) {
            float u = st1[it], u2 = st2[it];
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) { u += __shfl_xor(u, o, 64); u2 += __shfl_xor(u2, o, 64); }
            if (ck == 0) { bred[wv][0][it * 8 + rr] = u; bred[wv][1][it * 8 + rr] = u2; }
        }
        __syncthreads();
        if (threadIdx.x < 2 * MR) {
            const int which = threadIdx.x / MR, co = threadIdx.x - which * MR;
            if (co < a.Cout) {
                const double t = (double)bred[0][which][co] + (double)bred[1][which][co] + (double)bred[2][which][co] + (double)bred[3][which][co];
                atomicAdd(a.bn_sums + (size_t)(blockIdx.x % a.bn_nrep) * 2 * a.Cout + which * a.Cout + co, t);
            }
        }
    }
    if constexpr (EPI == 2) {
        __shared__ float bred4[4][4][MR];
#pragma unroll
        for (int it = 0; it < MR / 8; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float u = bq[it][k];
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) u += __shfl_xor(u, o, 64);
                if (ck == 0) bred4[wv][k][it * 8 + rr] = u;
            }
        __syncthreads();
        for (int i = threadIdx.x; i < 4 * MR; i += 256) {
            const int which = i / MR, co = i - which * MR;
            if (co < a.Cout) {
                const double t = (double)bred4[0][which][co] + (double)bred4[1][which][co] + (double)bred4[2][which][co] + (double)bred4[3][which][co];
                atomicAdd(a.bb_r + (size_t)(blockIdx.x % a.bb_nrep) * 4 * a.Cout + which * a.Cout + co, t);
            }
        }
    }
}

// DEPTHWISE convolution (groups == Cin == Cout, stride 1), either form: TFCM_Block's dilated causal 3x3 (mtfaa.py:174-176) and
// its data gradient, the depthwise halves of the separable cust_conv blocks.  The general kernel above spends its time on
// per-element 64-bit index arithmetic and branches (0.3 TB/s on [8,24,161,401]); here a block is one output row chunk of one
// (image, channel) plane -- no division in the loop, the KH*KW weights of the channel in registers, taps read coalesced along W.
struct TapTab { int dh[9], dw[9], n; };        // input offset of tap k relative to the output position (either form)

template <typename T>
__global__ __launch_bounds__(256) void gconv_depthwise_kernel(GConv<T> a, TapTab tt) {
    // A block is DW_U * 256 consecutive outputs of one (image, channel) plane.  Every tap is an UNCONDITIONAL load from a clamped
    // address, masked afterwards: with a branch per tap the nine loads of an output were nine serialised round trips (one output
    // row per block, 480 GB/s on [8,24,161,401] f16); now DW_U * 9 loads are in flight per thread.
    constexpr int DW_U = 8;
    const int plane = blockIdx.y, c = plane % a.Cout;
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = k < tt.n ? a.w[c * tt.n + k] : 0.f;
    const float bias = a.bias ? a.bias[c] : 0.f;
    const float slope = a.act == 2 ? a.slope[c] : 0.f;
    const T* xp = a.x + (long long)plane * a.Hin * a.Win;
    T* yp = a.y + (long long)plane * a.Hout * a.Wout;
    const int hwo = a.Hout * a.Wout;
    const int e = blockIdx.x * (DW_U * 256) + threadIdx.x;
    // (straight-line code: positions first, then all loads, then the arithmetic -- no && / while between them, the taps beyond
    // tt.n carry weight 0 and offset 0)
    int hos[DW_U], wos[DW_U];
#pragma unroll
    for (int j = 0; j < DW_U; ++j) {
        const unsigned ej = (unsigned)(e + j * 256);
        hos[j] = (int)(ej / (unsigned)a.Wout);
        wos[j] = (int)(ej - (unsigned)hos[j] * (unsigned)a.Wout);
    }
    T xv[DW_U][9];
    unsigned okm[DW_U];
#pragma unroll
    for (int j = 0; j < DW_U; ++j) {
        const int in = (e + j * 256 < hwo) ? 1 : 0;
        okm[j] = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int hi = hos[j] + tt.dh[k], wi = wos[j] + tt.dw[k];
            const int ok = in & (hi >= 0 ? 1 : 0) & (hi < a.Hin ? 1 : 0) & (wi >= 0 ? 1 : 0) & (wi < a.Win ? 1 : 0);
            okm[j] |= (unsigned)ok << k;
            xv[j][k] = xp[ok * (hi * a.Win + wi)];
        }
    }
#pragma unroll
    for (int j = 0; j < DW_U; ++j) {
        float acc = bias;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc += ((okm[j] >> k) & 1) ? w[k] * (float)xv[j][k] : 0.f;
        if (a.act == 1) acc = fmaxf(acc, 0.f);
        else if (a.act == 2) acc = acc >= 0.f ? acc : slope * acc;
        const int ej = e + j * 256;
        if (ej < hwo) {
            if (a.accumulate) acc += (float)yp[ej];
            yp[ej] = (T)acc;
        }
    }
}

// eight consecutive f16 of one plane row <-> floats: one (unaligned) 16-byte access for a whole group, element accesses for a tail
__device__ __forceinline__ void ld8_f16(const f16* p, int valid, float (&v)[8]) {
    if (valid >= 8) {
        f16x8 t;
        __builtin_memcpy(&t, p, 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = e < valid ? (float)p[e] : 0.f;
    }
}
__device__ __forceinline__ void st8_f16(f16* p, int valid, const float (&v)[8]) {
    if (valid >= 8) {
        f16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (f16)v[e];
        __builtin_memcpy(p, &t, 16);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e < valid) p[e] = (f16)v[e];
    }
}

// f16 storage, either form: the input rows of a band of RB output rows are staged in LDS as a ZERO-PADDED image (PADL / PADR halo
// columns, zero rows above / below the plane), so the compute loop has no bounds logic at all: a thread owns 8 consecutive outputs
// of a row and every tap is one (unaligned) 16-byte LDS read + 8 conversions + 8 FMAs.  The element-per-thread forms above spend
// ~160 VALU instructions per output on index arithmetic and masks -- at 4 cycles per wave64 instruction that, not memory, was
// their bound (62-84 us on [8,24,161,401]).  Rows are copied with unaligned 16-byte global loads.
struct DwGeom { int RB, rows, Ws, PADL, dh_lo, gpr; };        // staged rows, LDS row stride (elements), groups per output row

__device__ __forceinline__ void dw_stage_f16(const f16* xp, int Hin, int Win, int row_lo, const DwGeom& g, f16* sx) {
    // LDS row i holds input row row_lo + i: [PADL zeros][Win elements][zeros up to Ws]
    const int cpr = g.Ws / 8;                                          // 16-byte chunks per LDS row
    for (int q = threadIdx.x; q < g.rows * cpr; q += 256) {
        const int i = q / cpr, ck = q - i * cpr;
        const int hi = row_lo + i, w0 = ck * 8 - g.PADL;               // first input column of the chunk
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hi >= 0 && hi < Hin && w0 + 8 > 0 && w0 < Win) {
            const f16* src = xp + (long long)hi * Win + w0;
            if (w0 >= 0 && w0 + 8 <= Win) __builtin_memcpy(&v, src, 16);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (w0 + e >= 0 && w0 + e < Win) v[e] = src[e];
            }
        }
        *reinterpret_cast<f16x8*>(sx + (size_t)i * g.Ws + ck * 8) = v;
    }
}

__global__ __launch_bounds__(256) void gconv_depthwise_f16_kernel(GConv<f16> a, TapTab tt, DwGeom g) {
    extern __shared__ __align__(16) unsigned char dsm_raw[];
    f16* sx = reinterpret_cast<f16*>(dsm_raw);
    const int plane = blockIdx.y, c = plane % a.Cout;
    const int r0 = blockIdx.x * g.RB, nr = min(a.Hout, r0 + g.RB) - r0;
    dw_stage_f16(a.x + (long long)plane * a.Hin * a.Win, a.Hin, a.Win, r0 + g.dh_lo, g, sx);
    float w[9];
    int toff[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        w[k] = k < tt.n ? a.w[c * tt.n + k] : 0.f;
        toff[k] = (tt.dh[k] - g.dh_lo) * g.Ws + g.PADL + tt.dw[k];       // (taps beyond tt.n: offset of tap 0's row, weight 0)
    }
    const float bias = a.bias ? a.bias[c] : 0.f;
    const float slope = a.act == 2 ? a.slope[c] : 0.f;
    f16* yp = a.y + ((long long)plane * a.Hout + r0) * a.Wout;
    float bn1 = 0.f, bn2 = 0.f;                                        // BatchNorm sums of the stored outputs (a.bn_sums)
    float bq0 = 0.f, bq1 = 0.f, bq2 = 0.f, bq3 = 0.f;                  // BatchNorm-backward sums (a.bb_r)
    const float bm_ = a.bb_r ? a.bb_mean[c] : 0.f, brs_ = a.bb_r ? a.bb_rstd[c] : 1.f, bga_ = a.bb_r ? a.bb_gamma[c] : 1.f, bbe_ = a.bb_r ? a.bb_beta[c] : 0.f;
    const float bsl_ = (a.bb_r && a.bb_act == 2) ? a.bb_slope[c] : 0.f;
    __syncthreads();
    for (int q = threadIdx.x; q < nr * g.gpr; q += 256) {
        const int hr = q / g.gpr, wo = (q - hr * g.gpr) * 8;
        const f16* base = sx + hr * g.Ws + wo;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bias;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            f16x8 v;                                                   // (taps beyond tt.n: weight 0, a valid offset)
            __builtin_memcpy(&v, base + toff[k], 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += w[k] * (float)v[e];
        }
        f16* dst = yp + hr * a.Wout + wo;
        const int valid = a.Wout - wo;
        if (a.accumulate) {
            float old[8];
            ld8_f16(dst, valid, old);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += old[e];              // (accumulate comes with act 0: the data-gradient form)
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (a.act == 1) acc[e] = fmaxf(acc[e], 0.f);
            else if (a.act == 2) acc[e] = acc[e] >= 0.f ? acc[e] : slope * acc[e];
        }
        st8_f16(dst, valid, acc);
        if (a.bn_sums != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e < valid) { const float u = (float)(f16)acc[e]; bn1 += u; bn2 += u * u; }
        }
        if (a.bb_r != nullptr) {
            float xv[8];
            ld8_f16(a.bb_x + ((long long)plane * a.Hout + r0 + hr) * a.Wout + wo, valid, xv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e < valid) {
                    const float xh = (xv[e] - bm_) * brs_;
                    const float z = xh * bga_ + bbe_;
                    float d = (float)(f16)acc[e];
                    if (a.bb_act == 1) d = z > 0.f ? d : 0.f;
                    else if (a.bb_act == 2) { if (z < 0.f) { bq2 += d * z; d *= bsl_; } }
                    else if (a.bb_act == 3) { const float sg = 1.f / (1.f + expf(-z)); d *= sg * (1.f - sg); }
                    bq0 += d; bq1 += d * xh; bq3 += xh;
                }
            }
        }
    }
    if (a.bb_r != nullptr) {
        __shared__ float bred4[4][4];
        bq0 = wave_sum(bq0); bq1 = wave_sum(bq1); bq2 = wave_sum(bq2); bq3 = wave_sum(bq3);
        if ((threadIdx.x & 63) == 0) { const int w_ = threadIdx.x >> 6; bred4[0][w_] = bq0; bred4[1][w_] = bq1; bred4[2][w_] = bq2; bred4[3][w_] = bq3; }
        __syncthreads();
        if (threadIdx.x < 4) {
            const double t = (double)bred4[threadIdx.x][0] + (double)bred4[threadIdx.x][1] + (double)bred4[threadIdx.x][2] + (double)bred4[threadIdx.x][3];
            atomicAdd(a.bb_r + (size_t)((blockIdx.x + blockIdx.y) % a.bb_nrep) * 4 * a.Cout + threadIdx.x * a.Cout + c, t);
        }
    }
    if (a.bn_sums != nullptr) {                                        // one (image, channel) band per block: two f64 atomics
        __shared__ float bred[2][4];
        bn1 = wave_sum(bn1); bn2 = wave_sum(bn2);
        if ((threadIdx.x & 63) == 0) { bred[0][threadIdx.x >> 6] = bn1; bred[1][threadIdx.x >> 6] = bn2; }
        __syncthreads();
        if (threadIdx.x < 2) {
            const double t = (double)bred[threadIdx.x][0] + (double)bred[threadIdx.x][1] + (double)bred[threadIdx.x][2] + (double)bred[threadIdx.x][3];
            atomicAdd(a.bn_sums + (size_t)((blockIdx.x + blockIdx.y) % a.bn_nrep) * 2 * a.Cout + threadIdx.x * a.Cout + c, t);
        }
    }
}

// Weight gradient of both forms as ONE contraction:
//   dw[ca][cb_l][kh][kw] += sum_{n,h,w} S[n,ca,h,w] * Bg[n, g*CBg + cb_l, h*sh - pt + kh*dh, (w*sw - pl + kw*dw) / up_w]
// Conv2d:          S = dy (ca = co, HxW = output size), Bg = x;   ConvTranspose2d: S = x (ca = ci), Bg = dy.
template <typename T> struct GWgrad {
    const T* S; const T* Bg; float* dw;
    int N, CA, HS, WS, CB, HB, WB;
    int KH, KW, sh, sw, dh, dw_, pt, pl, groups, up_w;
    float* db;                             // optional: db[ca] += sum_{n,h,w} S[n,ca,h,w] (the bias gradient of a Conv2d: S = dy)
};
template <typename T>
__global__ __launch_bounds__(256) void gconv_wgrad_kernel(GWgrad<T> a) {
    __shared__ float red[4];
    const int cb_g = a.CB / a.groups, ca_g = a.CA / a.groups;
    int wi_ = blockIdx.x;                                   // weight element
    const int kw = wi_ % a.KW; wi_ /= a.KW;
    const int kh = wi_ % a.KH; wi_ /= a.KH;
    const int cb_l = wi_ % cb_g;
    const int ca = wi_ / cb_g;
    const int g = ca / ca_g;
    const int cb = g * cb_g + cb_l;
    const int Wup = a.WB * a.up_w;
    const int hw = a.HS * a.WS;
    float acc = 0.f;
    for (int n = blockIdx.y; n < a.N; n += gridDim.y) {
        const T* sp = a.S + ((long long)n * a.CA + ca) * hw;
        const T* bp = a.Bg + ((long long)n * a.CB + cb) * a.HB * a.WB;
        for (int i = threadIdx.x; i < hw; i += 256) {
            const int h = i / a.WS, w = i - h * a.WS;
            const int hb = h * a.sh - a.pt + kh * a.dh;
            int wb = w * a.sw - a.pl + kw * a.dw_;
            if (hb < 0 || hb >= a.HB || wb < 0 || wb >= Wup) continue;
            wb /= a.up_w;
            acc += (float)sp[i] * (float)bp[(long long)hb * a.WB + wb];
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&a.dw[blockIdx.x], red[0] + red[1] + red[2] + red[3]);
}

// Row-tiled form of the same contraction for small channel counts: a block owns one (n, h) row of S, stages it ([CA][WS]) and,
// for each kh, the matching row of Bg ([CB][WB]) in LDS ONCE, and every thread accumulates its weight elements (ca, cb_l, kw)
// over the row from LDS; one atomic per weight element and block.  The per-weight kernel above streams one plane of S and one of
// Bg per BLOCK -- CA*CB/groups*KH*KW times the tensors' bytes (2.3 GB for a 24 x 24 pointwise layer on [8,24,161,401]).
template <typename T>
__global__ __launch_bounds__(256) void gconv_wgrad_rows_kernel(GWgrad<T> a) {
    extern __shared__ float sm[];
    float* Ss = sm;                                   // [CA][WS]
    float* Bs = sm + (size_t)a.CA * a.WS;             // [CB][WB]
    const int n = blockIdx.x / a.HS, h = blockIdx.x - n * a.HS;
    const int cb_g = a.CB / a.groups, ca_g = a.CA / a.groups;
    const int nwk = a.CA * cb_g * a.KW;               // weight elements per kh
    const int Wup = a.WB * a.up_w;
    for (int i = threadIdx.x; i < a.CA * a.WS; i += 256) {
        const int ca = i / a.WS, w = i - ca * a.WS;
        Ss[i] = (float)a.S[(((long long)n * a.CA + ca) * a.HS + h) * a.WS + w];
    }
    for (int kh = 0; kh < a.KH; ++kh) {
        const int hb = h * a.sh - a.pt + kh * a.dh;
        __syncthreads();                              // S staged / previous kh's reads done
        if (hb < 0 || hb >= a.HB) continue;           // block-uniform
        for (int i = threadIdx.x; i < a.CB * a.WB; i += 256) {
            const int cb = i / a.WB, w = i - cb * a.WB;
            Bs[i] = (float)a.Bg[(((long long)n * a.CB + cb) * a.HB + hb) * a.WB + w];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < nwk; e += 256) {
            const int kw = e % a.KW;
            const int r = e / a.KW;
            const int cb_l = r % cb_g, ca = r / cb_g;
            const int cb = (ca / ca_g) * cb_g + cb_l;
            const float* sp = Ss + (size_t)ca * a.WS;
            const float* bp = Bs + (size_t)cb * a.WB;
            const int off = kw * a.dw_ - a.pl;
            float acc = 0.f;
            if (a.up_w == 1) {
                // valid w range hoisted out of the loop: 0 <= w*sw + off < WB  (no per-element division or branch)
                int w_lo = off >= 0 ? 0 : (-off + a.sw - 1) / a.sw;
                int w_hi = (Wup - 1 - off) >= 0 ? (Wup - 1 - off) / a.sw + 1 : 0;
                if (w_hi > a.WS) w_hi = a.WS;
                const float* bq = bp + off;
                int w = w_lo;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                for (; w + 3 < w_hi; w += 4) {
                    a0 += sp[w] * bq[w * a.sw];
                    a1 += sp[w + 1] * bq[(w + 1) * a.sw];
                    a2 += sp[w + 2] * bq[(w + 2) * a.sw];
                    a3 += sp[w + 3] * bq[(w + 3) * a.sw];
                }
                for (; w < w_hi; ++w) a0 += sp[w] * bq[w * a.sw];
                acc = (a0 + a1) + (a2 + a3);
            } else {
                for (int w = 0; w < a.WS; ++w) {
                    const int wb = w * a.sw + off;
                    if (wb >= 0 && wb < Wup) acc += sp[w] * bp[wb / a.up_w];
                }
            }
            atomicAdd(&a.dw[((long long)(ca * cb_g + cb_l) * a.KH + kh) * a.KW + kw], acc);
        }
    }
}

// Weight gradient of a POINTWISE convolution on the matrix cores (f16 storage): dW[ca][cb] += sum_{n,p} S[n,ca,p] Bg[n,cb,p]
// with K = positions -- the natural MFMA shape on NCHW: both fragments are 8 consecutive positions of one channel row.  A wave
// owns a run of positions of one image and keeps the whole [CA x CB] result (MT x NT tiles) in its accumulators; one atomic
// add per weight and wave at the end.  (Rows of an odd-sized plane are only 2-byte aligned: the 16-byte fragment loads are
// unaligned global loads.)
template <int MT, int NT>
__global__ __launch_bounds__(1024) void gconv_wgrad_pw_mfma_f16_kernel(GWgrad<f16> a, int run_dbg) {
    const int run = run_dbg & 0xffffff, dbg = run_dbg >> 24;
    extern __shared__ __align__(16) unsigned char dsm_raw[];
    float* red = reinterpret_cast<float*>(dsm_raw);                      // [16 waves][MT * NT * 256]: one slab per wave, no LDS atomics
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;            // 16 waves: enough loads in flight to cover HBM latency
    const int r = lane & 15, kg = lane >> 4;
    const long long hw = (long long)a.HS * a.WS;
    const int runs_img = (int)((hw + run - 1) / run);
    const int n = blockIdx.x / runs_img;
    const long long p0 = (long long)(blockIdx.x - n * runs_img) * run, p1 = min(hw, p0 + run);
    const f16* sp = a.S + (long long)n * a.CA * hw;
    const f16* bp = a.Bg + (long long)n * a.CB * hw;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // wave wv takes the 32-position chunks wv, wv + 16, ... of the block's run, two at a time
    for (long long p = p0 + wv * 32; p < p1; p += 2 * 16 * 32) {
        f16x8 fa[2][MT], fb[2][NT];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            // a fragment = 8 consecutive positions of one channel row = ONE 16-byte load (rows of an odd-sized plane are only 2-byte
            // aligned: global loads take unaligned addresses on this target, the compiler emits global_load_dwordx4 for the
            // memcpy); the run's last, partial group is gathered element by element
            const long long pk = p + u * 16 * 32 + kg * 8;
            const int last = (int)min<long long>(max<long long>(p1 - 1 - pk, -1), 7);      // last valid e (-1: none)
            const long long pc = min(pk, p1 - 1);
            const bool full = last == 7;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int ca = i * 16 + r;
                const f16* q = sp + (long long)min(ca, a.CA - 1) * hw + pc;
                f16x8 v;
                if (full) __builtin_memcpy(&v, q, 16);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = e <= last ? q[min(e, max(last, 0))] : (f16)0.f;
                }
                fa[u][i] = ca < a.CA ? v : zero8;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int cb = j * 16 + r;
                const f16* q = bp + (long long)min(cb, a.CB - 1) * hw + pc;
                f16x8 v;
                if (full) __builtin_memcpy(&v, q, 16);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = e <= last ? q[min(e, max(last, 0))] : (f16)0.f;
                }
                if (a.db != nullptr && cb == a.CB) {           // first padding column: the constant 1 -> its dW column is the bias gradient
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = e <= last ? (f16)1.f : (f16)0.f;
                    fb[u][j] = v;
                } else {
                    fb[u][j] = cb < a.CB ? v : zero8;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[u][i], fb[u][j], acc[i][j], 0, 0, 0);
    }
    // acc[i][j][q] = dW[ca = i*16 + kg*4 + q][cb = j*16 + r]: every wave stores its tile sums into its own LDS slab (LDS float
    // atomics from 16 waves cost 20-40 us per launch), a thread then adds the 16 slabs of its weight: one global atomic per
    // weight and block
    float* mine = red + wv * (MT * NT * 256);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) mine[((i * NT + j) * 16 + kg * 4 + q) * 16 + r] = acc[i][j][q];
    __syncthreads();
    for (int idx = threadIdx.x; idx < MT * NT * 256; idx += 1024) {
        const int t = idx >> 8, row = (idx >> 4) & 15, col = idx & 15;
        const int ca = (t / NT) * 16 + row, cb = (t % NT) * 16 + col;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += red[w * (MT * NT * 256) + idx];
        if (ca < a.CA && cb < a.CB && !(dbg & 1)) atomicAdd(&a.dw[(long long)ca * a.CB + cb], v);
        else if (ca < a.CA && cb == a.CB && a.db != nullptr) atomicAdd(&a.db[ca], v);
    }
}

// Weight gradient of a DEPTHWISE convolution: dW[c][kh][kw] += sum_{n,h,w} S[n,c,h,w] Bg[n,c, h*sh - pt + kh*dh, w*sw - pl + kw*dw].
// A block is a band of rows of one (image, channel) plane; a thread walks W and keeps the KH*KW partial sums in registers.
template <typename T>
__global__ __launch_bounds__(256) void gconv_wgrad_depthwise_kernel(GWgrad<T> a, int band, TapTab tt) {
    // band = S positions per thread (a block takes band * 256 consecutive positions of one plane); unconditional clamped tap
    // loads masked afterwards, as in gconv_depthwise_kernel
    __shared__ float red[4][9];
    const int plane = blockIdx.y, c = plane % a.CA;
    const T* sp = a.S + (long long)plane * a.HS * a.WS;
    const T* bp = a.Bg + (long long)plane * a.HB * a.WB;
    const int hws = a.HS * a.WS;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    // (straight-line groups of WG_U positions: positions, then all loads, then the arithmetic; taps beyond tt.n have offset 0 and
    // their sums are never written)
    constexpr int WG_U = 4;
    const int e0 = blockIdx.x * (band * 256) + threadIdx.x;
    for (int j0 = 0; j0 < band; j0 += WG_U) {
        T bv[WG_U][9];
        float sv[WG_U];
        unsigned okm[WG_U];
#pragma unroll
        for (int j = 0; j < WG_U; ++j) {
            const unsigned ej = (unsigned)(e0 + (j0 + j) * 256);
            const int in = ((int)ej < hws && j0 + j < band) ? 1 : 0;
            const int h = (int)(ej / (unsigned)a.WS), w = (int)(ej - (unsigned)h * (unsigned)a.WS);
            sv[j] = in ? (float)sp[in * (int)ej] : 0.f;
            okm[j] = 0;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int hb = h * a.sh + tt.dh[k], wb = w * a.sw + tt.dw[k];
                const int ok = in & (hb >= 0 ? 1 : 0) & (hb < a.HB ? 1 : 0) & (wb >= 0 ? 1 : 0) & (wb < a.WB ? 1 : 0);
                okm[j] |= (unsigned)ok << k;
                bv[j][k] = bp[ok * (hb * a.WB + wb)];
            }
        }
#pragma unroll
        for (int j = 0; j < WG_U; ++j)
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] += ((okm[j] >> k) & 1) ? sv[j] * (float)bv[j][k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float v = wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < tt.n) atomicAdd(&a.dw[c * tt.n + threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ... f16 storage, stride 1: the Bg rows of a band of S rows staged as the same zero-padded LDS image; a thread owns 8 consecutive
// S positions (one 16-byte global load) and every tap is one 16-byte LDS read + 8 conversions + 8 FMAs into that tap's sum.
__global__ __launch_bounds__(256) void gconv_wgrad_depthwise_f16_kernel(GWgrad<f16> a, TapTab tt, DwGeom g) {
    extern __shared__ __align__(16) unsigned char dsm_raw[];
    f16* sb = reinterpret_cast<f16*>(dsm_raw);
    __shared__ float red[4][9];
    const int plane = blockIdx.y, c = plane % a.CA;
    const int r0 = blockIdx.x * g.RB, nr = min(a.HS, r0 + g.RB) - r0;
    dw_stage_f16(a.Bg + (long long)plane * a.HB * a.WB, a.HB, a.WB, r0 + g.dh_lo, g, sb);
    int toff[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) toff[k] = (tt.dh[k] - g.dh_lo) * g.Ws + g.PADL + tt.dw[k];
    const f16* sp = a.S + ((long long)plane * a.HS + r0) * a.WS;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    __syncthreads();
    for (int q = threadIdx.x; q < nr * g.gpr; q += 256) {
        const int hr = q / g.gpr, w0 = (q - hr * g.gpr) * 8;
        float sv[8];
        ld8_f16(sp + hr * a.WS + w0, a.WS - w0, sv);                   // (zeros beyond the row's end)
        const f16* base = sb + hr * g.Ws + w0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            f16x8 v;                                                   // (taps beyond tt.n: a valid offset, the sum is never written)
            __builtin_memcpy(&v, base + toff[k], 16);
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) t += sv[e] * (float)v[e];
            acc[k] += t;
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float v = wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < tt.n) atomicAdd(&a.dw[c * tt.n + threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[c] += sum_{n,hw} x[n,c,hw]   (conv bias gradient)
template <typename T>
__global__ __launch_bounds__(256) void nchw_channel_sum_kernel(const T* x, int N, int C, int HW, float* out) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float acc = 0.f;
    // planes are split into gridDim.z chunks: a [8,24,161,401] tensor gives 24 x 8 planes only, far too few blocks to fill
    // the chip with one block per plane (these reductions ran at 0.4 TB/s)
    const int chunk = (HW + gridDim.z - 1) / gridDim.z, i0 = blockIdx.z * chunk, i1 = min(HW, i0 + chunk);
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const T* p = x + ((long long)n * C + c) * HW;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int i = i0 + threadIdx.x;
        for (; i + 768 < i1; i += 1024) { a0 += (float)p[i]; a1 += (float)p[i + 256]; a2 += (float)p[i + 512]; a3 += (float)p[i + 768]; }
        for (; i < i1; i += 256) a0 += (float)p[i];
        acc += (a0 + a1) + (a2 + a3);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&out[c], red[0] + red[1] + red[2] + red[3]);
}

// dx[.., w0] = sum_{j < up} dxu[.., w0*up + j]   (gradient of the nearest FreqUpsample)
template <typename T>
__global__ void downsum_w_kernel(const T* dxu, long long rows, int W, int up, T* dx) {
    const long long n = rows * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / W;
        const int w = (int)(i - r * W);
        float s = 0.f;
        for (int j = 0; j < up; ++j) s += (float)dxu[r * W * up + (long long)w * up + j];
        dx[i] = (T)s;
    }
}

// xu[.., w*up + j] = x[.., w]  (nearest FreqUpsample of the frame-major upsample decoder: one thread per 4 outputs when up == 2)
template <typename T>
__global__ void upsample_w_kernel(const T* x, long long rows, int W, int up, T* xu) {
    const long long n = rows * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const T v = x[i];
        for (int j = 0; j < up; ++j) xu[i * up + j] = v;
    }
}
__global__ void upsample_w2_f32_kernel(const float2* x, long long n2, float4* xu) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
        const float2 v = x[i];
        xu[i] = make_float4(v.x, v.x, v.y, v.y);
    }
}
// dx[.., w] = dxu[.., 2w] + dxu[.., 2w + 1] on whole float4 loads
__global__ void downsum_w2_f32_kernel(const float4* dxu, long long n2, float2* dx) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = dxu[i];
        dx[i] = make_float2(v.x + v.y, v.z + v.w);
    }
}

// ---- BatchNorm2d on NCHW (+ ReLU / PReLU) ---------------------------------------------------------
// sums[c] = sum x, sums[C + c] = sum x^2 in f64 (cruse_bn_finalize turns them into mean / rstd / running stats)
template <typename T>
__global__ __launch_bounds__(256) void bn_nchw_stats_kernel(const T* x, int N, int C, int HW, double* sums) {
    __shared__ double red[2][4];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    const int chunk = (HW + gridDim.z - 1) / gridDim.z, i0 = blockIdx.z * chunk, i1 = min(HW, i0 + chunk);
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const T* p = x + ((long long)n * C + c) * HW;
        float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
        int i = i0 + threadIdx.x;
        for (; i + 768 < i1; i += 1024) {
            const float v0 = (float)p[i], v1 = (float)p[i + 256], v2 = (float)p[i + 512], v3 = (float)p[i + 768];
            a1 += v0 + v1; b1 += v2 + v3; a2 += v0 * v0 + v1 * v1; b2 += v2 * v2 + v3 * v3;
        }
        for (; i < i1; i += 256) { const float v = (float)p[i]; a1 += v; a2 += v * v; }
        s1 += a1 + b1; s2 += a2 + b2;
    }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&sums[c], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(&sums[C + c], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// y = act(gamma * (x - mean) * rstd + beta); mean == NULL: no normalisation (plain activation of x); act 0 none, 1 ReLU,
// 2 PReLU(slope[c]), 3 sigmoid
template <typename T>
__global__ void bn_nchw_fwd_kernel(const T* x, const float* mean, const float* rstd, const float* gamma,
                                   const float* beta, const float* slope, int act, long long total, int C, int HW, T* y) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW) % C);
        float z = (float)x[i];
        if (mean) z = (z - mean[c]) * rstd[c] * gamma[c] + beta[c];
        if (act == 1) z = fmaxf(z, 0.f);
        else if (act == 2) z = z >= 0.f ? z : slope[c] * z;
        else if (act == 3) z = 1.f / (1.f + expf(-z));
        y[i] = (T)z;
    }
}

// per channel: r[c] = sum dz, r[C+c] = sum dz * xhat, r[2C+c] = sum dy * min(z, 0) (PReLU slope gradient); dz = dy through act
template <typename T>
__global__ __launch_bounds__(256) void bn_nchw_bwd_reduce_kernel(const T* dy, const T* x, const float* mean,
                                                                 const float* rstd, const float* gamma, const float* beta,
                                                                 const float* slope, int act, int N, int C, int HW,
                                                                 double* r) {
    __shared__ double red[3][4];
    const int c = blockIdx.x;
    const float m = mean ? mean[c] : 0.f, rs = mean ? rstd[c] : 1.f, ga = mean ? gamma[c] : 1.f, be = mean ? beta[c] : 0.f;
    const float sl = act == 2 ? slope[c] : 0.f;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const int chunk = (HW + gridDim.z - 1) / gridDim.z, i0 = blockIdx.z * chunk, i1 = min(HW, i0 + chunk);
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const long long o = ((long long)n * C + c) * HW;
        float a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
        for (int i = i0 + threadIdx.x; i < i1; i += 256) {
            const float xh = ((float)x[o + i] - m) * rs;
            const float z = xh * ga + be;
            float d = (float)dy[o + i];
            if (act == 1) d = z > 0.f ? d : 0.f;
            else if (act == 2) { if (z < 0.f) { a3 += d * z; d *= sl; } }
            else if (act == 3) { const float sg = 1.f / (1.f + expf(-z)); d *= sg * (1.f - sg); }
            a1 += d; a2 += d * xh;
        }
        s1 += a1; s2 += a2; s3 += a3;
    }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2); s3 = wave_sum_d(s3);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; red[2][threadIdx.x >> 6] = s3; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&r[c], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(&r[C + c], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        atomicAdd(&r[2 * C + c], red[2][0] + red[2][1] + red[2][2] + red[2][3]);
    }
}

// dx = gamma * rstd * (dz - [training] (mean(dz) + xhat * mean(dz * xhat)));  mean == NULL: dx = dz
template <typename T>
__global__ void bn_nchw_bwd_apply_kernel(const T* dy, const T* x, const float* mean, const float* rstd,
                                         const float* gamma, const float* beta, const float* slope, int act,
                                         const double* r, double inv_count, int training, long long total, int C, int HW,
                                         T* dx) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW) % C);
        const float m = mean ? mean[c] : 0.f, rs = mean ? rstd[c] : 1.f, ga = mean ? gamma[c] : 1.f, be = mean ? beta[c] : 0.f;
        const float xh = ((float)x[i] - m) * rs;
        const float z = xh * ga + be;
        float d = (float)dy[i];
        if (act == 1) d = z > 0.f ? d : 0.f;
        else if (act == 2) d = z >= 0.f ? d : slope[c] * d;
        else if (act == 3) { const float sg = 1.f / (1.f + expf(-z)); d *= sg * (1.f - sg); }
        if (mean) {
            if (training) d -= (float)(r[c] * inv_count) + xh * (float)(r[C + c] * inv_count);
            d *= ga * rs;
        }
        dx[i] = (T)d;
    }
}

// ---- f16 storage: the same four BatchNorm passes and the channel sum on 16-byte groups -----------------------------------------
// A plane (n, c) is HW contiguous f16, 2-byte aligned when HW is odd ([.., 161, 401]): a thread takes 8 consecutive elements of
// ONE plane with one unaligned 16-byte load (the element-per-thread kernels above also pay a 64-bit i / HW % C per element:
// 0.7-1.5 TB/s on 25 MB tensors); the plane's last, partial group is handled element by element.
__global__ __launch_bounds__(256) void bn_nchw_stats_f16v_kernel(const f16* x, int N, int C, int HW, double* sums) {
    __shared__ double red[2][4];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    const int chunk = ((HW + gridDim.z - 1) / gridDim.z + 7) & ~7, i0 = blockIdx.z * chunk, i1 = min(HW, i0 + chunk);
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const f16* p = x + ((long long)n * C + c) * HW;
        float a1 = 0.f, a2 = 0.f;
#pragma unroll 2
        for (int i = i0 + threadIdx.x * 8; i < i1; i += 2048) {
            float v[8];
            ld8_f16(p + i, i1 - i, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { a1 += v[e]; a2 += v[e] * v[e]; }
        }
        s1 += a1; s2 += a2;
    }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&sums[c], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(&sums[C + c], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

__global__ __launch_bounds__(256) void nchw_channel_sum_f16v_kernel(const f16* x, int N, int C, int HW, float* out) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float acc = 0.f;
    const int chunk = ((HW + gridDim.z - 1) / gridDim.z + 7) & ~7, i0 = blockIdx.z * chunk, i1 = min(HW, i0 + chunk);
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const f16* p = x + ((long long)n * C + c) * HW;
        float a0 = 0.f;
#pragma unroll 2
        for (int i = i0 + threadIdx.x * 8; i < i1; i += 2048) {
            float v[8];
            ld8_f16(p + i, i1 - i, v);
            a0 += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        acc += a0;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&out[c], red[0] + red[1] + red[2] + red[3]);
}

// grid (groups of a plane / 256, N * C)
__global__ __launch_bounds__(256) void bn_nchw_fwd_f16v_kernel(const f16* x, const float* mean, const float* rstd, const float* gamma,
                                                               const float* beta, const float* slope, int act, int C, int HW, f16* y) {
    const int plane = blockIdx.y, c = plane % C;
    const int i = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= HW) return;
    const float m = mean ? mean[c] : 0.f, rs = mean ? rstd[c] : 1.f, ga = mean ? gamma[c] : 1.f, be = mean ? beta[c] : 0.f;
    const float sl = act == 2 ? slope[c] : 0.f;
    const long long o = (long long)plane * HW + i;
    float v[8];
    ld8_f16(x + o, HW - i, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float z = mean ? (v[e] - m) * rs * ga + be : v[e];
        if (act == 1) z = fmaxf(z, 0.f);
        else if (act == 2) z = z >= 0.f ? z : sl * z;
        else if (act == 3) z = 1.f / (1.f + expf(-z));
        v[e] = z;
    }
    st8_f16(y + o, HW - i, v);
}

// The same with the batch statistics FINALISED in the kernel (training): every block forms its channel's mean / rstd from the f64 sums
// (two loads, one rsqrt), block (0, plane c of clip 0) also publishes them for the backward pass, updates the running statistics and --
// block (0, 0) -- the batch counter: cruse_bn_finalize + cruse_counters_add + cruse_bn_nchw_fwd were three launches per BatchNorm2d.
__global__ __launch_bounds__(256) void bn_nchw_fwd_train_f16v_kernel(const f16* x, const double* sums, int nrep, double inv_count, double unbias, float eps,
                                                                     float momentum, const float* gamma, const float* beta, const float* slope, int act,
                                                                     int C, int HW, f16* y, float* mean_o, float* rstd_o, float* rmean, float* rvar,
                                                                     long long* nbt) {
    const int plane = blockIdx.y, c = plane % C;
    // the statistic: wave 0 folds the replicas (one lane each, fixed order of the shuffle tree) and hands mean / rstd to the block
    __shared__ float s_m, s_rs;
    __shared__ double s_md, s_var;
    if (threadIdx.x < 64) {
        double t1 = 0.0, t2 = 0.0;
        for (int r = threadIdx.x; r < nrep; r += 64) { t1 += sums[(size_t)r * 2 * C + c]; t2 += sums[(size_t)r * 2 * C + C + c]; }
        t1 = wave_sum_d(t1); t2 = wave_sum_d(t2);
        if (threadIdx.x == 0) {
            const double md0 = t1 * inv_count;
            double var0 = t2 * inv_count - md0 * md0;
            if (var0 < 0.0) var0 = 0.0;
            s_md = md0; s_var = var0; s_m = (float)md0; s_rs = (float)(1.0 / sqrt(var0 + (double)eps));
        }
    }
    __syncthreads();
    const double md = s_md, var = s_var;
    const float m = s_m, rs = s_rs;
    if (blockIdx.x == 0 && plane < C && threadIdx.x == 0) {
        mean_o[c] = m; rstd_o[c] = rs;
        if (rmean) {
            rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * md);
            rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * var * unbias);
        }
        if (nbt && plane == 0) *nbt += 1;
    }
    const int i = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= HW) return;
    const float ga = gamma[c], be = beta[c];
    const float sl = act == 2 ? slope[c] : 0.f;
    const long long o = (long long)plane * HW + i;
    float v[8];
    ld8_f16(x + o, HW - i, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float z = (v[e] - m) * rs * ga + be;
        if (act == 1) z = fmaxf(z, 0.f);
        else if (act == 2) z = z >= 0.f ? z : sl * z;
        else if (act == 3) z = 1.f / (1.f + expf(-z));
        v[e] = z;
    }
    st8_f16(y + o, HW - i, v);
}

__global__ __launch_bounds__(256) void bn_nchw_bwd_reduce_f16v_kernel(const f16* dy, const f16* x, const float* mean, const float* rstd,
                                                                      const float* gamma, const float* beta, const float* slope, int act,
                                                                      int N, int C, int HW, double* r, int r4) {
    __shared__ double red[4][4];
    const int c = blockIdx.x;
    const float m = mean ? mean[c] : 0.f, rs = mean ? rstd[c] : 1.f, ga = mean ? gamma[c] : 1.f, be = mean ? beta[c] : 0.f;
    const float sl = act == 2 ? slope[c] : 0.f;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
    const int chunk = ((HW + gridDim.z - 1) / gridDim.z + 7) & ~7, i0 = blockIdx.z * chunk, i1 = min(HW, i0 + chunk);
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const long long o = ((long long)n * C + c) * HW;
        float a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll 2
        for (int i = i0 + threadIdx.x * 8; i < i1; i += 2048) {
            float xv[8], dv[8];
            const int valid = i1 - i;
            ld8_f16(x + o + i, valid, xv);
            ld8_f16(dy + o + i, valid, dv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (xv[e] - m) * rs;
                const float z = xh * ga + be;
                float d = e < valid ? dv[e] : 0.f;
                if (act == 1) d = z > 0.f ? d : 0.f;
                else if (act == 2) { if (z < 0.f) { a3 += d * z; d *= sl; } }
                else if (act == 3) { const float sg = 1.f / (1.f + expf(-z)); d *= sg * (1.f - sg); }
                a1 += d; a2 += d * xh;
                if (e < valid) a4 += xh;
            }
        }
        s1 += a1; s2 += a2; s3 += a3; s4 += a4;
    }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2); s3 = wave_sum_d(s3); s4 = wave_sum_d(s4);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; red[2][threadIdx.x >> 6] = s3; red[3][threadIdx.x >> 6] = s4; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&r[c], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(&r[C + c], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        atomicAdd(&r[2 * C + c], red[2][0] + red[2][1] + red[2][2] + red[2][3]);
        if (r4) atomicAdd(&r[3 * C + c], red[3][0] + red[3][1] + red[3][2] + red[3][3]);
    }
}

// pg: block (0, plane c of clip 0) also adds the parameter gradients from the finished sums r (the separate bn_nchw_param_grads launch) and,
// with dx_sum, the sum over the channel of dx -- the bias gradient of the convolution that feeds this BatchNorm -- in closed form from r
// (r[3C + c] = sum of xhat, accumulated by the reduce pass), which would otherwise be a channel-sum pass over the dx just stored.
__global__ __launch_bounds__(256) void bn_nchw_bwd_apply_f16v_kernel(const f16* dy, const f16* x, const float* mean, const float* rstd,
                                                                     const float* gamma, const float* beta, const float* slope, int act,
                                                                     const double* rg, int nrep, double inv_count, int training, int C, int HW, f16* dx,
                                                                     float* dx_sum, int pg, float* dgamma, float* dbeta, float* dslope) {
    const int plane = blockIdx.y, c = plane % C;
    const int i = (blockIdx.x * 256 + threadIdx.x) * 8;
    // the channel's four sums: rg is [nrep][4][C] when a convolution's epilogue delivered them (cruse_conv2d_nchw_bnbwd), else [4 or 3][C]
    __shared__ double r[4];
    if (threadIdx.x < 32) {           // lane = replica * 4 + sum: eight replicas load at once, three exchanges fold them
        const int k = threadIdx.x & 3;
        double t = 0.0;
        if (nrep > 1) { for (int q = threadIdx.x >> 2; q < nrep; q += 8) t += rg[((size_t)q * 4 + k) * C + c]; }
        else if (threadIdx.x < 4 && (k < 3 || dx_sum != nullptr)) t = rg[(size_t)k * C + c];
        t += __shfl_xor(t, 4); t += __shfl_xor(t, 8); t += __shfl_xor(t, 16);
        if (threadIdx.x < 4) r[k] = t;
    }
    __syncthreads();
    if (pg && blockIdx.x == 0 && plane < C && threadIdx.x == 0) {
        if (dbeta) dbeta[c] += (float)r[0];
        if (dgamma) dgamma[c] += (float)r[1];
        if (dslope) dslope[c] += (float)r[2];
        if (dx_sum) {
            // sum over the channel of dx = ga rs (d - k1 - xh k2): the d and k1 terms cancel exactly, what is left is the rounding of the
            // mean in sum(xh) -- computed from the sums instead of re-reading the dx just stored (eval mode: ga rs sum(d))
            const double gr = mean ? (double)gamma[c] * (double)rstd[c] : 1.0;
            dx_sum[c] += (float)((mean && training) ? -gr * (r[1] * inv_count) * r[3] : gr * r[0]);
        }
    }
    if (i >= HW) return;
    const float m = mean ? mean[c] : 0.f, rs = mean ? rstd[c] : 1.f, ga = mean ? gamma[c] : 1.f, be = mean ? beta[c] : 0.f;
    const float sl = act == 2 ? slope[c] : 0.f;
    const float k1 = (mean && training) ? (float)(r[0] * inv_count) : 0.f, k2 = (mean && training) ? (float)(r[1] * inv_count) : 0.f;
    const long long o = (long long)plane * HW + i;
    float xv[8], dv[8];
    ld8_f16(x + o, HW - i, xv);
    ld8_f16(dy + o, HW - i, dv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float xh = (xv[e] - m) * rs;
        const float z = xh * ga + be;
        float d = dv[e];
        if (act == 1) d = z > 0.f ? d : 0.f;
        else if (act == 2) d = z >= 0.f ? d : sl * d;
        else if (act == 3) { const float sg = 1.f / (1.f + expf(-z)); d *= sg * (1.f - sg); }
        if (mean) {
            if (training) d -= k1 + xh * k2;
            d *= ga * rs;
        }
        dv[e] = d;
    }
    st8_f16(dx + o, HW - i, dv);
}

// dgamma += r[C+c], dbeta += r[c], dslope += r[2C+c]
__global__ void bn_nchw_param_grads_kernel(const double* r, int C, float* dgamma, float* dbeta, float* dslope) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (dbeta) dbeta[c] += (float)r[c];
    if (dgamma) dgamma[c] += (float)r[C + c];
    if (dslope) dslope[c] += (float)r[2 * C + c];
}

// out = a + b (residual adds of TFCM_Block / GroupGRU add_outputs on f16 storage) and f32 <-> f16 casts (where the f16
// part of a model starts / ends), 8 elements per thread
template <typename T>
__global__ void add_t_kernel(const T* a, const T* b, T* out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = (T)((float)a[i] + (float)b[i]);
}
template <typename S, typename D>
__global__ void cast_t_kernel(const S* src, D* dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = (D)(float)src[i];
}

// chunks per plane for the per-channel reductions: enough blocks to fill 256 CUs several times over, at least 4096
// elements per chunk
// LDS image of the f16 depthwise kernels: band of RB rows of the iterated tensor (Hiter rows of Witer columns), the staged tensor
// has Wsrc columns; false if the image does not fit 60 KB
inline bool dw_geometry(const TapTab& tt, int Hiter, int Wsrc, int Witer, long long planes, DwGeom* g, size_t* lds) {
    int dh_lo = 0, dh_hi = 0, dw_lo = 0, dw_hi = 0;
    for (int k = 0; k < tt.n; ++k) {
        dh_lo = tt.dh[k] < dh_lo ? tt.dh[k] : dh_lo; dh_hi = tt.dh[k] > dh_hi ? tt.dh[k] : dh_hi;
        dw_lo = tt.dw[k] < dw_lo ? tt.dw[k] : dw_lo; dw_hi = tt.dw[k] > dw_hi ? tt.dw[k] : dw_hi;
    }
    int RB = Hiter < 16 ? Hiter : 16;
    while (planes * cdiv(Hiter, RB) < 1024 && RB > 4) RB >>= 1;
    g->RB = RB; g->dh_lo = dh_lo; g->rows = RB + dh_hi - dh_lo;
    g->PADL = (-dw_lo + 7) & ~7;
    g->gpr = cdiv(Witer, 8);
    // a group reads columns [wo + dw, wo + dw + 8), wo <= 8 * (gpr - 1): the row must reach PADL + 8 * gpr + dw_hi
    int need = g->PADL + 8 * g->gpr + dw_hi;
    if (need < g->PADL + Wsrc) need = g->PADL + Wsrc;
    g->Ws = (need + 7) & ~7;
    *lds = (size_t)g->rows * g->Ws * 2 + 16;
    return *lds <= 60 * 1024 && Hiter > 0;
}

inline int zchunks(int gx, int gy, int HW) {
    int z = 1;
    while ((long long)gx * gy * z < 512 && HW / (z * 2) >= 1024) z *= 2;     // (768 blocks at the config-5 shape: 20 us; 1536: 25, 6144: 49 -- the f64 atomics of many small blocks)
    return z;
}

inline int gblocks(long long n, int per = 1024, int cap = 8192) {
    long long g = (n + per - 1) / per;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace

#define ST(s) ((hipStream_t)(s))

#define CRUSE_DT_CHECK(name) CRUSE_REQUIRE(dtype == CRUSE_DT_F32 || dtype == CRUSE_DT_F16, CRUSE_E_DTYPE, name ": unknown storage dtype %d", dtype)

namespace {
template <typename T>
int conv2d_nchw_t(const void* x, const float* w, const float* bias, void* y, int B, int Cin, int Hin, int Win, int Cout, int Hout,
                  int Wout, int KH, int KW, int sh, int sw, int dh, int dw, int pt, int pl, int groups, int up_w, int transposed,
                  int act, const float* slope, int accumulate, hipStream_t s, double* bn_sums = nullptr, int bn_nrep = 1,
                  bool* bn_done = nullptr, const void* res = nullptr, bool* res_done = nullptr, const GConv<T>* bb = nullptr, bool* bb_done = nullptr) {
    GConv<T> a = {(const T*)x, w, bias, (T*)y, B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, sh, sw, dh, dw, pt, pl, groups, up_w,
                  transposed, act, accumulate, slope, nullptr, 1, nullptr};
    if (bn_done) *bn_done = false;
    if (res_done) *res_done = false;
    if (bb_done) *bb_done = false;
    auto set_bb = [&]() {
        if (bb == nullptr) return;
        a.bb_x = bb->bb_x; a.bb_mean = bb->bb_mean; a.bb_rstd = bb->bb_rstd; a.bb_gamma = bb->bb_gamma; a.bb_beta = bb->bb_beta; a.bb_slope = bb->bb_slope;
        a.bb_act = bb->bb_act; a.bb_r = bb->bb_r; a.bb_nrep = bb->bb_nrep;
        *bb_done = true;
    };
    const bool pointwise = KH == 1 && KW == 1 && sh == 1 && sw == 1 && pt == 0 && pl == 0 && groups == 1 && up_w == 1 &&
                           Hout == Hin && Wout == Win;
    if constexpr (sizeof(T) == 2) {
        // f16 storage: the pointwise convolutions run on the matrix cores (weights resident as A fragments)
        const int MT = cdiv(Cout, 16), KS = cdiv(Cin, 32);
        if (pointwise && MT <= 2 && KS <= 2 && !accumulate && !cruse_opt("pw_valu", 0)) {        // (pw_valu = 2: the gather kernel below)
            // operand transposed through LDS (gconv_pointwise_tr_f16_kernel): 16-byte accesses on both sides
            const long long nchunk = (long long)B * cdivl((long long)Hin * Win, 64);
            const int cap = 768;                                                             // (blocks; measured 256 .. 2048 at the config-5 shape: 26.9 / 19.0 / 17.9 / 19.2 / 21.3 / 23.3 us)
            const int nb = (int)(cdivl(nchunk, 4) > cap ? cap : cdivl(nchunk, 4));
            const int mtc = MT <= 1 ? 1 : 2, ksc = KS <= 1 ? 1 : 2;
            const size_t lds = (size_t)4 * (ksc * 32 + mtc * 16) * 144;
            if (bn_sums) { a.bn_sums = bn_sums; a.bn_nrep = bn_nrep; *bn_done = true; }
            if (res) { a.res = (const T*)res; *res_done = true; }
            set_bb();
            const int epi = a.bb_r ? 2 : (a.bn_sums ? 1 : 0);                                // (never both: the sums of the output / of the BatchNorm in front of a data gradient)
#define PWT_LAUNCH(mt, ks, ep) do { int rc = cruse_ensure_dyn_lds((const void*)gconv_pointwise_tr_f16_kernel<mt, ks, ep>, lds, "conv2d_nchw pointwise"); \
                if (rc) return rc; \
                hipLaunchKernelGGL((gconv_pointwise_tr_f16_kernel<mt, ks, ep>), dim3(nb), dim3(256), lds, s, a); } while (0)
#define PWT_CASE(mt, ks) do { if (epi == 0) PWT_LAUNCH(mt, ks, 0); else if (epi == 1) PWT_LAUNCH(mt, ks, 1); else PWT_LAUNCH(mt, ks, 2); } while (0)
            if (mtc == 1 && ksc == 1) PWT_CASE(1, 1); else if (mtc == 1) PWT_CASE(1, 2); else if (ksc == 1) PWT_CASE(2, 1); else PWT_CASE(2, 2);
#undef PWT_CASE
#undef PWT_LAUNCH
            CRUSE_LAUNCH_CHECK("conv2d_nchw pointwise mfma f16 (LDS-transposed)");
            return CRUSE_OK;
        }
        if (pointwise && MT <= 4 && KS <= 4 && cruse_opt("pw_valu", 0) != 1) {
            const long long ntile = (long long)B * cdivl((long long)Hin * Win, 16);
            const int nb = (int)(cdivl(ntile, 16) > 8192 ? 8192 : cdivl(ntile, 16));
#define PW_CASE(mt, ks) hipLaunchKernelGGL((gconv_pointwise_mfma_f16_kernel<mt, ks>), dim3(nb), dim3(256), 0, s, a)
            const int mtc = MT <= 1 ? 1 : MT <= 2 ? 2 : 4, ksc = KS <= 1 ? 1 : KS <= 2 ? 2 : 4;
            if (mtc == 1 && ksc == 1) PW_CASE(1, 1); else if (mtc == 1 && ksc == 2) PW_CASE(1, 2); else if (mtc == 1) PW_CASE(1, 4);
            else if (mtc == 2 && ksc == 1) PW_CASE(2, 1); else if (mtc == 2 && ksc == 2) PW_CASE(2, 2); else if (mtc == 2) PW_CASE(2, 4);
            else if (ksc == 1) PW_CASE(4, 1); else if (ksc == 2) PW_CASE(4, 2); else PW_CASE(4, 4);
#undef PW_CASE
            CRUSE_LAUNCH_CHECK("conv2d_nchw pointwise mfma f16");
            return CRUSE_OK;
        }
    }
    if (pointwise && Cout <= 64 && Cin <= 256) {
        const size_t lds = (size_t)Cout * Cin * sizeof(float);
        const int nb = gblocks((long long)B * Hin * Win, 256, 8192);
        if (Cout <= 16) hipLaunchKernelGGL((gconv_pointwise_kernel<T, 16>), dim3(nb), dim3(256), lds, s, a);
        else if (Cout <= 32) hipLaunchKernelGGL((gconv_pointwise_kernel<T, 32>), dim3(nb), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((gconv_pointwise_kernel<T, 64>), dim3(nb), dim3(256), lds, s, a);
        CRUSE_LAUNCH_CHECK("conv2d_nchw pointwise");
        return CRUSE_OK;
    }
    if (groups == Cin && Cin == Cout && up_w == 1 && sh == 1 && sw == 1 && KH * KW <= 9 && (long long)B * Cout < 65536 &&
        (long long)Hin * Win < (1ll << 30) && (long long)Hout * Wout < (1ll << 30) && cruse_opt("pw_valu", 0) != 1) {
        TapTab tt = {};
        tt.n = KH * KW;
        for (int kh = 0; kh < KH; ++kh)
            for (int kw = 0; kw < KW; ++kw) {
                tt.dh[kh * KW + kw] = transposed ? pt - kh * dh : kh * dh - pt;
                tt.dw[kh * KW + kw] = transposed ? pl - kw * dw : kw * dw - pl;
            }
        if constexpr (sizeof(T) == 2) {
            DwGeom g;
            size_t lds;
            if (!cruse_opt("dw_nolds", 0) && dw_geometry(tt, Hout, Win, Wout, B * Cout, &g, &lds)) {
                int rc = cruse_ensure_dyn_lds((const void*)gconv_depthwise_f16_kernel, lds, "conv2d_nchw depthwise");
                if (rc) return rc;
                if (bn_sums) { a.bn_sums = bn_sums; a.bn_nrep = bn_nrep; *bn_done = true; }
                set_bb();
                hipLaunchKernelGGL(gconv_depthwise_f16_kernel, dim3(cdiv(Hout, g.RB), B * Cout), dim3(256), lds, s, a, tt, g);
                CRUSE_LAUNCH_CHECK("conv2d_nchw depthwise (f16, LDS image)");
                return CRUSE_OK;
            }
        }
        hipLaunchKernelGGL(gconv_depthwise_kernel<T>, dim3(cdiv(Hout * Wout, 8 * 256), B * Cout), dim3(256), 0, s, a, tt);
        CRUSE_LAUNCH_CHECK("conv2d_nchw depthwise");
        return CRUSE_OK;
    }
    hipLaunchKernelGGL(gconv_kernel<T>, dim3(gblocks((long long)B * Cout * Hout * Wout, 256, 16384)), dim3(256), 0, s, a);
    CRUSE_LAUNCH_CHECK("conv2d_nchw");
    return CRUSE_OK;
}

template <typename T>
int wgrad_nchw_t(const void* S, const void* Bg, float* dw, int N, int CA, int HS, int WS, int CB, int HB, int WB, int KH, int KW,
                 int sh, int sw, int dh, int dw_, int pt, int pl, int groups, int up_w, hipStream_t s, float* db = nullptr) {
    GWgrad<T> a = {(const T*)S, (const T*)Bg, dw, N, CA, HS, WS, CB, HB, WB, KH, KW, sh, sw, dh, dw_, pt, pl, groups, up_w, nullptr};
    const bool fast = cruse_opt("pw_valu", 0) != 1;
    // db: only the pointwise MFMA kernel delivers it (a spare column of its last column tile carries the constant 1); every other form
    // runs the channel-sum pass first
    const bool pw_db = db != nullptr && sizeof(T) == 2 && fast && KH == 1 && KW == 1 && groups == 1 && sh == 1 && sw == 1 && pt == 0 && pl == 0 &&
                       up_w == 1 && HS == HB && WS == WB && CA <= 32 && CB < 32 && CB % 16 != 0;
    if (db != nullptr && !pw_db) {
        const int rc = cruse_nchw_channel_sum(S, N, CA, HS * WS, db, sizeof(T) == 2 ? CRUSE_DT_F16 : CRUSE_DT_F32, s);
        if (rc) return rc;
    }
    a.db = pw_db ? db : nullptr;
    if constexpr (sizeof(T) == 2) {
        if (fast && KH == 1 && KW == 1 && groups == 1 && sh == 1 && sw == 1 && pt == 0 && pl == 0 && up_w == 1 && HS == HB && WS == WB &&
            CA <= 32 && CB <= 32) {
            const long long hw = (long long)HS * WS;
            int run = cruse_opt("wgpw_run", 4096);              // positions per 16-wave block (a multiple of 1024)
            { const long long want = cruse_opt("wgpw_blocks", 256); while ((long long)N * cdivl(hw, run) < want && run > 1024) run >>= 1; }
            const dim3 grid((unsigned)((long long)N * cdivl(hw, run)));
            run |= cruse_opt("wgpw_dbg", 0) << 24;
            const int mt = cdiv(CA, 16), ntl = cdiv(CB, 16);
#define WGPW_CASE(m_, n_) do { const size_t lds = (size_t)16 * m_ * n_ * 256 * sizeof(float); \
                int rc = cruse_ensure_dyn_lds((const void*)gconv_wgrad_pw_mfma_f16_kernel<m_, n_>, lds, "conv2d_nchw_wgrad pointwise"); \
                if (rc) return rc; \
                hipLaunchKernelGGL((gconv_wgrad_pw_mfma_f16_kernel<m_, n_>), grid, dim3(1024), lds, s, a, run); } while (0)
            if (mt == 1 && ntl == 1) WGPW_CASE(1, 1);
            else if (mt == 1) WGPW_CASE(1, 2);
            else if (ntl == 1) WGPW_CASE(2, 1);
            else WGPW_CASE(2, 2);
#undef WGPW_CASE
            CRUSE_LAUNCH_CHECK("conv2d_nchw_wgrad pointwise mfma f16");
            return CRUSE_OK;
        }
    }
    if (fast && groups == CA && CA == CB && up_w == 1 && KH * KW <= 9 && (long long)N * CA < 65536 && (long long)HB * WB < (1ll << 30) &&
        (long long)HS * WS < (1ll << 30)) {
        int band = 16;                                   // S positions per thread
        while ((long long)N * CA * cdiv(HS * WS, band * 256) < 2048 && band > 4) band >>= 1;
        TapTab tt = {};
        tt.n = KH * KW;
        for (int kh = 0; kh < KH; ++kh)
            for (int kw = 0; kw < KW; ++kw) { tt.dh[kh * KW + kw] = kh * dh - pt; tt.dw[kh * KW + kw] = kw * dw_ - pl; }
        if constexpr (sizeof(T) == 2) {
            DwGeom g;
            size_t lds;
            if (sh == 1 && sw == 1 && !cruse_opt("dw_nolds", 0) && dw_geometry(tt, HS, WB, WS, N * CA, &g, &lds)) {
                int rc = cruse_ensure_dyn_lds((const void*)gconv_wgrad_depthwise_f16_kernel, lds, "conv2d_nchw_wgrad depthwise");
                if (rc) return rc;
                hipLaunchKernelGGL(gconv_wgrad_depthwise_f16_kernel, dim3(cdiv(HS, g.RB), N * CA), dim3(256), lds, s, a, tt, g);
                CRUSE_LAUNCH_CHECK("conv2d_nchw_wgrad depthwise (f16, LDS image)");
                return CRUSE_OK;
            }
        }
        hipLaunchKernelGGL(gconv_wgrad_depthwise_kernel<T>, dim3(cdiv(HS * WS, band * 256), N * CA), dim3(256), 0, s, a, band, tt);
        CRUSE_LAUNCH_CHECK("conv2d_nchw_wgrad depthwise");
        return CRUSE_OK;
    }
    const size_t row_lds = ((size_t)CA * WS + (size_t)CB * WB) * sizeof(float);
    if (row_lds <= 150 * 1024 && (long long)N * HS >= 8) {        // a row pair fits LDS (the per-weight kernel below is the fallback)
        int rc = cruse_ensure_dyn_lds((const void*)gconv_wgrad_rows_kernel<T>, row_lds, "conv2d_nchw_wgrad");
        if (rc) return rc;
        hipLaunchKernelGGL(gconv_wgrad_rows_kernel<T>, dim3(N * HS), dim3(256), row_lds, s, a);
        CRUSE_LAUNCH_CHECK("conv2d_nchw_wgrad rows");
        return CRUSE_OK;
    }
    const int nw = CA * (CB / groups) * KH * KW;
    int ny = 1;
    while (ny < N && (long long)nw * ny < 2048) ny *= 2;
    if (ny > N) ny = N;
    hipLaunchKernelGGL(gconv_wgrad_kernel<T>, dim3(nw, ny), dim3(256), 0, s, a);
    CRUSE_LAUNCH_CHECK("conv2d_nchw_wgrad");
    return CRUSE_OK;
}

}  // namespace
extern "C" int cruse_nchw_channel_sum(const void* x, int N, int C, int HW, float* out, int dtype, void* stream);
namespace {
template <typename T>
int bn_nchw_bwd_t(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                  const float* slope, int act, int training, int N, int C, int HW, double* scratch, void* dx, float* dgamma,
                  float* dbeta, float* dslope, hipStream_t s, float* dx_sum = nullptr, int reduce_done = 0, int r_nrep = 1) {
    const long long total = (long long)N * C * HW;
    if constexpr (sizeof(T) == 2) {
        if ((long long)N * C < 65536) {                      // (grid.y = planes)
            if (!reduce_done) {
                hipLaunchKernelGGL(bn_nchw_bwd_reduce_f16v_kernel, dim3(C, N < 32 ? N : 32, zchunks(C, N < 32 ? N : 32, HW)), dim3(256), 0, s, (const f16*)dy,
                                   (const f16*)x, mean, rstd, gamma, beta, slope, act, N, C, HW, scratch, dx_sum != nullptr ? 1 : 0);
                CRUSE_LAUNCH_CHECK("bn_nchw_bwd_reduce");
            }
            hipLaunchKernelGGL(bn_nchw_bwd_apply_f16v_kernel, dim3(cdiv(cdiv(HW, 8), 256), N * C), dim3(256), 0, s, (const f16*)dy, (const f16*)x, mean, rstd,
                               gamma, beta, slope, act, scratch, reduce_done ? r_nrep : 1, 1.0 / ((double)N * HW), training, C, HW, (f16*)dx, dx_sum, 1,
                               mean ? dgamma : nullptr, mean ? dbeta : nullptr, act == 2 ? dslope : nullptr);
            CRUSE_LAUNCH_CHECK("bn_nchw_bwd_apply");
            return CRUSE_OK;
        }
    }
    if (reduce_done) { cruse_set_error("bn_nchw_bwd_ex: sums delivered by a convolution come with the f16 kernels only"); return CRUSE_E_SHAPE; }
    hipLaunchKernelGGL(bn_nchw_bwd_reduce_kernel<T>, dim3(C, N < 32 ? N : 32, zchunks(C, N < 32 ? N : 32, HW)), dim3(256), 0, s, (const T*)dy, (const T*)x, mean, rstd,
                       gamma, beta, slope, act, N, C, HW, scratch);
    CRUSE_LAUNCH_CHECK("bn_nchw_bwd_reduce");
    hipLaunchKernelGGL(bn_nchw_bwd_apply_kernel<T>, dim3(gblocks(total)), dim3(256), 0, s, (const T*)dy, (const T*)x, mean, rstd, gamma,
                       beta, slope, act, scratch, 1.0 / ((double)N * HW), training, total, C, HW, (T*)dx);
    CRUSE_LAUNCH_CHECK("bn_nchw_bwd_apply");
    hipLaunchKernelGGL(bn_nchw_param_grads_kernel, dim3(cdiv(C, 64)), dim3(64), 0, s, scratch, C, mean ? dgamma : nullptr,
                       mean ? dbeta : nullptr, act == 2 ? dslope : nullptr);
    CRUSE_LAUNCH_CHECK("bn_nchw_param_grads");
    if (dx_sum != nullptr) return cruse_nchw_channel_sum(dx, N, C, HW, dx_sum, sizeof(T) == 2 ? CRUSE_DT_F16 : CRUSE_DT_F32, s);
    return CRUSE_OK;
}
}  // namespace

extern "C" int cruse_conv2d_nchw(const void* x, const float* w, const float* bias, void* y,
                                 int B, int Cin, int Hin, int Win, int Cout, int Hout, int Wout,
                                 int KH, int KW, int sh, int sw, int dh, int dw, int pt, int pl,
                                 int groups, int up_w, int transposed, int act, const float* slope, int accumulate,
                                 int dtype, void* stream) {
    CRUSE_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, CRUSE_E_SHAPE,
                  "conv2d_nchw: bad shape B=%d Cin=%d Cout=%d in %dx%d out %dx%d", B, Cin, Cout, Hin, Win, Hout, Wout);
    CRUSE_REQUIRE(KH > 0 && KW > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && groups > 0 && up_w > 0, CRUSE_E_SHAPE,
                  "conv2d_nchw: bad kernel geometry");
    CRUSE_REQUIRE(Cin % groups == 0 && Cout % groups == 0, CRUSE_E_SHAPE, "conv2d_nchw: channels %d/%d not divisible by groups %d", Cin, Cout, groups);
    CRUSE_REQUIRE(!(transposed && up_w != 1), CRUSE_E_SHAPE, "conv2d_nchw: upsampling only with the conv form");
    CRUSE_REQUIRE(act >= 0 && act <= 2 && (act != 2 || slope) && !(accumulate && act), CRUSE_E_SHAPE, "conv2d_nchw: bad activation");
    CRUSE_DT_CHECK("conv2d_nchw");
    if (dtype == CRUSE_DT_F16)
        return conv2d_nchw_t<f16>(x, w, bias, y, B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, sh, sw, dh, dw, pt, pl, groups, up_w,
                                  transposed, act, slope, accumulate, ST(stream));
    return conv2d_nchw_t<float>(x, w, bias, y, B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, sh, sw, dh, dw, pt, pl, groups, up_w,
                                transposed, act, slope, accumulate, ST(stream));
}

extern "C" int cruse_bn_nchw_stats_ex(const void* x, int N, int C, int HW, double* sums, int zeroed, int dtype, void* stream);

extern "C" int cruse_add_nchw(const void* a, const void* b, void* out, long long n, int dtype, void* stream);

// cruse_conv2d_nchw (no bias, no activation, no accumulation) whose output y is the GRADIENT wrt the output of a BatchNorm2d (+ act) with input
// bn_x (the data gradient of the convolution that consumed that BatchNorm's output): the f16 pointwise / depthwise kernels also accumulate that
// BatchNorm's backward sums of the stored y into r[r_nrep][4][Cout] f64 (cleared by the caller; cruse_bn_nchw_bwd_ex(sums_replicas = r_nrep) then
// skips its reduce pass over y and bn_x).  Returns *delivered = 0 when the form that ran has no such epilogue (the caller runs the plain backward).
extern "C" int cruse_conv2d_nchw_bnbwd(const void* x, const float* w, void* y,
                                       int B, int Cin, int Hin, int Win, int Cout, int Hout, int Wout,
                                       int KH, int KW, int sh, int sw, int dh, int dw, int pt, int pl, int groups, int transposed,
                                       const void* bn_x, const float* bn_mean, const float* bn_rstd, const float* bn_gamma, const float* bn_beta,
                                       const float* bn_slope, int bn_act, double* r, int r_nrep, int* delivered, int dtype, void* stream) {
    CRUSE_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, CRUSE_E_SHAPE, "conv2d_nchw_bnbwd: bad shape");
    CRUSE_REQUIRE(KH > 0 && KW > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && groups > 0 && Cin % groups == 0 && Cout % groups == 0, CRUSE_E_SHAPE,
                  "conv2d_nchw_bnbwd: bad kernel geometry");
    CRUSE_REQUIRE(bn_x && bn_mean && bn_rstd && bn_gamma && bn_beta && bn_act >= 0 && bn_act <= 3 && (bn_act != 2 || bn_slope) && r && r_nrep >= 1 && delivered,
                  CRUSE_E_SHAPE, "conv2d_nchw_bnbwd: BatchNorm arguments");
    CRUSE_DT_CHECK("conv2d_nchw_bnbwd");
    bool done = false;
    int rc;
    if (dtype == CRUSE_DT_F16) {
        GConv<f16> bb = {};
        bb.bb_x = (const f16*)bn_x; bb.bb_mean = bn_mean; bb.bb_rstd = bn_rstd; bb.bb_gamma = bn_gamma; bb.bb_beta = bn_beta; bb.bb_slope = bn_slope;
        bb.bb_act = bn_act; bb.bb_r = r; bb.bb_nrep = r_nrep;
        rc = conv2d_nchw_t<f16>(x, w, nullptr, y, B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, sh, sw, dh, dw, pt, pl, groups, 1, transposed, 0, nullptr,
                                0, ST(stream), nullptr, 1, nullptr, nullptr, nullptr, &bb, &done);
    } else {
        rc = conv2d_nchw_t<float>(x, w, nullptr, y, B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, sh, sw, dh, dw, pt, pl, groups, 1, transposed, 0, nullptr,
                                  0, ST(stream));
    }
    *delivered = done ? 1 : 0;
    return rc;
}

// cruse_conv2d_nchw (no accumulation) with the two things that follow a convolution in the reference's blocks folded in (both optional):
//   residual != NULL: y = conv(x) + residual (TFCM_Block's `outs + inps`, mtfaa.py:191) -- in the epilogue of the f16 LDS-transposed pointwise
//     kernel, else by cruse_add_nchw on y;
//   bn_sums != NULL: the BatchNorm batch sums of the output, [bn_nrep][2 * Cout] f64 CLEARED BY THE CALLER, the statistic is the sum over the
//     replicas (cruse_bn_nchw_fwd_train folds them) -- from the epilogue of the f16 pointwise-MFMA and LDS-depthwise kernels (nn.Conv2d ->
//     nn.BatchNorm2d, mtfaa.py:170-183; Conv2dNormAct, cust_conv.py:15-111), else by the statistics pass over y into replica 0.
extern "C" int cruse_conv2d_nchw_ex(const void* x, const float* w, const float* bias, const void* residual, void* y,
                                    int B, int Cin, int Hin, int Win, int Cout, int Hout, int Wout,
                                    int KH, int KW, int sh, int sw, int dh, int dw, int pt, int pl,
                                    int groups, int up_w, int transposed, int act, const float* slope,
                                    double* bn_sums, int bn_nrep, int dtype, void* stream) {
    CRUSE_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, CRUSE_E_SHAPE, "conv2d_nchw_ex: bad shape");
    CRUSE_REQUIRE(KH > 0 && KW > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && groups > 0 && up_w > 0, CRUSE_E_SHAPE,
                  "conv2d_nchw_ex: bad kernel geometry");
    CRUSE_REQUIRE(Cin % groups == 0 && Cout % groups == 0 && !(transposed && up_w != 1), CRUSE_E_SHAPE, "conv2d_nchw_ex: groups / form");
    CRUSE_REQUIRE(act >= 0 && act <= 2 && (act != 2 || slope) && (bn_sums == nullptr || bn_nrep >= 1) && !(bn_sums && residual), CRUSE_E_SHAPE,
                  "conv2d_nchw_ex: arguments (statistics of a sum are not offered)");
    CRUSE_DT_CHECK("conv2d_nchw_ex");
    bool done = false, rdone = false;
    int rc;
    if (dtype == CRUSE_DT_F16)
        rc = conv2d_nchw_t<f16>(x, w, bias, y, B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, sh, sw, dh, dw, pt, pl, groups, up_w, transposed, act, slope,
                                0, ST(stream), bn_sums, bn_nrep, &done, residual, &rdone);
    else
        rc = conv2d_nchw_t<float>(x, w, bias, y, B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, sh, sw, dh, dw, pt, pl, groups, up_w, transposed, act, slope,
                                  0, ST(stream), bn_sums, bn_nrep, &done, residual, &rdone);
    if (rc) return rc;
    if (residual && !rdone) return cruse_add_nchw(y, residual, y, (long long)B * Cout * Hout * Wout, dtype, stream);
    if (bn_sums && !done) return cruse_bn_nchw_stats_ex(y, B, Cout, Hout * Wout, bn_sums, 1, dtype, stream);
    return CRUSE_OK;
}

extern "C" int cruse_conv2d_nchw_wgrad(const void* S, const void* Bg, float* dw,
                                       int N, int CA, int HS, int WS, int CB, int HB, int WB,
                                       int KH, int KW, int sh, int sw, int dh, int dw_, int pt, int pl,
                                       int groups, int up_w, int dtype, void* stream) {
    CRUSE_REQUIRE(N > 0 && CA > 0 && CB > 0 && HS > 0 && WS > 0 && HB > 0 && WB > 0 && KH > 0 && KW > 0, CRUSE_E_SHAPE,
                  "conv2d_nchw_wgrad: bad shape");
    CRUSE_REQUIRE(CA % groups == 0 && CB % groups == 0 && up_w > 0, CRUSE_E_SHAPE, "conv2d_nchw_wgrad: groups");
    CRUSE_DT_CHECK("conv2d_nchw_wgrad");
    if (dtype == CRUSE_DT_F16)
        return wgrad_nchw_t<f16>(S, Bg, dw, N, CA, HS, WS, CB, HB, WB, KH, KW, sh, sw, dh, dw_, pt, pl, groups, up_w, ST(stream));
    return wgrad_nchw_t<float>(S, Bg, dw, N, CA, HS, WS, CB, HB, WB, KH, KW, sh, sw, dh, dw_, pt, pl, groups, up_w, ST(stream));
}

// cruse_conv2d_nchw_wgrad of a Conv2d (S = dy) that also accumulates the bias gradient db[ca] += sum_{n,h,w} dy[n,ca,h,w] -- inside the
// pointwise MFMA kernel where that form runs (one launch for nn.Conv2d's weight + bias gradients), else by the channel-sum pass
extern "C" int cruse_conv2d_nchw_wgrad_ex(const void* S, const void* Bg, float* dw, float* db,
                                          int N, int CA, int HS, int WS, int CB, int HB, int WB,
                                          int KH, int KW, int sh, int sw, int dh, int dw_, int pt, int pl,
                                          int groups, int up_w, int dtype, void* stream) {
    CRUSE_REQUIRE(N > 0 && CA > 0 && CB > 0 && HS > 0 && WS > 0 && HB > 0 && WB > 0 && KH > 0 && KW > 0, CRUSE_E_SHAPE,
                  "conv2d_nchw_wgrad_ex: bad shape");
    CRUSE_REQUIRE(CA % groups == 0 && CB % groups == 0 && up_w > 0, CRUSE_E_SHAPE, "conv2d_nchw_wgrad_ex: groups");
    CRUSE_DT_CHECK("conv2d_nchw_wgrad_ex");
    if (dtype == CRUSE_DT_F16)
        return wgrad_nchw_t<f16>(S, Bg, dw, N, CA, HS, WS, CB, HB, WB, KH, KW, sh, sw, dh, dw_, pt, pl, groups, up_w, ST(stream), db);
    return wgrad_nchw_t<float>(S, Bg, dw, N, CA, HS, WS, CB, HB, WB, KH, KW, sh, sw, dh, dw_, pt, pl, groups, up_w, ST(stream), db);
}

extern "C" int cruse_nchw_channel_sum(const void* x, int N, int C, int HW, float* out, int dtype, void* stream) {
    CRUSE_REQUIRE(N > 0 && C > 0 && HW > 0, CRUSE_E_SHAPE, "nchw_channel_sum: bad shape");
    CRUSE_DT_CHECK("nchw_channel_sum");
    if (dtype == CRUSE_DT_F16)
        hipLaunchKernelGGL(nchw_channel_sum_f16v_kernel, dim3(C, N < 16 ? N : 16, zchunks(C, N < 16 ? N : 16, HW)), dim3(256), 0, ST(stream), (const f16*)x, N, C, HW, out);
    else
        hipLaunchKernelGGL(nchw_channel_sum_kernel<float>, dim3(C, N < 16 ? N : 16, zchunks(C, N < 16 ? N : 16, HW)), dim3(256), 0, ST(stream), (const float*)x, N, C, HW, out);
    CRUSE_LAUNCH_CHECK("nchw_channel_sum");
    return CRUSE_OK;
}

extern "C" int cruse_downsum_w(const void* dxu, long long rows, int W, int up, void* dx, int dtype, void* stream) {
    CRUSE_REQUIRE(rows > 0 && W > 0 && up > 0, CRUSE_E_SHAPE, "downsum_w: bad shape");
    CRUSE_DT_CHECK("downsum_w");
    if (dtype == CRUSE_DT_F32 && up == 2 && ((rows * W) & 1) == 0 && (((uintptr_t)dxu | (uintptr_t)dx) & 15) == 0) {
        hipLaunchKernelGGL(downsum_w2_f32_kernel, dim3(gblocks(rows * W / 2)), dim3(256), 0, ST(stream), (const float4*)dxu, rows * W / 2, (float2*)dx);
        CRUSE_LAUNCH_CHECK("downsum_w");
        return CRUSE_OK;
    }
    if (dtype == CRUSE_DT_F16)
        hipLaunchKernelGGL(downsum_w_kernel<f16>, dim3(gblocks(rows * W)), dim3(256), 0, ST(stream), (const f16*)dxu, rows, W, up, (f16*)dx);
    else
        hipLaunchKernelGGL(downsum_w_kernel<float>, dim3(gblocks(rows * W)), dim3(256), 0, ST(stream), (const float*)dxu, rows, W, up, (float*)dx);
    CRUSE_LAUNCH_CHECK("downsum_w");
    return CRUSE_OK;
}

extern "C" int cruse_upsample_w(const void* x, long long rows, int W, int up, void* xu, int dtype, void* stream) {
    CRUSE_REQUIRE(rows > 0 && W > 0 && up > 0, CRUSE_E_SHAPE, "upsample_w: bad shape");
    CRUSE_DT_CHECK("upsample_w");
    if (dtype == CRUSE_DT_F32 && up == 2 && ((rows * W) & 1) == 0 && (((uintptr_t)x | (uintptr_t)xu) & 15) == 0)
        hipLaunchKernelGGL(upsample_w2_f32_kernel, dim3(gblocks(rows * W / 2)), dim3(256), 0, ST(stream), (const float2*)x, rows * W / 2, (float4*)xu);
    else if (dtype == CRUSE_DT_F16)
        hipLaunchKernelGGL(upsample_w_kernel<f16>, dim3(gblocks(rows * W)), dim3(256), 0, ST(stream), (const f16*)x, rows, W, up, (f16*)xu);
    else
        hipLaunchKernelGGL(upsample_w_kernel<float>, dim3(gblocks(rows * W)), dim3(256), 0, ST(stream), (const float*)x, rows, W, up, (float*)xu);
    CRUSE_LAUNCH_CHECK("upsample_w");
    return CRUSE_OK;
}

extern "C" int cruse_bn_nchw_stats(const void* x, int N, int C, int HW, double* sums, int dtype, void* stream) {
    CRUSE_REQUIRE(N > 0 && C > 0 && HW > 0, CRUSE_E_SHAPE, "bn_nchw_stats: bad shape");
    CRUSE_DT_CHECK("bn_nchw_stats");
    { int rc = cruse_zero_async(sums, 2 * (size_t)C * sizeof(double), ST(stream), "bn_nchw_stats"); if (rc) return rc; }
    if (dtype == CRUSE_DT_F16)
        hipLaunchKernelGGL(bn_nchw_stats_f16v_kernel, dim3(C, N < 32 ? N : 32, zchunks(C, N < 32 ? N : 32, HW)), dim3(256), 0, ST(stream), (const f16*)x, N, C, HW, sums);
    else
        hipLaunchKernelGGL(bn_nchw_stats_kernel<float>, dim3(C, N < 32 ? N : 32, zchunks(C, N < 32 ? N : 32, HW)), dim3(256), 0, ST(stream), (const float*)x, N, C, HW, sums);
    CRUSE_LAUNCH_CHECK("bn_nchw_stats");
    return CRUSE_OK;
}

// cruse_bn_nchw_stats into sums the caller has already cleared (zeroed != 0: no fill launch in front of the pass)
extern "C" int cruse_bn_nchw_stats_ex(const void* x, int N, int C, int HW, double* sums, int zeroed, int dtype, void* stream) {
    CRUSE_REQUIRE(N > 0 && C > 0 && HW > 0, CRUSE_E_SHAPE, "bn_nchw_stats_ex: bad shape");
    CRUSE_DT_CHECK("bn_nchw_stats_ex");
    if (!zeroed) { int rc = cruse_zero_async(sums, 2 * (size_t)C * sizeof(double), ST(stream), "bn_nchw_stats_ex"); if (rc) return rc; }
    if (dtype == CRUSE_DT_F16)
        hipLaunchKernelGGL(bn_nchw_stats_f16v_kernel, dim3(C, N < 32 ? N : 32, zchunks(C, N < 32 ? N : 32, HW)), dim3(256), 0, ST(stream), (const f16*)x, N, C, HW, sums);
    else
        hipLaunchKernelGGL(bn_nchw_stats_kernel<float>, dim3(C, N < 32 ? N : 32, zchunks(C, N < 32 ? N : 32, HW)), dim3(256), 0, ST(stream), (const float*)x, N, C, HW, sums);
    CRUSE_LAUNCH_CHECK("bn_nchw_stats_ex");
    return CRUSE_OK;
}

extern "C" int cruse_bn_nchw_fwd(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                 const float* slope, int act, int N, int C, int HW, void* y, int dtype, void* stream) {
    CRUSE_REQUIRE(N > 0 && C > 0 && HW > 0 && act >= 0 && act <= 3 && (act != 2 || slope), CRUSE_E_SHAPE, "bn_nchw_fwd: bad arguments");
    CRUSE_REQUIRE((mean == nullptr) == (rstd == nullptr) && (mean == nullptr || (gamma && beta)), CRUSE_E_SHAPE, "bn_nchw_fwd: statistics");
    CRUSE_DT_CHECK("bn_nchw_fwd");
    const long long total = (long long)N * C * HW;
    if (dtype == CRUSE_DT_F16 && (long long)N * C < 65536)
        hipLaunchKernelGGL(bn_nchw_fwd_f16v_kernel, dim3(cdiv(cdiv(HW, 8), 256), N * C), dim3(256), 0, ST(stream), (const f16*)x, mean, rstd, gamma, beta,
                           slope, act, C, HW, (f16*)y);
    else if (dtype == CRUSE_DT_F16)
        hipLaunchKernelGGL(bn_nchw_fwd_kernel<f16>, dim3(gblocks(total)), dim3(256), 0, ST(stream), (const f16*)x, mean, rstd, gamma, beta,
                           slope, act, total, C, HW, (f16*)y);
    else
        hipLaunchKernelGGL(bn_nchw_fwd_kernel<float>, dim3(gblocks(total)), dim3(256), 0, ST(stream), (const float*)x, mean, rstd, gamma,
                           beta, slope, act, total, C, HW, (float*)y);
    CRUSE_LAUNCH_CHECK("bn_nchw_fwd");
    return CRUSE_OK;
}

extern "C" int cruse_bn_nchw_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                 const float* beta, const float* slope, int act, int training, int N, int C, int HW,
                                 double* scratch, void* dx, float* dgamma, float* dbeta, float* dslope, int dtype, void* stream) {
    CRUSE_REQUIRE(N > 0 && C > 0 && HW > 0 && act >= 0 && act <= 3 && (act != 2 || slope), CRUSE_E_SHAPE, "bn_nchw_bwd: bad arguments");
    CRUSE_DT_CHECK("bn_nchw_bwd");
    { int rc = cruse_zero_async(scratch, 3 * (size_t)C * sizeof(double), ST(stream), "bn_nchw_bwd"); if (rc) return rc; }
    if (dtype == CRUSE_DT_F16)
        return bn_nchw_bwd_t<f16>(dy, x, mean, rstd, gamma, beta, slope, act, training, N, C, HW, scratch, dx, dgamma, dbeta, dslope, ST(stream));
    return bn_nchw_bwd_t<float>(dy, x, mean, rstd, gamma, beta, slope, act, training, N, C, HW, scratch, dx, dgamma, dbeta, dslope, ST(stream));
}

// cruse_bn_nchw_bwd that ALSO accumulates dx_sum[c] += sum of the stored dx of channel c (nullable): the bias gradient of the convolution in
// front of the BatchNorm (nn.Conv2d -> nn.BatchNorm2d -> act: mtfaa.py:166-193, cust_conv.py:15-111) without a channel-sum pass over dx
extern "C" int cruse_bn_nchw_bwd_ex(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                    const float* beta, const float* slope, int act, int training, int N, int C, int HW,
                                    double* scratch, int scratch_zeroed, int sums_replicas, void* dx, float* dgamma, float* dbeta, float* dslope,
                                    float* dx_sum, int dtype, void* stream) {
    CRUSE_REQUIRE(N > 0 && C > 0 && HW > 0 && act >= 0 && act <= 3 && (act != 2 || slope) && sums_replicas >= 0, CRUSE_E_SHAPE, "bn_nchw_bwd_ex: bad arguments");
    CRUSE_DT_CHECK("bn_nchw_bwd_ex");
    if (sums_replicas > 0) {       // scratch = [sums_replicas][4][C], ALREADY FILLED by cruse_conv2d_nchw_bnbwd: the reduce pass is skipped
        CRUSE_REQUIRE(dtype == CRUSE_DT_F16 && (long long)N * C < 65536, CRUSE_E_SHAPE, "bn_nchw_bwd_ex: delivered sums come with the f16 kernels");
        return bn_nchw_bwd_t<f16>(dy, x, mean, rstd, gamma, beta, slope, act, training, N, C, HW, scratch, dx, dgamma, dbeta, dslope, ST(stream), dx_sum, 1,
                                  sums_replicas);
    }
    if (!scratch_zeroed) { int rc = cruse_zero_async(scratch, 4 * (size_t)C * sizeof(double), ST(stream), "bn_nchw_bwd_ex"); if (rc) return rc; }
    if (dtype == CRUSE_DT_F16)
        return bn_nchw_bwd_t<f16>(dy, x, mean, rstd, gamma, beta, slope, act, training, N, C, HW, scratch, dx, dgamma, dbeta, dslope, ST(stream), dx_sum);
    return bn_nchw_bwd_t<float>(dy, x, mean, rstd, gamma, beta, slope, act, training, N, C, HW, scratch, dx, dgamma, dbeta, dslope, ST(stream), dx_sum);
}

extern "C" int cruse_bn_finalize(const double* sums, long long count, int C, float eps, float momentum,
                                 float* mean, float* rstd, float* running_mean, float* running_var, void* stream);
extern "C" int cruse_counters_add(long long* const* counters, int n, long long v, void* stream);

// Training-mode BatchNorm2d (+ activation) forward from the batch SUMS of cruse_bn_nchw_stats: mean / rstd are formed in the kernel and written to
// mean_out / rstd_out for the backward pass, the running statistics (nullable pair) and the batch counter (nullable) are updated as
// nn.BatchNorm2d does (momentum, unbiased variance) -- one launch for cruse_bn_finalize + cruse_counters_add + cruse_bn_nchw_fwd
extern "C" int cruse_bn_nchw_fwd_train(const void* x, const double* sums, int sum_replicas, float eps, float momentum, const float* gamma, const float* beta,
                                       const float* slope, int act, int N, int C, int HW, void* y, float* mean_out, float* rstd_out,
                                       float* running_mean, float* running_var, long long* num_batches_tracked, int dtype, void* stream) {
    CRUSE_REQUIRE(N > 0 && C > 0 && HW > 0 && act >= 0 && act <= 3 && (act != 2 || slope), CRUSE_E_SHAPE, "bn_nchw_fwd_train: bad arguments");
    CRUSE_REQUIRE(sums && sum_replicas >= 1 && gamma && beta && mean_out && rstd_out && (running_mean == nullptr) == (running_var == nullptr), CRUSE_E_SHAPE,
                  "bn_nchw_fwd_train: statistics / affine / output pointers");
    CRUSE_DT_CHECK("bn_nchw_fwd_train");
    const long long count = (long long)N * HW;
    if (dtype == CRUSE_DT_F16 && (long long)N * C < 65536) {
        hipLaunchKernelGGL(bn_nchw_fwd_train_f16v_kernel, dim3(cdiv(cdiv(HW, 8), 256), N * C), dim3(256), 0, ST(stream), (const f16*)x, sums, sum_replicas,
                           1.0 / (double)count, count > 1 ? (double)count / (double)(count - 1) : 1.0, eps, momentum, gamma, beta, slope, act, C, HW,
                           (f16*)y, mean_out, rstd_out, running_mean, running_var, num_batches_tracked);
        CRUSE_LAUNCH_CHECK("bn_nchw_fwd_train");
        return CRUSE_OK;
    }
    CRUSE_REQUIRE(sum_replicas == 1, CRUSE_E_SHAPE, "bn_nchw_fwd_train: replicated sums come with the f16 kernels only");
    int rc = cruse_bn_finalize(sums, count, C, eps, momentum, mean_out, rstd_out, running_mean, running_var, stream);
    if (rc) return rc;
    if (num_batches_tracked) { long long* one[1] = {num_batches_tracked}; rc = cruse_counters_add(one, 1, 1, stream); if (rc) return rc; }
    return cruse_bn_nchw_fwd(x, mean_out, rstd_out, gamma, beta, slope, act, N, C, HW, y, dtype, stream);
}

extern "C" int cruse_add_nchw(const void* a, const void* b, void* out, long long n, int dtype, void* stream) {
    CRUSE_REQUIRE(n > 0 && a && b && out, CRUSE_E_SHAPE, "add_nchw: bad arguments");
    CRUSE_DT_CHECK("add_nchw");
    if (dtype == CRUSE_DT_F16)
        hipLaunchKernelGGL(add_t_kernel<f16>, dim3(gblocks(n)), dim3(256), 0, ST(stream), (const f16*)a, (const f16*)b, (f16*)out, n);
    else
        hipLaunchKernelGGL(add_t_kernel<float>, dim3(gblocks(n)), dim3(256), 0, ST(stream), (const float*)a, (const float*)b, (float*)out, n);
    CRUSE_LAUNCH_CHECK("add_nchw");
    return CRUSE_OK;
}

extern "C" int cruse_cast_f16(const void* src, void* dst, long long n, int to_f16, void* stream) {
    CRUSE_REQUIRE(n > 0 && src && dst && src != dst, CRUSE_E_SHAPE, "cast_f16: bad arguments (out of place)");
    if (to_f16)
        hipLaunchKernelGGL((cast_t_kernel<float, f16>), dim3(gblocks(n)), dim3(256), 0, ST(stream), (const float*)src, (f16*)dst, n);
    else
        hipLaunchKernelGGL((cast_t_kernel<f16, float>), dim3(gblocks(n)), dim3(256), 0, ST(stream), (const f16*)src, (float*)dst, n);
    CRUSE_LAUNCH_CHECK("cast_f16");
    return CRUSE_OK;
}
