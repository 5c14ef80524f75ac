// Implicit-GEMM convolutions on MFMA for the CRUSE encoder / decoder / skip paths
// (nn.Conv2d / nn.ConvTranspose2d forward and backward-data, model/cruse_net.py:138-143,149-164).
//
//   Out[co, n] = sum_k W[co, k] * Patch[k, n],   n = (frame, position),  k = (tap, ci)
//
// M = output channels (16-row MFMA tiles), N = 16 positions, K = taps x Cin in steps of 32.
// A tile of 8 frames (+1 halo) is read from HBM once and copied RAW into LDS (frame-major
// [t][c][f] rows, conflict-free float4 stores); a lane's B fragment -- 8 consecutive ci of one
// tap at one position -- is gathered with 8 scalar LDS reads and converted in registers.
// Weight fragments are pre-converted once per workgroup in LDS in fragment order.
// The stride-2 transposed forms (ConvTranspose2d forward, backward-data of the stride-2
// encoder conv) are two position classes (even / odd output bins) with their own tap lists,
// so every class is a dense GEMM.  The kernel moves 2 x 640 floats per frame through HBM
// against <= 0.25 MFLOP per frame: it is HBM-bound, the MFMA time is negligible.
//
// BUILD: this source is compiled FOUR times (build.sh, -DCM_TU=k): k = 0 the dispatcher (cruse_conv_mfma_try, no kernels), k = 1 / 2 / 3
// the kernels of one precision mode each (exact f32 / split-bf16 x3 / plain bf16) -- ~100 kernels per object in parallel compile jobs
// instead of ~300 in one (100 s).
#include "common.h"
#include <stdlib.h>

#ifndef CM_TU
#define CM_TU 0
#endif

namespace cruse_cm {

constexpr int TFM = 8;                  // frames per workgroup tile
// 5 wavefronts per workgroup: every U-Net level has 8 frames x F = 10 * 2^k positions = 5 * 2^k N-tiles of 16 per tile,
// so five waves share them evenly.  (With four, the 64-channel layers -- 5 N-tiles -- left three waves idle behind the
// tile barrier while the fourth ran its second N-tile: 62 % of the N-tile slots used, SQ_WAIT_ANY 61 % of wave time.)
// NW is a template parameter: 5 where a tile has 5 or 10 N-tiles (the 64-channel level), 4 elsewhere (levels with 20+
// N-tiles lose more to the larger workgroup than they gain: measured 38 -> 45 us on the 8 -> 16 layer).
constexpr int MAXTAP = 6;
constexpr int MAXV = 8;                 // float4 per thread of a staged tile (<= 8192 floats)

struct TapClass {
    int ntaps, par;
    int dt[MAXTAP], df[MAXTAP], wk[MAXTAP];     // frame offset, input-bin offset, weight tap index (kt*3+kf)
};

// profiling (library option cm_dbg = 1): s_memtime phase sums of workgroup 0 / wave 0 of the LAST launch:
// [0] prologue (weight fragments) [1] waiting for + storing the staged tile (two barriers) [2] k-loops (gather + MFMA)
// [3] epilogues (stores) [4] tiles [5] N-tiles of wave 0 [6] total
static __device__ unsigned long long g_cm_stamps[8];

struct CMArgs {
    int tdbg, kint, swp_ok;
    int y_bf16;                                  // y too (swapped-role forms only: vector stores)
    // FUSED INPUT BatchNorm (IOB 3 / 4; forward convs of the bf16 mode, training): x is the PRE-BatchNorm tensor of the layer below and
    // the staging applies e = relu((x - mean) * rstd * gamma + beta) [+ in_add] per input channel -- mean / rstd derived by every
    // workgroup from that layer's batch sums (in_sums, in_nrep replicas); block 0 of the launch that owns the statistics also writes
    // them out and updates the running statistics (in_mean_o != null).  in_copy (nullable): a bf16 copy of the transformed rows
    // [B,T,Cin,Fin] -- the operand the weight gradients of the bf16 mode read.  The normalised tensor itself never exists in HBM.
    const double* in_sums; int in_nrep; double in_inv_count, in_unb; float in_eps, in_mom;
    const float* in_gamma; const float* in_beta; float* in_mean_o; float* in_rstd_o; float* in_rmean; float* in_rvar;
    const float* in_add; void* in_copy;
    // FUSED INPUT BatchNorm BACKWARD (IOB 5 / 6; data-gradient convs of the bf16 mode): x is the gradient wrt the OUTPUT of the BatchNorm(+ReLU)
    // (bf16), bb_y that BatchNorm's pre-BN tensor (f32); the staging forms dy = gamma rstd (g - mean g - xhat mean(g xhat)) -- the arithmetic of
    // bn_act_bwd_apply_kernel, from the same batch sums -- rounds it to bf16 for the MFMA (what the separate pass stored) and writes the bf16
    // copy the weight gradient reads (bb_copy).  Block 0 adds the parameter gradients (dgamma, dbeta; dbias in eval mode).
    const float* bb_y; const double* bb_sums; int bb_nrep; double bb_inv_count;
    const float* bb_mean; const float* bb_rstd; const float* bb_gamma; const float* bb_beta; int bb_relu, bb_training;
    void* bb_copy; float* bb_dgamma; float* bb_dbeta; float* bb_dbias;
    int x_bf16;                                  // x holds bf16 elements (a backward-only tensor stored in bf16): widened while staging
    const float* x; const float* w; const float* bias; float* y;
    int B, T, Cin, Fin, Cout, Fout;
    int S, OS, nclass, halo_lo;                  // input bin stride, output bin stride, classes, frames of halo before t0
    int nrows;                                   // staged frames = TFM + halo
    long long sco, sci;                          // weight strides of co and ci (tap index is fastest, 3*KT long)
    int act, accum;
    double* sums;                                // BatchNorm batch sums of y ([replica][2][Cout]: sum, sum of squares) or null
    // BACKWARD statistics instead (bn_y != null): y is the gradient wrt the OUTPUT of a BatchNorm(+ReLU) whose pre-BN tensor is
    // bn_y (same [B,T,Cout,Fout] layout); the sums are then those of cruse_bn_act_bwd_reduce -- sum of g and of g * xhat per
    // channel, g = y masked by the ReLU -- so that pass over (y, bn_y) is not needed
    const float* bn_y; const float* bn_mean; const float* bn_rstd; const float* bn_gamma; const float* bn_beta;
    int bn_relu;
    TapClass cls[2];
};

// LDS operand storage per precision: f32 keeps floats (8 x v_mfma_f32_16x16x4_f32 per fragment pair);
// bf16 / bf16x3 keep 1 / 2 planes of bf16 converted ONCE at staging time, so a fragment is a single
// ds_read_b128 per plane and the inner loop has no conversion VALU work.
template <int PREC> struct OpStore {
    typedef __bf16 elem;
    static constexpr int NPL = (PREC == CRUSE_PREC_BF16X3) ? 2 : 1;
};
template <> struct OpStore<CRUSE_PREC_F32> {
    typedef float elem;
    static constexpr int NPL = 1;
};

template <int PREC>
__device__ __forceinline__ void put_elem(typename OpStore<PREC>::elem* base, size_t plane_stride, size_t off, float v) {
    if constexpr (PREC == CRUSE_PREC_F32) {
        base[off] = v;
    } else if constexpr (PREC == CRUSE_PREC_BF16) {
        base[off] = (__bf16)v;
    } else {
        __bf16 h, l;
        split_bf16(v, h, l);
        base[off] = h;
        base[plane_stride + off] = l;
    }
}

template <int PREC>
__device__ __forceinline__ Frag<PREC> get_frag(const typename OpStore<PREC>::elem* base, size_t plane_stride, size_t off) {
    Frag<PREC> f;
    if constexpr (PREC == CRUSE_PREC_F32) {
        const float4 a0 = *reinterpret_cast<const float4*>(base + off);
        const float4 a1 = *reinterpret_cast<const float4*>(base + off + 4);
        f.v[0] = a0.x; f.v[1] = a0.y; f.v[2] = a0.z; f.v[3] = a0.w;
        f.v[4] = a1.x; f.v[5] = a1.y; f.v[6] = a1.z; f.v[7] = a1.w;
    } else if constexpr (PREC == CRUSE_PREC_BF16) {
        f.h = *reinterpret_cast<const bf16x8*>(base + off);
    } else {
        f.h = *reinterpret_cast<const bf16x8*>(base + off);
        f.l = *reinterpret_cast<const bf16x8*>(base + plane_stride + off);
    }
    return f;
}

// STATS: the epilogue also accumulates the per-channel sum / sum of squares of the values it stores (the
// batch statistics of the BatchNorm2d that follows every encoder / decoder conv, cruse_net.py:139,150) --
// per lane over its tiles in f32 (<= a few hundred values), then 16 lanes -> 4 waves -> one f64 atomic per
// channel and workgroup, into replica (block id mod CRUSE_BN_STAT_REPLICAS) of the sums.  This replaces a separate pass over y (65 MB, ~28 us, on the serial chain).
// NV: float4 per thread of the staged tile (the next tile's prefetch lives in registers across the whole N-tile
// loop: 5 (of 320 threads) instead of 8 is what keeps the MT = 2 / 4 statistics variants at 4 / 3 waves per SIMD).
// EPI: 0 plain epilogue; 1 forward statistics (STATS above); 2 the epilogue READS per output element -- the old value of y
// (accum) and / or the pre-BN tensor of the backward statistics (bn_y).  Those reads are issued one N-tile AHEAD, before the
// MFMAs of the current one: vmcnt retires in order, so a read issued in the epilogue itself waits for the previous N-tile's
// stores and exposes a full memory round trip per N-tile (the accumulating data gradients of the encoder ran at half the
// speed of the plain ones; with the backward statistics read that way the step got 0.15 ms SLOWER than with the separate
// reduce pass, r03).
template <int PREC, int MT, int EPI, int NV, int NW, int SWM, int IOB>
// (IOB: 1 = x holds bf16 elements, 2 = x AND y do (backward-only tensors of the bf16 mode) -- compile-time variants, so that the f32 instances stay instruction for instruction what they were:
//  this kernel sits at its register cap and a run-time dtype branch cost the f32 forward convs 10 %)
// (two 5-wave workgroups per CU need 4 wave slots on some SIMD: the 5-wave variants are held to 128 registers)
__global__ __launch_bounds__(NW * 64, NW == 5 ? 4 : 1) void conv_mfma_kernel(CMArgs a) {
    constexpr bool BBI = IOB == 5 || IOB == 6;             // fused input BatchNorm backward (y f32 / bf16)
    constexpr bool XB = IOB == 1 || IOB == 2 || BBI, YB = IOB == 2 || IOB == 6;
    constexpr bool INB = IOB == 3 || IOB == 4, INA = IOB == 4;      // fused input BatchNorm (+ added tensor)
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    __bf16* const yb = reinterpret_cast<__bf16*>(a.y);       // (YB: the output tensor's elements)
    constexpr bool STATS = EPI != 0;
    constexpr bool SW = SWM == 1;              // swapped roles, gather forms; SWM == 2: swapped roles, scatter forms, both parity classes per N-tile
    constexpr int NTHR = NW * 64;
    typedef typename OpStore<PREC>::elem elem;
    constexpr int NPL = OpStore<PREC>::NPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ks0 = (a.cls[0].ntaps * a.Cin + 31) >> 5;
    const int ks1 = a.nclass > 1 ? (a.cls[1].ntaps * a.Cin + 31) >> 5 : 0;
    const int nfrag = MT * (ks0 + ks1);
    const size_t wplane = (size_t)nfrag * 512;             // elements per weight plane
    elem* wl = reinterpret_cast<elem*>(smem_raw);          // [NPL][nfrag][64 lanes][8]  (pre-converted once)
    float* xl = reinterpret_cast<float*>(wl + NPL * wplane);   // [nrows][Cin][Fin]: RAW copy of the frame rows (f32 mode)
    // CF (the bf16 modes): the staged tile is kept CHANNEL-FASTEST and PRE-CONVERTED -- [plane][row][1 + Fin + 1 positions][Cin]
    // bf16, zero columns at both ends of a row -- so that a lane's MFMA fragment (8 consecutive channels of one tap at one position)
    // is ONE ds_read_b128 per plane and the k-loop has no conversion and no bounds logic.  (The raw [t][c][f] image made it 8 scalar
    // ds_reads + 8 (x3: 24) conversions per lane and k-step, repeated for every tap that touches an element: the k-loops were 65-80 %
    // of a tile's time, issue-bound -- cm_dbg stamps, tools/conv_probe.py.)
    constexpr bool CF = PREC != CRUSE_PREC_F32;
    // image geometry in 16-byte CHUNKS (8 channels of one position): chunk L = (row * FP + 1 + bin) * nch + octet lives at chunk
    // L ^ ((L >> 4) & (d - 1)), d = min(S * nch, 16): the 16 lanes of a ds_read_b128 group read positions S bins apart, i.e. chunks
    // d apart -- un-swizzled they fall on 16 / d distinct chunks of the 256-byte bank window (d-way conflicts); the XOR spreads
    // every aligned run of d chunks differently from one 16-chunk window to the next: 16 distinct chunks, no padding bytes
    const int FP = a.Fin + 2, nch = a.Cin >> 3, lg_nch = 31 - __clz(nch);
    const int swz_m = min(a.S * nch, 16) - 1;
    const int cplane = ((a.nrows * FP * nch + 15) & ~15) * 8;      // elements per plane of the CF image (whole 16-chunk windows: the swizzle stays inside)
    __bf16* const xc = reinterpret_cast<__bf16*>(wl + NPL * wplane);
    auto cf_off = [&](int L) -> int { return (L ^ ((L >> 4) & swz_m)) << 3; };       // chunk index -> element offset
    __shared__ int2 s_tap2[2][MAXTAP];
    __shared__ float s_bias[MT * 16];
    __shared__ float s_bnp[STATS ? 4 : 1][MT * 16];       // mean, rstd, gamma, beta of the backward-statistics form
    __shared__ float s_inp[(INB || BBI) ? (BBI ? 6 : 4) : 1][64];   // mean, rstd, gamma, beta (BBI: + mean g, mean g xhat) of the fused INPUT BatchNorm (Cin <= 64)

    const int ntile = (a.T + TFM - 1) / TFM;
    const bool tdbg = a.tdbg != 0 && blockIdx.x == 0 && wv == 0;
    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tq0 = 0, tq1 = 0, tbeg = 0;
    if (tdbg) { tbeg = tq0 = __builtin_amdgcn_s_memtime(); }

    if (tid < 2 * MAXTAP) {
        const int c = tid / MAXTAP, i = tid % MAXTAP;
        s_tap2[c][i] = CF ? make_int2((a.cls[c].dt[i] * FP + a.cls[c].df[i]) * nch, a.cls[c].df[i])
                          : make_int2(a.cls[c].dt[i] * a.Cin * a.Fin + a.cls[c].df[i], a.cls[c].df[i]);
    }
    // per-channel biases in LDS (read back per N-tile epilogue: a global load inside the N-tile loop forces an in-order
    // vmcnt wait on every older request -- the next tile's prefetch and the previous N-tile's stores -- ~2 us per N-tile;
    // holding them in registers instead costs 4 x MT VGPRs across the whole kernel)
    if (tid < MT * 16) s_bias[tid] = (a.bias && tid < a.Cout) ? a.bias[tid] : 0.f;
    if constexpr (INB) {
        if (tid < a.Cin) {                                  // bn_fin_act_fwd's arithmetic, channel by channel
            double t1 = 0.0, t2 = 0.0;
            for (int r = 0; r < a.in_nrep; ++r) { t1 += a.in_sums[(long long)r * 2 * a.Cin + tid]; t2 += a.in_sums[(long long)r * 2 * a.Cin + a.Cin + tid]; }
            const double m = t1 * a.in_inv_count;
            double var = t2 * a.in_inv_count - m * m;
            if (var < 0.0) var = 0.0;
            const float mf = (float)m, rs = (float)(1.0 / sqrt(var + (double)a.in_eps));
            s_inp[0][tid] = mf; s_inp[1][tid] = rs; s_inp[2][tid] = a.in_gamma[tid]; s_inp[3][tid] = a.in_beta[tid];
            if (blockIdx.x == 0 && a.in_mean_o != nullptr) {
                a.in_mean_o[tid] = mf; a.in_rstd_o[tid] = rs;
                if (a.in_rmean) {
                    a.in_rmean[tid] = (float)((1.0 - a.in_mom) * a.in_rmean[tid] + a.in_mom * m);
                    a.in_rvar[tid] = (float)((1.0 - a.in_mom) * a.in_rvar[tid] + a.in_mom * var * a.in_unb);
                }
            }
        }
    }
    if constexpr (BBI) {
        if (tid < a.Cin) {                                  // bn_act_bwd_apply_kernel's table, channel by channel
            double sg = 0.0, sgx = 0.0;
            for (int r = 0; r < a.bb_nrep; ++r) { sg += a.bb_sums[(size_t)r * 2 * a.Cin + tid]; sgx += a.bb_sums[(size_t)r * 2 * a.Cin + a.Cin + tid]; }
            s_inp[0][tid] = a.bb_mean[tid]; s_inp[1][tid] = a.bb_rstd[tid]; s_inp[2][tid] = a.bb_gamma[tid]; s_inp[3][tid] = a.bb_beta[tid];
            s_inp[4][tid] = a.bb_training ? (float)(sg * a.bb_inv_count) : 0.f;
            s_inp[5][tid] = a.bb_training ? (float)(sgx * a.bb_inv_count) : 0.f;
            if (blockIdx.x == 0) {
                if (a.bb_dgamma) a.bb_dgamma[tid] += (float)sgx;
                if (a.bb_dbeta) a.bb_dbeta[tid] += (float)sg;
                if (a.bb_dbias && !a.bb_training) a.bb_dbias[tid] += a.bb_gamma[tid] * a.bb_rstd[tid] * (float)sg;
            }
        }
    }
    if constexpr (STATS) {
        if (a.bn_y != nullptr && tid < MT * 16) {
            const bool ok = tid < a.Cout;
            s_bnp[0][tid] = ok ? a.bn_mean[tid] : 0.f; s_bnp[1][tid] = ok ? a.bn_rstd[tid] : 0.f;
            s_bnp[2][tid] = ok ? a.bn_gamma[tid] : 0.f; s_bnp[3][tid] = ok ? a.bn_beta[tid] : 0.f;
        }
    }
    const int ntaps0 = a.cls[0].ntaps, ntaps1 = a.cls[1].ntaps, par0 = a.cls[0].par, par1 = a.cls[1].par;
    // weight fragments: frag (c, mt, ks), lane l, element e -> W[co = mt*16 + (l&15)][k = ks*32 + (l>>4)*8 + e].
    // One (fragment, lane) pair per work item: the index arithmetic is done once per 8 elements and the 8 loads
    // are independent (Cin is a power of two, so tap/ci come from shifts).
    const int lg_cin = 31 - __clz(a.Cin);                  // Cin is a power of two (host-checked)
    // K ORDER inside a 32-deep k-step.  The lane groups g = lane >> 4 that share a tap (ngrp = min(Cin, 32) / 8 of them) take
    // INTERLEAVED channels -- element e of group g is channel cbase + e*ngrp + (g % ngrp) -- instead of 8 consecutive ones: the
    // gather below then reads, per element, channels that are ONE row (Fin floats) apart across the groups instead of EIGHT
    // (8*Fin floats = a multiple of 32 banks for Fin = 20, 40, 80: all four groups on the same 16 banks, a 4-way conflict on each
    // of the 8 reads of every k-step; PMC: 25-40 % of these kernels' LDS-active cycles).  The weight fragments are built in the
    // same order, so nothing else changes.
    const int cblk = (a.kint && !CF) ? (a.Cin < 32 ? a.Cin : 32) : 8, ngrp = cblk >> 3;       // (option cm_kint = 0: consecutive channels, the A/B switch; CF: consecutive)
    for (int it = tid; it < nfrag * 64; it += NTHR) {
        const int l = it & 63, fr = it >> 6;
        int c = 0, rem = fr;
        if (rem >= MT * ks0) { c = 1; rem -= MT * ks0; }
        const int ksn = c ? ks1 : ks0;
        const int mt = rem / ksn, ks = rem - mt * ksn;
        const int co = mt * 16 + (l & 15);
        const int k0 = ks * 32 + (l >> 4) * 8;
        const int tap = k0 >> lg_cin;                              // 8 | Cin: the 8 elements share one tap
        // element e of lane group g is channel cbase + e*ngrp + (g % ngrp) of that tap (see the gather below)
        const int ci0 = (k0 & (a.Cin - 1) & ~(cblk - 1)) + ((l >> 4) & (ngrp - 1));
        float v[8];
        const bool ok = co < a.Cout && tap < a.cls[c].ntaps;
        const float* wp = a.w + co * a.sco + ci0 * a.sci + (ok ? a.cls[c].wk[tap] : 0);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ok ? wp[e * ngrp * a.sci] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) put_elem<PREC>(wl, wplane, (size_t)it * 8 + e, v[e]);
    }
    const int rowlen = a.Cin * a.Fin;
    const int Mpos = a.Fout / a.OS;                        // positions per frame and class
    const int ntile_c = TFM * Mpos / 16;                   // N tiles per class
    const float inv_mpos = 1.0f / (float)Mpos;
    const int q8 = (lane >> 4) * 8;
    const int nvec = a.nrows * rowlen / 4;                 // rowlen % 4 == 0 (checked by the host)
    // Staging is a straight, conflict-free float4 copy of the frame rows ([t][ci][f], as in HBM).  (A first
    // version transposed to channel-fastest rows while storing: with 32..80-byte pitches those 2-byte stores
    // were 8..16-way bank conflicted and cost ~10 us per 8-frame tile.)  The MFMA B fragment -- 8 consecutive
    // ci of one tap at one position -- is gathered with 8 scalar ds_reads (lanes differ in f: conflict-light).
    // GATHER forms (output bin stride 1) run the MFMA with the PATCHES as its A operand and the weights as B ("swp"): a lane then
    // holds ONE output channel (lane & 15 of the tile) at FOUR consecutive positions (4 * (lane >> 4) .. +3 of the N-tile) --
    // consecutive output bins of one frame, written as one 16-byte (or two 8-byte) store per channel tile instead of four 4-byte
    // ones: a 4-byte-per-lane store instruction costs ~150 cycles of the epilogue (s_memtime stamps, cm_dbg = 1); forward
    // 16 -> 32: 62 -> 50 us, 32 -> 64: 87 -> 66 us.  The SCATTER forms (stride-2 output bins: no vector store) keep channels in
    // rows and positions in columns -- with the roles swapped their 4-byte stores land in 64 different rows per instruction and
    // they run 25-75 % SLOWER (convT 32 -> 16: 41 -> 72 us).
    // (SW is chosen by the host: gather form, even Mpos, 16-byte aligned tensors)
    const int vw = (Mpos & 3) == 0 ? 4 : 2;
    float s1[STATS ? MT : 1][4], s2[STATS ? MT : 1][4];
    if constexpr (STATS) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) { s1[mt][r4] = 0.f; s2[mt][r4] = 0.f; }
    }
    float4 pre[CF ? 1 : NV];
    float4 pre2[(INA && !CF) ? NV : 1];                    // the added tensor's slots (IOB 4)
    // CF staging items: item = (row r, channel octet o, bin group j of VW bins), it = (r * nch + o) * Fin/VW + j; a thread loads
    // the 8 channels of its VW bins (a channel row of Fin elements is contiguous over the lanes) and writes VW 16-byte chunks per
    // plane.  VW = 4 where Fin % 4 == 0 (16-byte loads of f32, 8-byte loads of bf16; one item per thread), else 2 (two items per
    // thread: slot 0 in the .xy halves of the registers below, slot 1 in .zw)
    float4 c4[(CF && (!XB || BBI)) ? 8 : 1], c4b[(CF && INA) ? 8 : 1];        // (BBI: c4 holds the pre-BN tensor, cb the gradient)
    uint2 cb[(CF && XB) ? 8 : 1];
    const bool vw4 = (a.Fin & 3) == 0;
    int c_src[CF ? 2 : 1], c_pos[CF ? 2 : 1], c_row[CF ? 2 : 1], c_oct[CF ? 2 : 1];
    if constexpr (CF) {
        const int vw = vw4 ? 4 : 2;
        const int jn = a.Fin / vw, per_row = jn * nch, items = a.nrows * per_row;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int it = tid + NTHR * q;
            const bool v = it < items && (q == 0 || !vw4);
            const int itc = v ? it : 0;
            const int r = itc / per_row, rem = itc - r * per_row, o = rem / jn, j = rem - o * jn;
            c_row[q] = v ? r : -1;
            c_oct[q] = o;
            c_src[q] = (r * a.Cin + 8 * o) * a.Fin + vw * j;
            c_pos[q] = r * FP + vw * j + 1;
        }
        for (int i = tid; i < NPL * cplane / 8; i += NTHR) reinterpret_cast<float4*>(xc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto prefetch = [&](int tile) {
        const int b = tile / ntile;
        const int t0 = (tile - b * ntile) * TFM;
        if constexpr (CF) {
            // unconditional loads (rows outside the clip read frame 0 of the tensor and are zeroed when stored): a branch around
            // a load makes the compiler wait for it at the join
            const long long base = ((long long)b * a.T + (t0 - a.halo_lo)) * rowlen;
            auto slot_off = [&](int q) -> long long {
                const int t = t0 - a.halo_lo + c_row[q];
                const bool ok = c_row[q] >= 0 && t >= 0 && t < a.T;
                // (a row outside the clip reads the same columns of FRAME 0 OF THE TENSOR -- not of tile row r: a tensor of fewer than
                //  nrows frames, e.g. one clip of four, ends before that)
                return ok ? base + c_src[q] : (long long)(c_src[q] - max(c_row[q], 0) * rowlen);
            };
            if (vw4) {
                const long long off = slot_off(0);
                if constexpr (XB) {
                    const __bf16* sp = reinterpret_cast<const __bf16*>(a.x) + off;
#pragma unroll
                    for (int e = 0; e < 8; ++e) cb[e] = *reinterpret_cast<const uint2*>(sp + e * a.Fin);
                    if constexpr (BBI) {
                        const float* sy = a.bb_y + off;
#pragma unroll
                        for (int e = 0; e < 8; ++e) c4[e] = *reinterpret_cast<const float4*>(sy + e * a.Fin);
                    }
                } else {
                    const float* sp = a.x + off;
#pragma unroll
                    for (int e = 0; e < 8; ++e) c4[e] = *reinterpret_cast<const float4*>(sp + e * a.Fin);
                    if constexpr (INA) {
                        const float* sp2 = a.in_add + off;
#pragma unroll
                        for (int e = 0; e < 8; ++e) c4b[e] = *reinterpret_cast<const float4*>(sp2 + e * a.Fin);
                    }
                }
            } else {
                const long long off0 = slot_off(0), off1 = slot_off(1);
                if constexpr (XB) {
                    const __bf16* sp0 = reinterpret_cast<const __bf16*>(a.x) + off0;
                    const __bf16* sp1 = reinterpret_cast<const __bf16*>(a.x) + off1;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { cb[e].x = *reinterpret_cast<const unsigned*>(sp0 + e * a.Fin); cb[e].y = *reinterpret_cast<const unsigned*>(sp1 + e * a.Fin); }
                    if constexpr (BBI) {
                        const float* sy0 = a.bb_y + off0;
                        const float* sy1 = a.bb_y + off1;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float2 u0 = *reinterpret_cast<const float2*>(sy0 + e * a.Fin), u1 = *reinterpret_cast<const float2*>(sy1 + e * a.Fin);
                            c4[e] = make_float4(u0.x, u0.y, u1.x, u1.y);
                        }
                    }
                } else {
                    const float* sp0 = a.x + off0;
                    const float* sp1 = a.x + off1;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float2 u0 = *reinterpret_cast<const float2*>(sp0 + e * a.Fin), u1 = *reinterpret_cast<const float2*>(sp1 + e * a.Fin);
                        c4[e] = make_float4(u0.x, u0.y, u1.x, u1.y);
                    }
                    if constexpr (INA) {
                        const float* sq0 = a.in_add + off0;
                        const float* sq1 = a.in_add + off1;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float2 u0 = *reinterpret_cast<const float2*>(sq0 + e * a.Fin), u1 = *reinterpret_cast<const float2*>(sq1 + e * a.Fin);
                            c4b[e] = make_float4(u0.x, u0.y, u1.x, u1.y);
                        }
                    }
                }
            }
            return;
        }
        if constexpr (XB) {                                // 8 bf16 = 16 bytes per slot (rowlen % 8 == 0, host-checked): half the slots
            const __bf16* srcb = reinterpret_cast<const __bf16*>(a.x) + ((long long)b * a.T + (t0 - a.halo_lo)) * rowlen;
#pragma unroll
            for (int q = 0; q < (NV + 1) / 2; ++q) {
                const int i = tid + NTHR * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < (nvec >> 1)) {
                    const int r = (i * 8) / rowlen;
                    const int t = t0 - a.halo_lo + r;
                    if (t >= 0 && t < a.T) v = *reinterpret_cast<const float4*>(srcb + i * 8);
                }
                pre[q] = v;
            }
            return;
        }
        const float* src = a.x + ((long long)b * a.T + (t0 - a.halo_lo)) * rowlen;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int i = tid + NTHR * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < nvec) {
                const int r = (i * 4) / rowlen;
                const int t = t0 - a.halo_lo + r;
                if (t >= 0 && t < a.T) v = *reinterpret_cast<const float4*>(src + i * 4);
            }
            pre[q] = v;
        }
        if constexpr (INA) {
            const float* src2 = a.in_add + ((long long)b * a.T + (t0 - a.halo_lo)) * rowlen;
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int i = tid + NTHR * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < nvec) {
                    const int r = (i * 4) / rowlen;
                    const int t = t0 - a.halo_lo + r;
                    if (t >= 0 && t < a.T) v = *reinterpret_cast<const float4*>(src2 + i * 4);
                }
                pre2[q] = v;
            }
        }
    };
    if ((int)blockIdx.x < a.B * ntile) prefetch(blockIdx.x);
    if (tdbg) { tq1 = __builtin_amdgcn_s_memtime(); tsum[0] += tq1 - tq0; tq0 = tq1; }
    for (int tile = blockIdx.x; tile < a.B * ntile; tile += gridDim.x) {
        const int b = tile / ntile;
        const int t0 = (tile - b * ntile) * TFM;
        const int cfo = a.Cout * a.Fout;
        const long long tbase = ((long long)b * a.T + t0) * cfo;          // this tile's first output row (uniform)
        float* const yt = a.y + tbase;
        __bf16* const ybt = yb + tbase;
        const float* const bnyt = a.bn_y + tbase;                          // (dereferenced only where a.bn_y != null)
        __syncthreads();                                   // previous tile's reads of xl are done
        if constexpr (CF) {
            __bf16* const cpy = (INB && a.in_copy) ? reinterpret_cast<__bf16*>(a.in_copy) + ((long long)b * a.T + (t0 - a.halo_lo)) * rowlen : nullptr;
            // bins [e0, e0 + nb) of the register set (0..3) belong to slot q
            auto put_slot = [&](int q, int e0, int nb) {
                if (c_row[q] < 0) return;
                const int t = t0 - a.halo_lo + c_row[q];
                const bool ok = t >= 0 && t < a.T;         // (rows outside the clip are ZERO: the padding applies to e, not to x)
                const int L0 = (c_pos[q] << lg_nch) + c_oct[q];
                if constexpr (BBI) {
                    __bf16* const bcp = a.bb_copy ? reinterpret_cast<__bf16*>(a.bb_copy) + ((long long)b * a.T + (t0 - a.halo_lo)) * rowlen : nullptr;
                    const int c0 = 8 * c_oct[q];
                    float cm[8], cr[8], cg[8], cbt[8], c1[8], c2[8];
                    *reinterpret_cast<float4*>(&cm[0]) = *reinterpret_cast<const float4*>(&s_inp[0][c0]); *reinterpret_cast<float4*>(&cm[4]) = *reinterpret_cast<const float4*>(&s_inp[0][c0 + 4]);
                    *reinterpret_cast<float4*>(&cr[0]) = *reinterpret_cast<const float4*>(&s_inp[1][c0]); *reinterpret_cast<float4*>(&cr[4]) = *reinterpret_cast<const float4*>(&s_inp[1][c0 + 4]);
                    *reinterpret_cast<float4*>(&cg[0]) = *reinterpret_cast<const float4*>(&s_inp[2][c0]); *reinterpret_cast<float4*>(&cg[4]) = *reinterpret_cast<const float4*>(&s_inp[2][c0 + 4]);
                    *reinterpret_cast<float4*>(&cbt[0]) = *reinterpret_cast<const float4*>(&s_inp[3][c0]); *reinterpret_cast<float4*>(&cbt[4]) = *reinterpret_cast<const float4*>(&s_inp[3][c0 + 4]);
                    *reinterpret_cast<float4*>(&c1[0]) = *reinterpret_cast<const float4*>(&s_inp[4][c0]); *reinterpret_cast<float4*>(&c1[4]) = *reinterpret_cast<const float4*>(&s_inp[4][c0 + 4]);
                    *reinterpret_cast<float4*>(&c2[0]) = *reinterpret_cast<const float4*>(&s_inp[5][c0]); *reinterpret_cast<float4*>(&c2[4]) = *reinterpret_cast<const float4*>(&s_inp[5][c0 + 4]);
                    bf16x8 hs[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (e < e0 || e >= e0 + nb) continue;
                        bf16x8 h;
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const unsigned dw = (e & 2) ? cb[c].y : cb[c].x;
                            const float dd = (e & 1) ? __uint_as_float(dw & 0xffff0000u) : __uint_as_float(dw << 16);
                            const float in = e == 0 ? c4[c].x : e == 1 ? c4[c].y : e == 2 ? c4[c].z : c4[c].w;
                            const float xh = (in - cm[c]) * cr[c];
                            float gr = dd;
                            if (a.bb_relu && !(xh * cg[c] + cbt[c] > 0.f)) gr = 0.f;
                            const float o = cg[c] * cr[c] * (gr - c1[c] - xh * c2[c]);
                            h[c] = (__bf16)(ok ? o : 0.f);
                        }
                        hs[e] = h;
                        *reinterpret_cast<bf16x8*>(xc + cf_off(L0 + ((e - e0) << lg_nch))) = h;
                    }
                    if (bcp != nullptr && ok && c_row[q] >= a.halo_lo) {          // own rows only: the halo belongs to the tile before
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            if (nb == 4) {
                                bf16x4_ p4;
                                p4[0] = hs[0][c]; p4[1] = hs[1][c]; p4[2] = hs[2][c]; p4[3] = hs[3][c];
                                *reinterpret_cast<bf16x4_*>(bcp + c_src[q] + c * a.Fin) = p4;
                            } else {
                                bf16x2_ p2;
                                p2[0] = e0 == 0 ? hs[0][c] : hs[2][c]; p2[1] = e0 == 0 ? hs[1][c] : hs[3][c];
                                *reinterpret_cast<bf16x2_*>(bcp + c_src[q] + c * a.Fin) = p2;
                            }
                        }
                    }
                } else if constexpr (XB) {
                    // dword (e >> 1) of channel c holds bins e (low half) and e + 1: per bin one 16-byte run of 8 channels
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (e < e0 || e >= e0 + nb) continue;
                        u32x4_ w;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const unsigned lo = (e & 2) ? cb[2 * i].y : cb[2 * i].x, hi = (e & 2) ? cb[2 * i + 1].y : cb[2 * i + 1].x;
                            w[i] = ok ? __builtin_amdgcn_perm(hi, lo, (e & 1) ? 0x07060302u : 0x05040100u) : 0u;
                        }
                        *reinterpret_cast<u32x4_*>(xc + cf_off(L0 + ((e - e0) << lg_nch))) = w;
                    }
                } else {
                    float cm[8], cr[8], cg[8], cbt[8];
                    if constexpr (INB) {
                        const int c0 = 8 * c_oct[q];
                        *reinterpret_cast<float4*>(&cm[0]) = *reinterpret_cast<const float4*>(&s_inp[0][c0]); *reinterpret_cast<float4*>(&cm[4]) = *reinterpret_cast<const float4*>(&s_inp[0][c0 + 4]);
                        *reinterpret_cast<float4*>(&cr[0]) = *reinterpret_cast<const float4*>(&s_inp[1][c0]); *reinterpret_cast<float4*>(&cr[4]) = *reinterpret_cast<const float4*>(&s_inp[1][c0 + 4]);
                        *reinterpret_cast<float4*>(&cg[0]) = *reinterpret_cast<const float4*>(&s_inp[2][c0]); *reinterpret_cast<float4*>(&cg[4]) = *reinterpret_cast<const float4*>(&s_inp[2][c0 + 4]);
                        *reinterpret_cast<float4*>(&cbt[0]) = *reinterpret_cast<const float4*>(&s_inp[3][c0]); *reinterpret_cast<float4*>(&cbt[4]) = *reinterpret_cast<const float4*>(&s_inp[3][c0 + 4]);
                    }
                    bf16x8 hs[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (e < e0 || e >= e0 + nb) continue;
                        float v[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const float xin = e == 0 ? c4[c].x : e == 1 ? c4[c].y : e == 2 ? c4[c].z : c4[c].w;
                            float tv = xin;
                            if constexpr (INB) {
                                tv = (xin - cm[c]) * cr[c] * cg[c] + cbt[c];
                                tv = fmaxf(tv, 0.f);
                                if constexpr (INA) tv += e == 0 ? c4b[c].x : e == 1 ? c4b[c].y : e == 2 ? c4b[c].z : c4b[c].w;
                            }
                            v[c] = ok ? tv : 0.f;
                        }
                        bf16x8 h;
#pragma unroll
                        for (int c = 0; c < 8; ++c) h[c] = (__bf16)v[c];
                        hs[e] = h;
                        const int off = cf_off(L0 + ((e - e0) << lg_nch));
                        *reinterpret_cast<bf16x8*>(xc + off) = h;
                        if constexpr (NPL == 2) {
                            bf16x8 l;
#pragma unroll
                            for (int c = 0; c < 8; ++c) l[c] = (__bf16)(v[c] - (float)h[c]);
                            *reinterpret_cast<bf16x8*>(xc + cplane + off) = l;
                        }
                    }
                    if constexpr (INB) {
                        if (cpy != nullptr && ok && c_row[q] >= a.halo_lo) {       // own rows only: the halo belongs to the tile before
                            // bf16 copy in the tensor's own [c][f] order: nb bins per channel
#pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                if (nb == 4) {
                                    bf16x4_ p4;
                                    p4[0] = hs[0][c]; p4[1] = hs[1][c]; p4[2] = hs[2][c]; p4[3] = hs[3][c];
                                    *reinterpret_cast<bf16x4_*>(cpy + c_src[q] + c * a.Fin) = p4;
                                } else {
                                    bf16x2_ p2;
                                    p2[0] = e0 == 0 ? hs[0][c] : hs[2][c]; p2[1] = e0 == 0 ? hs[1][c] : hs[3][c];
                                    *reinterpret_cast<bf16x2_*>(cpy + c_src[q] + c * a.Fin) = p2;
                                }
                            }
                        }
                    }
                }
            };
            if (vw4) put_slot(0, 0, 4);
            else { put_slot(0, 0, 2); put_slot(1, 2, 2); }
        } else if constexpr (XB) {
#pragma unroll
            for (int q = 0; q < (NV + 1) / 2; ++q) {
                const int i = tid + NTHR * q;
                if (i < (nvec >> 1)) {
                    const unsigned w0 = __float_as_uint(pre[q].x), w1 = __float_as_uint(pre[q].y), w2 = __float_as_uint(pre[q].z),
                                   w3 = __float_as_uint(pre[q].w);
                    *reinterpret_cast<float4*>(xl + i * 8) = make_float4(__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u),
                                                                         __uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u));
                    *reinterpret_cast<float4*>(xl + i * 8 + 4) = make_float4(__uint_as_float(w2 << 16), __uint_as_float(w2 & 0xffff0000u),
                                                                             __uint_as_float(w3 << 16), __uint_as_float(w3 & 0xffff0000u));
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int i = tid + NTHR * q;
                if (i < nvec) *reinterpret_cast<float4*>(xl + i * 4) = pre[q];
            }
        }
        __syncthreads();
        // EPI == 2: old output values / pre-BN values of this lane's MT x 4 outputs of an N-tile
        float old_c[EPI == 2 ? MT : 1][4], by_c[EPI == 2 ? MT : 1][4], old_n[EPI == 2 ? MT : 1][4], by_n[EPI == 2 ? MT : 1][4];
        // swp: output addressing of an N-tile for this lane -- positions 4*(lane>>4) + e of channel (lane & 15) + 16*mt; the
        // positions of a vector (4, or 2 + 2) are consecutive bins of one frame
        // (32-bit element offsets inside the tile's output rows -- yt / ybt / bnyt below: the 64-bit products per N-tile were a third of
        //  the ~165 instructions between an N-tile's last MFMA and the next one's first, which is 70 % of an N-tile's instruction count)
        auto out_pos = [&](int nt_, int (&off)[2], bool (&okt)[2]) {
            const int p0 = nt_ * 16 + (lane >> 4) * 4;     // (one class in the gather forms)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h == 1 && vw == 4) { okt[1] = okt[0]; off[1] = off[0] + 2; continue; }     // (one 4-bin vector: the same frame)
                const int p_ = p0 + 2 * h;
                const int tl_ = (int)(((float)p_ + 0.5f) * inv_mpos);
                const int m_ = p_ - tl_ * Mpos;
                okt[h] = t0 + tl_ < a.T;
                off[h] = tl_ * cfo + (lane & 15) * a.Fout + m_ + par0;
            }
        };
        auto aux_load_swp = [&](int nt_, float (&o)[EPI == 2 ? MT : 1][4], float (&y_)[EPI == 2 ? MT : 1][4]) {
            int off[2];
            bool okt[2];
            out_pos(nt_, off, okt);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bool okc = mt * 16 + (lane & 15) < a.Cout;
                const int cm = mt * 16 * a.Fout;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bool ok = okc && okt[h];
                    const float2 z2 = make_float2(0.f, 0.f);
                    float2 ov = z2;
                    if (ok && a.accum) {
                        if constexpr (YB) { const bf16x2_ t2 = *reinterpret_cast<const bf16x2_*>(ybt + off[h] + cm); ov = make_float2((float)t2[0], (float)t2[1]); }
                        else ov = *reinterpret_cast<const float2*>(yt + off[h] + cm);
                    }
                    const float2 yv = (ok && a.bn_y != nullptr) ? *reinterpret_cast<const float2*>(bnyt + off[h] + cm) : z2;
                    o[mt][2 * h] = ov.x; o[mt][2 * h + 1] = ov.y;
                    y_[mt][2 * h] = yv.x; y_[mt][2 * h + 1] = yv.y;
                }
            }
        };
        auto aux_load_std = [&](int nt_, float (&o)[EPI == 2 ? MT : 1][4], float (&y_)[EPI == 2 ? MT : 1][4]) {
            const int c_ = nt_ >= ntile_c ? 1 : 0;
            const int p_ = (nt_ - c_ * ntile_c) * 16 + (lane & 15);
            const int tl_ = (int)(((float)p_ + 0.5f) * inv_mpos);
            const int m_ = p_ - tl_ * Mpos;
            const int t_ = t0 + tl_;
            const long long off = (((long long)b * a.T + t_) * a.Cout + (lane >> 4) * 4) * a.Fout + a.OS * m_ + (c_ ? par1 : par0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const bool ok = t_ < a.T && mt * 16 + (lane >> 4) * 4 + r4 < a.Cout;
                    const long long e = off + (long long)(mt * 16 + r4) * a.Fout;
                    o[mt][r4] = (ok && a.accum) ? a.y[e] : 0.f;
                    y_[mt][r4] = (ok && a.bn_y != nullptr) ? a.bn_y[e] : 0.f;
                }
        };
        auto aux_load = [&](int nt_, float (&o)[EPI == 2 ? MT : 1][4], float (&y_)[EPI == 2 ? MT : 1][4]) {
            if constexpr (SW) aux_load_swp(nt_, o, y_); else aux_load_std(nt_, o, y_);
        };
        if constexpr (EPI == 2 && SWM != 2) {              // (ahead of the next tile's prefetch in the in-order queue)
            if (wv < a.nclass * ntile_c) aux_load(wv, old_c, by_c);
        }
        if (tile + (int)gridDim.x < a.B * ntile) prefetch(tile + gridDim.x);
        if (tdbg) { tq1 = __builtin_amdgcn_s_memtime(); tsum[1] += tq1 - tq0; tq0 = tq1; tsum[4] += 1; }

        // N-tile loop, kept free of integer divisions (Cin is a power of two, positions via a float reciprocal)
        // and of global loads; one 64-bit base address per N-tile.  (Keeping several N-tiles in flight per wave
        // was measured and is slower: 78 vs 57 us on the 8->16 layer.)
        if constexpr (SWM == 2) {
            // SCATTER forms with swapped roles: an N-tile is 16 positions of BOTH parity classes (two k-loops, two accumulator
            // sets), so a lane holds its channel at output bins 2m .. 2m+3 for each of its two position pairs: two 16-byte stores
            // per channel tile where the class-by-class form writes eight 4-byte ones with a stride of two bins.  The epilogue's
            // reads (old value, pre-BN value) are issued at the START of the N-tile: two k-loops later they have long arrived.
            for (int nt = wv; nt < ntile_c; nt += NW) {
                const int p = nt * 16 + (lane & 15);
                const int tl = (int)(((float)p + 0.5f) * inv_mpos);
                const int m = p - tl * Mpos;
                const int rowbase = (tl + a.halo_lo) * rowlen + a.S * m;
                const int rowbase_cf = ((tl + a.halo_lo) * FP + a.S * m + 1) << lg_nch;
                int off[2];
                bool okt[2];
                {
                    const int p0 = nt * 16 + (lane >> 4) * 4;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int p_ = p0 + 2 * h;
                        const int tl_ = (int)(((float)p_ + 0.5f) * inv_mpos);
                        const int m_ = p_ - tl_ * Mpos;
                        okt[h] = t0 + tl_ < a.T;
                        off[h] = tl_ * cfo + (lane & 15) * a.Fout + 2 * m_;
                    }
                }
                float4 oldv[EPI == 2 ? MT : 1][2], byv[EPI == 2 ? MT : 1][2];
                if constexpr (EPI == 2) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const bool okc = mt * 16 + (lane & 15) < a.Cout;
                        const int cm = mt * 16 * a.Fout;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                            oldv[mt][h] = z4;
                            if (okc && okt[h] && a.accum) {
                                if constexpr (YB) {
                                    const bf16x4_ t4 = *reinterpret_cast<const bf16x4_*>(ybt + off[h] + cm);
                                    oldv[mt][h] = make_float4((float)t4[0], (float)t4[1], (float)t4[2], (float)t4[3]);
                                } else {
                                    oldv[mt][h] = *reinterpret_cast<const float4*>(yt + off[h] + cm);
                                }
                            }
                            byv[mt][h] = (okc && okt[h] && a.bn_y != nullptr) ? *reinterpret_cast<const float4*>(bnyt + off[h] + cm) : z4;
                        }
                    }
                }
                f32x4 acc2[2][MT];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc2[c][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    const int ksn = c ? ks1 : ks0, fbase = c ? MT * ks0 : 0, ntaps = c ? ntaps1 : ntaps0;
                    for (int ks = 0; ks < ksn; ++ks) {
                        const int k = ks * 32 + q8;
                        int tap = k >> lg_cin;
                        const int ci0 = (k & (a.Cin - 1) & ~(cblk - 1)) + ((lane >> 4) & (ngrp - 1));
                        if (tap >= ntaps) tap = ntaps - 1;
                        const int2 tp = s_tap2[c][tap];
                        Frag<PREC> fb;
                        if constexpr (CF) {
                            const __bf16* pb = xc + cf_off(rowbase_cf + tp.x + ((k & (a.Cin - 1)) >> 3));
                            fb.h = *reinterpret_cast<const bf16x8*>(pb);
                            if constexpr (NPL == 2) fb.l = *reinterpret_cast<const bf16x8*>(pb + cplane);
                        } else {
                        const int f = a.S * m + tp.y;
                        float bv[8];
                        if (f >= 0 && f < a.Fin) {
                            const float* pb = xl + rowbase + tp.x + ci0 * a.Fin;
                            const int cstep = ngrp * a.Fin;
#pragma unroll
                            for (int e = 0; e < 8; ++e) bv[e] = pb[e * cstep];
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) bv[e] = 0.f;
                        }
                        fb.set(bv);
                        }
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const Frag<PREC> fa = get_frag<PREC>(wl, wplane, ((size_t)(fbase + mt * ksn + ks) * 64 + lane) * 8);
                            acc2[c][mt] = mma_t<PREC>(fb, fa, acc2[c][mt]);
                        }
                    }
                }
                if (tdbg) { tq1 = __builtin_amdgcn_s_memtime(); tsum[2] += tq1 - tq0; tq0 = tq1; tsum[5] += 1; }
                // class c writes output bins of parity cls[c].par: even bins from the class with par 0
                const int ce = par0 == 0 ? 0 : 1, cod = 1 - ce;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int co = mt * 16 + (lane & 15);
                    if (co < a.Cout) {
                        const float bq = s_bias[co];
                        const int cm = mt * 16 * a.Fout;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            float v[4] = {acc2[ce][mt][2 * h] + bq, acc2[cod][mt][2 * h] + bq, acc2[ce][mt][2 * h + 1] + bq, acc2[cod][mt][2 * h + 1] + bq};
                            if constexpr (EPI == 2) { v[0] += oldv[mt][h].x; v[1] += oldv[mt][h].y; v[2] += oldv[mt][h].z; v[3] += oldv[mt][h].w; }
                            if (okt[h]) {
                                if constexpr (YB) {
                                    bf16x4_ t4;
                                    t4[0] = (__bf16)v[0]; t4[1] = (__bf16)v[1]; t4[2] = (__bf16)v[2]; t4[3] = (__bf16)v[3];
                                    *reinterpret_cast<bf16x4_*>(ybt + off[h] + cm) = t4;
                                } else {
                                    *reinterpret_cast<float4*>(yt + off[h] + cm) = make_float4(v[0], v[1], v[2], v[3]);
                                }
                                if constexpr (EPI == 1) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) { s1[mt][0] += v[e]; s2[mt][0] += v[e] * v[e]; }
                                }
                                if constexpr (EPI == 2) {
                                    if (a.bn_y != nullptr) {
                                        const float mu = s_bnp[0][co], rs_ = s_bnp[1][co], ga = s_bnp[2][co], be = s_bnp[3][co];
                                        const float yv[4] = {byv[mt][h].x, byv[mt][h].y, byv[mt][h].z, byv[mt][h].w};
#pragma unroll
                                        for (int e = 0; e < 4; ++e) {
                                            const float xh = (yv[e] - mu) * rs_;
                                            const float gr = (a.bn_relu && !(xh * ga + be > 0.f)) ? 0.f : v[e];
                                            s1[mt][0] += gr; s2[mt][0] += gr * xh;
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
                if (tdbg) { tq1 = __builtin_amdgcn_s_memtime(); tsum[3] += tq1 - tq0; tq0 = tq1; }
            }
        } else
        for (int nt = wv; nt < a.nclass * ntile_c; nt += NW) {
            if constexpr (EPI == 2) {
                if (nt + NW < a.nclass * ntile_c) aux_load(nt + NW, old_n, by_n);
            }
            const int c = nt >= ntile_c ? 1 : 0;
            const int p = (nt - c * ntile_c) * 16 + (lane & 15);
            const int tl = (int)(((float)p + 0.5f) * inv_mpos);
            const int m = p - tl * Mpos;
            const int ksn = c ? ks1 : ks0;
            const int fbase = c ? MT * ks0 : 0;
            const int ntaps = c ? ntaps1 : ntaps0;
            f32x4 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int rowbase = (tl + a.halo_lo) * rowlen + a.S * m;
            const int rowbase_cf = ((tl + a.halo_lo) * FP + a.S * m + 1) << lg_nch;
            for (int ks = 0; ks < ksn; ++ks) {
                const int k = ks * 32 + q8;
                int tap = k >> lg_cin;
                const int ci0 = (k & (a.Cin - 1) & ~(cblk - 1)) + ((lane >> 4) & (ngrp - 1));
                if (tap >= ntaps) tap = ntaps - 1;        // zero weights there; keep the address valid
                const int2 tp = s_tap2[c][tap];           // x = dt*rowlen + df, y = df
                Frag<PREC> fb;
                if constexpr (CF) {
                    const __bf16* pb = xc + cf_off(rowbase_cf + tp.x + ((k & (a.Cin - 1)) >> 3));
                    fb.h = *reinterpret_cast<const bf16x8*>(pb);
                    if constexpr (NPL == 2) fb.l = *reinterpret_cast<const bf16x8*>(pb + cplane);
                } else {
                const int f = a.S * m + tp.y;
                float bv[8];
                if (f >= 0 && f < a.Fin) {
                    const float* pb = xl + rowbase + tp.x + ci0 * a.Fin;
                    const int cstep = ngrp * a.Fin;
#pragma unroll
                    for (int e = 0; e < 8; ++e) bv[e] = pb[e * cstep];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bv[e] = 0.f;
                }
                fb.set(bv);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const Frag<PREC> fa = get_frag<PREC>(wl, wplane, ((size_t)(fbase + mt * ksn + ks) * 64 + lane) * 8);
                    if constexpr (SW) acc[mt] = mma_t<PREC>(fb, fa, acc[mt]); else acc[mt] = mma(fa, fb, acc[mt]);     // swp: rows = positions, columns = channels
                }
            }
            if (tdbg) { tq1 = __builtin_amdgcn_s_memtime(); tsum[2] += tq1 - tq0; tq0 = tq1; tsum[5] += 1; }
            if constexpr (SW) {
                int off[2];
                bool okt[2];
                out_pos(nt, off, okt);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int co = mt * 16 + (lane & 15);
                    if (co < a.Cout) {
                        const float bq = s_bias[co];
                        const int cm = mt * 16 * a.Fout;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[mt][e] + bq;
                            if constexpr (EPI == 2) v[e] += old_c[mt][e];                       // (0 unless accum)
                            else if (a.act == 1) v[e] = sigmoid_acc(v[e]);
                        }
                        if constexpr (YB) {
                            bf16x2_ t0, t1;
                            t0[0] = (__bf16)v[0]; t0[1] = (__bf16)v[1]; t1[0] = (__bf16)v[2]; t1[1] = (__bf16)v[3];
                            if (vw == 4) {
                                bf16x4_ t4;
                                t4[0] = t0[0]; t4[1] = t0[1]; t4[2] = t1[0]; t4[3] = t1[1];
                                if (okt[0]) *reinterpret_cast<bf16x4_*>(ybt + off[0] + cm) = t4;
                            } else {
                                if (okt[0]) *reinterpret_cast<bf16x2_*>(ybt + off[0] + cm) = t0;
                                if (okt[1]) *reinterpret_cast<bf16x2_*>(ybt + off[1] + cm) = t1;
                            }
                        } else if (vw == 4) {
                            if (okt[0]) *reinterpret_cast<float4*>(yt + off[0] + cm) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
                            if (okt[0]) *reinterpret_cast<float2*>(yt + off[0] + cm) = make_float2(v[0], v[1]);
                            if (okt[1]) *reinterpret_cast<float2*>(yt + off[1] + cm) = make_float2(v[2], v[3]);
                        }
                        if constexpr (EPI == 1) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (okt[e >> 1]) { s1[mt][0] += v[e]; s2[mt][0] += v[e] * v[e]; }
                        }
                        if constexpr (EPI == 2) {
                            if (a.bn_y != nullptr) {
                                const float mu = s_bnp[0][co], rs_ = s_bnp[1][co], ga = s_bnp[2][co], be = s_bnp[3][co];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if (okt[e >> 1]) {
                                        const float xh = (by_c[mt][e] - mu) * rs_;
                                        const float gr = (a.bn_relu && !(xh * ga + be > 0.f)) ? 0.f : v[e];
                                        s1[mt][0] += gr; s2[mt][0] += gr * xh;
                                    }
                                }
                            }
                        }
                    }
                }
            }
            const int t = t0 + tl;
            if (!SW && t < a.T) {
                const int fo = a.OS * m + (c ? par1 : par0);
                float* yb = a.y + (((long long)b * a.T + t) * a.Cout + (lane >> 4) * 4) * a.Fout + fo;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float4 bq = *reinterpret_cast<const float4*>(&s_bias[mt * 16 + (lane >> 4) * 4]);
                    const float bqv[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int co = mt * 16 + (lane >> 4) * 4 + r4;
                        if (co < a.Cout) {
                            float* yp = yb + (mt * 16 + r4) * a.Fout;
                            float v = acc[mt][r4] + bqv[r4];
                            if constexpr (EPI == 2) {
                                v += old_c[mt][r4];                               // (0 unless accum)
                                *yp = v;
                                if (a.bn_y != nullptr) {
                                    const float xh = (by_c[mt][r4] - s_bnp[0][co]) * s_bnp[1][co];
                                    const float gr = (a.bn_relu && !(xh * s_bnp[2][co] + s_bnp[3][co] > 0.f)) ? 0.f : v;
                                    s1[mt][r4] += gr; s2[mt][r4] += gr * xh;
                                }
                            } else {
                                if (a.accum) v += *yp;
                                else if (a.act == 1) v = sigmoid_acc(v);
                                *yp = v;
                                if constexpr (EPI == 1) { s1[mt][r4] += v; s2[mt][r4] += v * v; }
                            }
                        }
                    }
                }
            }
            if constexpr (EPI == 2) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) { old_c[mt][r4] = old_n[mt][r4]; by_c[mt][r4] = by_n[mt][r4]; }
            }
            if (tdbg) { tq1 = __builtin_amdgcn_s_memtime(); tsum[3] += tq1 - tq0; tq0 = tq1; }
        }
    }
    if (tdbg && lane == 0) {
        for (int i = 0; i < 6; ++i) g_cm_stamps[i] = tsum[i];
        g_cm_stamps[6] = __builtin_amdgcn_s_memtime() - tbeg;
    }
    if constexpr (STATS) {
        if (a.sums == nullptr) return;                     // (EPI 2 without backward statistics: accumulate only)
        __shared__ float s_red[NW][2][MT * 16];
        if constexpr (SWM != 0) {                          // a lane's channel is (lane & 15): the four lane groups meet
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float u = s1[mt][0], u2 = s2[mt][0];
                u += __shfl_xor(u, 16, 64); u2 += __shfl_xor(u2, 16, 64);
                u += __shfl_xor(u, 32, 64); u2 += __shfl_xor(u2, 32, 64);
                if (lane < 16) { s_red[wv][0][mt * 16 + lane] = u; s_red[wv][1][mt * 16 + lane] = u2; }
            }
        } else
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float u = s1[mt][r4], u2 = s2[mt][r4];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { u += __shfl_xor(u, o, 64); u2 += __shfl_xor(u2, o, 64); }
                if ((lane & 15) == 0) {
                    const int co = mt * 16 + (lane >> 4) * 4 + r4;
                    s_red[wv][0][co] = u; s_red[wv][1][co] = u2;
                }
            }
        __syncthreads();
        if (tid < 2 * MT * 16) {
            const int which = tid / (MT * 16), co = tid - which * (MT * 16);
            if (co < a.Cout) {
                float u = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) u += s_red[w][which][co];
                atomicAdd(a.sums + (size_t)(blockIdx.x % CRUSE_BN_STAT_REPLICAS) * 2 * a.Cout + which * a.Cout + co, (double)u);
            }
        }
    }
}

template <int PREC>
int launch_mt(const CMArgs& a, int grid, size_t lds, int nw, hipStream_t s) {
    const int mt = (a.Cout + 15) / 16;
    const int nthr = nw * 64;
    const int nv = (a.nrows * a.Cin * a.Fin / 4 + nthr - 1) / nthr;    // float4 per thread of one staged tile
    // swapped MFMA roles (vector stores): gather forms with an even number of positions per frame and 16-byte aligned tensors
    const bool al16 = ((reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.bn_y)) & 15) == 0;
    int swm = 0;
    if (a.swp_ok != 0 && al16 && ((a.Fout / a.OS) & 1) == 0) {
        if (a.OS == 1) swm = 1;
        else if (a.nclass == 2 && a.cls[0].par + a.cls[1].par == 1 && (a.sums != nullptr || a.accum || a.bn_y != nullptr) && !a.act)   // (a bf16 output is always one of these)
            swm = 2;                             // (plain-epilogue scatter launches keep the class-by-class form)
    }
    int rc;
#define CM_LAUNCH5(MTV, STV, NVV, NWV, SWV, XBV)                                                           \
    do {                                                                                                   \
        if ((rc = cruse_ensure_dyn_lds(reinterpret_cast<const void*>(conv_mfma_kernel<PREC, MTV, STV, NVV, NWV, SWV, XBV>), lds, "conv_mfma"))) return rc; \
        hipLaunchKernelGGL((conv_mfma_kernel<PREC, MTV, STV, NVV, NWV, SWV, XBV>), dim3(grid), dim3(NWV * 64), lds, s, a); \
    } while (0)
    // bf16 inputs: the data-gradient forms of the bf16 mode only (EPI 0 / 2, PREC bf16)
#define CM_LAUNCH4(MTV, STV, NVV, NWV, SWV)                                                                \
    do {                                                                                                   \
        if constexpr (STV != 2 && PREC == CRUSE_PREC_BF16X3) {      /* forward convs of the bf16 mode: fused input BatchNorm */ \
            if (a.in_sums != nullptr && a.in_add != nullptr) CM_LAUNCH5(MTV, STV, NVV, NWV, SWV, 4);       \
            else if (a.in_sums != nullptr) CM_LAUNCH5(MTV, STV, NVV, NWV, SWV, 3);                         \
            else CM_LAUNCH5(MTV, STV, NVV, NWV, SWV, 0);                                                   \
        } else if constexpr (STV != 1 && PREC == CRUSE_PREC_BF16) {                                        \
            if (a.bb_y != nullptr) {                                                                       \
                if constexpr (NWV != 4 || MTV > 2) { cruse_set_error("conv_mfma: the fused BatchNorm-backward input is a 4-wave, <= 32-row form"); return CRUSE_E_SHAPE; } \
                else if (a.y_bf16) {                                                                       \
                    if constexpr (SWV != 0) CM_LAUNCH5(MTV, STV, NVV, NWV, SWV, 6);                        \
                    else { cruse_set_error("conv_mfma: a bf16 output needs the swapped-role (vector-store) forms"); return CRUSE_E_DTYPE; } \
                } else CM_LAUNCH5(MTV, STV, NVV, NWV, SWV, 5);                                             \
            } else if (a.x_bf16 && a.y_bf16) {                                                             \
                if constexpr (SWV != 0) CM_LAUNCH5(MTV, STV, NVV, NWV, SWV, 2);                            \
                else { cruse_set_error("conv_mfma: a bf16 output needs the swapped-role (vector-store) forms"); return CRUSE_E_DTYPE; } \
            } else if (a.x_bf16) CM_LAUNCH5(MTV, STV, NVV, NWV, SWV, 1);                                   \
            else CM_LAUNCH5(MTV, STV, NVV, NWV, SWV, 0);                                                   \
        } else {                                                                                           \
            CM_LAUNCH5(MTV, STV, NVV, NWV, SWV, 0);                                                        \
        }                                                                                                  \
    } while (0)
#define CM_LAUNCH3(MTV, STV, NVV, NWV)                                                                     \
    do {                                                                                                   \
        if (swm == 1) CM_LAUNCH4(MTV, STV, NVV, NWV, 1);                                                   \
        else if (swm == 2) CM_LAUNCH4(MTV, STV, NVV, NWV, (STV == 0 ? 0 : 2));                             \
        else CM_LAUNCH4(MTV, STV, NVV, NWV, 0);                                                            \
    } while (0)
#define CM_LAUNCH2(MTV, STV, NVV)                                                                          \
    do {                                                                                                   \
        if constexpr (PREC == CRUSE_PREC_F32) CM_LAUNCH3(MTV, STV, NVV, 4);     /* (the exact-f32 gate mode: four waves everywhere) */ \
        else if (nw == 5) CM_LAUNCH3(MTV, STV, NVV, 5);                                                    \
        else CM_LAUNCH3(MTV, STV, NVV, 4);                                                                 \
    } while (0)
#define CM_LAUNCH1(MTV, STV)                                                                               \
    do {                                                                                                   \
        if constexpr (PREC != CRUSE_PREC_F32) CM_LAUNCH2(MTV, STV, 5);     /* (the channel-fastest staging has no NV) */ \
        else if (nv <= 5) CM_LAUNCH2(MTV, STV, 5);                                                         \
        else if (nv <= 6) CM_LAUNCH2(MTV, STV, 6);                                                         \
        else CM_LAUNCH2(MTV, STV, MAXV);                                                                   \
    } while (0)
#define CM_LAUNCH(MTV)                                                                                     \
    do {                                                                                                   \
        if (a.accum || a.bn_y) CM_LAUNCH1(MTV, 2);                                                         \
        else if (a.sums) CM_LAUNCH1(MTV, 1);                                                               \
        else CM_LAUNCH1(MTV, 0);                                                                           \
    } while (0)
    if (mt <= 1) CM_LAUNCH(1);
    else if (mt <= 2) CM_LAUNCH(2);
    else CM_LAUNCH(4);
#undef CM_LAUNCH
#undef CM_LAUNCH1
#undef CM_LAUNCH2
#undef CM_LAUNCH3
#undef CM_LAUNCH4
#undef CM_LAUNCH5
    return CRUSE_OK;
}

}  // namespace cruse_cm
using namespace cruse_cm;

// per-precision launchers and their phase-stamp readers: one translation unit each (CM_TU)
int cruse_cm_launch_f32(const CMArgs& a, int grid, size_t lds, int nw, hipStream_t s);
int cruse_cm_launch_x3(const CMArgs& a, int grid, size_t lds, int nw, hipStream_t s);
int cruse_cm_launch_bf16(const CMArgs& a, int grid, size_t lds, int nw, hipStream_t s);
int cruse_cm_stamps_f32(unsigned long long* out8);
int cruse_cm_stamps_x3(unsigned long long* out8);
int cruse_cm_stamps_bf16(unsigned long long* out8);
#define CM_DEFINE_TU(SUFFIX, PRECV)                                                                                   \
    int cruse_cm_launch_##SUFFIX(const CMArgs& a, int grid, size_t lds, int nw, hipStream_t s) { return launch_mt<PRECV>(a, grid, lds, nw, s); } \
    int cruse_cm_stamps_##SUFFIX(unsigned long long* out8) {                                                          \
        return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_cm_stamps), 8 * sizeof(unsigned long long)) == hipSuccess ? CRUSE_OK : CRUSE_E_HIP; \
    }
#if CM_TU == 1
CM_DEFINE_TU(f32, CRUSE_PREC_F32)
#elif CM_TU == 2
CM_DEFINE_TU(x3, CRUSE_PREC_BF16X3)
#elif CM_TU == 3
CM_DEFINE_TU(bf16, CRUSE_PREC_BF16)
#else
static int g_cm_last_prec = CRUSE_PREC_F32;

// Returns 1 if the MFMA path handled the call, 0 if the shape is not eligible (caller falls back
// to the VALU kernel), < 0 on error.
int cruse_conv_mfma_try(int scatter, const float* x, const float* w, const float* bias, float* y,
                        int B, int T, int Cin, int Fin, int Cout, int Fout, int KT, int S, int pad,
                        int w_layout, int act, int accum, int prec, double* bn_sums, const CruseBnBwd* bnb, int x_bf16, int y_bf16,
                        const CruseBnIn* bni, hipStream_t stream, const CruseBnBwdIn* bbi) {
    if (bni != nullptr && (prec != CRUSE_PREC_BF16X3 || Cin > 64 || accum || bnb != nullptr || x_bf16)) return 0;
    if (bbi != nullptr && (prec != CRUSE_PREC_BF16 || Cin > 64 || !x_bf16 || bni != nullptr || act)) return 0;
    if (y_bf16 && !x_bf16) return 0;                                    // (a bf16 output comes with a bf16 input: the backward chain)
    if (Cin % 8 != 0 || (Cin & (Cin - 1)) != 0 || Cout < 8 || Cout > 64 || (TFM * (Fout / (scatter ? 2 : 1))) % 16 != 0) return 0;
    CMArgs a = {};
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    a.B = B; a.T = T; a.Cin = Cin; a.Fin = Fin; a.Cout = Cout; a.Fout = Fout;
    a.act = act; a.accum = accum; a.sums = bn_sums;
    a.x_bf16 = x_bf16 ? 1 : 0; a.y_bf16 = y_bf16 ? 1 : 0;
    if (bbi != nullptr) {
        a.bb_y = bbi->y; a.bb_sums = bbi->sums; a.bb_nrep = bbi->nrep; a.bb_inv_count = 1.0 / (double)bbi->count;
        a.bb_mean = bbi->mean; a.bb_rstd = bbi->rstd; a.bb_gamma = bbi->gamma; a.bb_beta = bbi->beta; a.bb_relu = bbi->relu;
        a.bb_training = bbi->training; a.bb_copy = bbi->copy_bf16; a.bb_dgamma = bbi->dgamma; a.bb_dbeta = bbi->dbeta; a.bb_dbias = bbi->dbias;
    }
    if (bni != nullptr) {
        a.in_sums = bni->sums; a.in_nrep = bni->nrep; a.in_inv_count = 1.0 / (double)bni->count;
        a.in_unb = bni->count > 1 ? (double)bni->count / (double)(bni->count - 1) : 1.0;
        a.in_eps = bni->eps; a.in_mom = bni->momentum; a.in_gamma = bni->gamma; a.in_beta = bni->beta;
        a.in_mean_o = bni->mean_o; a.in_rstd_o = bni->rstd_o; a.in_rmean = bni->rmean; a.in_rvar = bni->rvar;
        a.in_add = bni->add; a.in_copy = bni->copy_bf16;
    }
    a.tdbg = cruse_opt("cm_dbg", 0);
    a.kint = cruse_opt("cm_kint", 1);
    a.swp_ok = cruse_opt("cm_swap", 1);                  // (A/B switch: 0 = channels in rows for every form)
    if (bnb != nullptr) {
        a.bn_y = bnb->y; a.bn_mean = bnb->mean; a.bn_rstd = bnb->rstd; a.bn_gamma = bnb->gamma; a.bn_beta = bnb->beta;
        a.bn_relu = bnb->relu;
    }
    if (!scatter) {
        // y[co,fo] = sum W(co,ci,kt,kf) x[t-(KT-1)+kt, ci, fo*S - pad + kf]
        a.S = S; a.OS = 1; a.nclass = 1; a.halo_lo = KT - 1;
        a.cls[0].ntaps = KT * 3; a.cls[0].par = 0;
        for (int kt = 0; kt < KT; ++kt)
            for (int kf = 0; kf < 3; ++kf) {
                const int i = kt * 3 + kf;
                a.cls[0].dt[i] = kt - (KT - 1);
                a.cls[0].df[i] = kf - pad;
                a.cls[0].wk[i] = w_layout == 0 ? kt * 3 + kf : (2 - kf);
            }
        if (w_layout == 0) { a.sco = (long long)Cin * KT * 3; a.sci = KT * 3; }
        else { a.sco = 3; a.sci = (long long)Cout * 3; }
    } else {
        // y[co,fo] = sum_{(fo+pad-kf) even} w[cs][co][kt][kf] g[t+(KT-1)-kt, cs, (fo+pad-kf)/2]; fo = 2m + par
        a.S = 1; a.OS = 2; a.nclass = 2; a.halo_lo = 0;
        a.sco = (long long)KT * 3; a.sci = (long long)Cout * KT * 3;
        for (int par = 0; par < 2; ++par) {
            TapClass& c = a.cls[par];
            c.par = par; c.ntaps = 0;
            for (int kt = 0; kt < KT; ++kt)
                for (int kf = 0; kf < 3; ++kf) {
                    const int q = par + pad - kf;            // fo + pad - kf = 2m + q
                    if (q & 1) continue;
                    const int i = c.ntaps++;
                    c.dt[i] = (KT - 1) - kt;
                    c.df[i] = q / 2;                         // q in {-2,0,2} -> -1,0,+1 (exact for even q)
                    c.wk[i] = kt * 3 + kf;
                }
        }
    }
    a.nrows = TFM + KT - 1;
    const int ks0 = (a.cls[0].ntaps * Cin + 31) / 32, ks1 = a.nclass > 1 ? (a.cls[1].ntaps * Cin + 31) / 32 : 0;
    const int mt = Cout <= 16 ? 1 : (Cout <= 32 ? 2 : 4);
    if ((Cin * Fin) % 4 != 0 || ((uintptr_t)x % 16) != 0) return 0;
    if (x_bf16 && ((Cin * Fin) % 8 != 0 || prec != CRUSE_PREC_BF16 || (bn_sums != nullptr && bnb == nullptr))) return 0;   // (bf16 slots cover 8 elements; bf16 mode, data-gradient forms)
    if ((TFM + KT - 1) * Cin * Fin > MAXV * 256 * 4) return 0;
    const size_t wbytes = (size_t)mt * (ks0 + ks1) * 512 *
                          (prec == CRUSE_PREC_F32 ? 4 : (prec == CRUSE_PREC_BF16X3 ? 4 : 2));
    const bool cf = prec != CRUSE_PREC_F32;      // channel-fastest pre-converted bf16 image (see the kernel)
    if (cf && ((Fin & 1) != 0 || a.nrows * (Fin / ((Fin & 3) ? 2 : 4)) * (Cin / 8) > ((Fin & 3) ? 2 : 1) * 256)) return 0;
    const size_t lds = wbytes + (cf ? (size_t)(prec == CRUSE_PREC_BF16X3 ? 2 : 1) * (size_t)((a.nrows * (Fin + 2) * (Cin / 8) + 15) & ~15) * 16
                                    : (size_t)a.nrows * Cin * Fin * sizeof(float));
    if (lds > 150 * 1024) return 0;
    const int ntiles = B * ((T + TFM - 1) / TFM);
    // small weight images: more, lighter workgroups hide latency better (44 vs 60 us on the 8->16 layer);
    // large ones (up to 48 KB of fragments per workgroup) amortise their prologue over more tiles
    int gmax = wbytes <= 16 * 1024 ? 1024 : 512;
    gmax = cruse_opt("cm_grid", gmax);
    const int grid = ntiles < gmax ? ntiles : gmax;
    int rc;
    const int ntile_wg = a.nclass * (TFM * (Fout / a.OS) / 16);       // N-tiles of one workgroup tile
    int nw = (ntile_wg == 5 || (ntile_wg == 10 && (Cin >= 64 || Cout >= 64))) ? 5 : 4;
    { const int e = cruse_opt("cm_nw", 0); if (e == 4 || e == 5) nw = e; }       // profiling option
    if (prec == CRUSE_PREC_F32) nw = 4;
    // the fused input BatchNorm backward holds 48 more prefetch registers per thread: the 5-wave (128-register) and 64-row variants spill
    // (decoder level 4: 223 us against 38 + 49 for the two separate kernels) -- those shapes keep the separate pass
    if (bbi != nullptr && (nw == 5 || mt > 2)) return 0;
    g_cm_last_prec = prec;
    if (prec == CRUSE_PREC_F32) rc = cruse_cm_launch_f32(a, grid, lds, nw, stream);
    else if (prec == CRUSE_PREC_BF16) rc = cruse_cm_launch_bf16(a, grid, lds, nw, stream);
    else rc = cruse_cm_launch_x3(a, grid, lds, nw, stream);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cruse_set_error("conv_mfma: HIP launch failed: %s", hipGetErrorString(e)); return CRUSE_E_HIP; }
    return 1;
}

// profiling: the phase stamps of the last conv_mfma launch with option cm_dbg = 1 (see g_cm_stamps)
extern "C" int cruse_conv_mfma_stamps(unsigned long long* out8) {
    if (g_cm_last_prec == CRUSE_PREC_F32) return cruse_cm_stamps_f32(out8);
    return g_cm_last_prec == CRUSE_PREC_BF16 ? cruse_cm_stamps_bf16(out8) : cruse_cm_stamps_x3(out8);
}
#endif   // CM_TU
