// Convolution weight gradients with REGISTER-DIRECT operands (backward-weights of nn.Conv2d / nn.ConvTranspose2d,
// model/cruse_net.py:138-143; the same contraction as wgrad_mfma.hip):
//
//   dW[ca, (tap, cb)] = sum over positions n = (frame, fa) of  a[frame][ca][fa] * bt[frame + kt - (KT-1)][cb][S*fa - pad + kf]
//
// The contraction index of the MFMA is the POSITION, and both tensors are bin-contiguous: the 8 consecutive positions a lane feeds
// v_mfma_f32_16x16x32_bf16 are 16 contiguous bytes of `a` and -- for stride 1 -- of `bt`, shifted by the tap; for stride 2 they are
// the even or the odd elements of 32 contiguous bytes.  So no LDS image, no patch matrix and no barrier: every wavefront loads its
// fragments straight from global memory (4-byte aligned 16-byte loads, 32-bit element offsets) and accumulates.  A SOURCE WINDOW --
// (frame offset kt, 16 channels cb) x the S * 8 bins under 8 positions, plus the dword before / after them -- is loaded once and serves
// the three taps kf: v_alignbit (stride 1) or v_perm (stride 2) of the same dwords.
//
// Lane (l & 15) is the row `ca` of the A fragment / the channel `cb` of the B fragments; lane group l >> 4 takes the (frame, window)
// pair 4 * step + group.  A row of Fa positions is cut into ceil(Fa / 8) windows of 8; the LAST window is end-aligned ([Fa - 8, Fa)) so
// that no load leaves its row.  Zeroed: the dword before the first / after the last window of a row (bins -1 and Fb), the whole source
// window of frame t - 1 at a clip's first frame, and -- on the A fragments, once per step -- the positions the end-aligned window shares
// with the one before it and the lanes past the frame range.
//
// A workgroup is 8 wavefronts that interleave 32-position steps of one contiguous range of frames and share nothing until the end,
// where their accumulators are summed through LDS (a fixed tree) into one slab -- the accumulator image itself, 256-byte stores -- that
// wgrad_rd_reduce_kernel sums over the slabs and scatters to dw[ca][cb][kt][kf].  ONE workgroup per CU: the per-workgroup costs (index
// set-up, LDS tree, slab + its reduction) outweigh what more resident waves hide (cruse_wgrad_rd_try).  Layers whose 3 * MT * pairs
// accumulator tiles exceed 12 per wave split their source windows over TG workgroups per slab, placed on one XCD (shared L2 for `a`).
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef u32x4 __attribute__((aligned(4))) u32x4_a4;      // the windows are 4-byte aligned (even element offsets of bf16 rows)
typedef u32x2 __attribute__((aligned(4))) u32x2_a4;

struct WRArgs {
    const void* a; const void* bt; float* partial;
    int T, Ca, Fa, Cb, Fb, KT;
    int NCH, ntaps, TG;                    // windows per row, taps, column-tile groups (workgroups per slab)
    int a_bf16, bt_bf16;
    int fpw;                               // frames per slab
    long long nframes;                     // B * T
    long long bt_elems;                    // elements of bt
    int fr_inc, c_inc;                     // (8 waves * 4 groups) / NCH and % NCH: the per-step advance of a lane group's (frame, window)
    int ns;                                // slabs
};



__device__ __forceinline__ unsigned pack_bf16(float x, float y) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 h;
    h[0] = (__bf16)x; h[1] = (__bf16)y;
    return __builtin_bit_cast(unsigned, h);
}

// A wavefront owns MTW row tiles (16 channels `ca` each: all of them) and NWP source windows; a source window is one (frame offset kt, tile of 16
// channels cb) pair and feeds THREE column tiles -- the taps kf = 0, 1, 2 are the same loaded dwords shifted.  S / PAD: the conv form (S1 P1:
// skip convs; S2 P1: encoder convs; S2 P0: transposed decoder convs); U: steps per loop iteration (the loads of U steps are in flight
// together); F32IN: an operand tensor may hold f32 (a_bf16 / bt_bf16 say which; the load registers are then sized for f32 windows).
template <int MTW, int NWP, int S, int PAD, int U, bool F32IN>
__global__ __launch_bounds__(512) void wgrad_rd_kernel(const WRArgs p) {
    static_assert((S == 1 && PAD == 1) || S == 2, "conv forms: stride 1 pad 1, stride 2 pad 0 / 1");
    constexpr int NWV = 8;
    constexpr int NMAIN = S == 2 ? 2 : 1;              // 16-byte pieces of the main part of a bf16 window (f32: twice as many)
    constexpr bool BEFORE = S == 1 || PAD == 1;        // a tap reaches one bin before the position window ...
    constexpr bool AFTER = S == 1 || PAD == 0;         // ... / one bin past its last source bin
    extern __shared__ __attribute__((aligned(16))) float red[];      // [4 slots][MTW * NWP * 3 tiles][4][64]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g = lane >> 4;
    // the TG workgroups of a slab read the same frames of `a`: they sit on ONE XCD (blockIdx % 8) so that its L2 fetches those rows once
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int tg = rr % p.TG, slab_i = (rr / p.TG) * 8 + xcd;
    if (slab_i >= p.ns) return;
    const int Ca = p.Ca, Fa = p.Fa, Cb = p.Cb, Fb = p.Fb, NCH = p.NCH, T = p.T;
    const long long f_lo = (long long)slab_i * p.fpw;
    const int nfr = (int)min((long long)p.fpw, p.nframes - f_lo);           // frames of this slab
    const bool a16 = !F32IN || p.a_bf16 != 0, b16 = !F32IN || p.bt_bf16 != 0;
    constexpr int FW = F32IN ? 2 : 1;

    // 32-bit ELEMENT offsets everywhere (the host checks both tensors stay below 2^31 elements)
    int a_row[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) a_row[i] = min(i * 16 + l15, Ca - 1) * Fa;
    const int rowa = Ca * Fa, rowb = Cb * Fb;
    // window x of this wave: pair index tg * NWP + x = kt * ncbt + cbt; the lane's channel cb = cbt * 16 + (lane & 15)
    const int ncbt = (Cb + 15) >> 4;
    int b_const[NWP]; bool b_prev[NWP];
#pragma unroll
    for (int x = 0; x < NWP; ++x) {
        const int pr = min(tg * NWP + x, p.KT * ncbt - 1);                   // (a pair past the end repeats the last one: computed, never stored)
        const int kt = pr / ncbt, cbt = pr - kt * ncbt;
        const int cb = cbt * 16 + l15;
        b_prev[x] = kt - (p.KT - 1) < 0;
        b_const[x] = min(cb, Cb - 1) * Fb + (kt - (p.KT - 1)) * rowb;
    }
    f32x4 acc[MTW][NWP][3];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int x = 0; x < NWP; ++x)
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[i][x][k] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this lane group's (frame, window) pair: q = (step * 8 + wave) * 4 + group
    int fr, c, tt;
    {
        const int q0 = wv * 4 + g;
        fr = q0 / NCH; c = q0 - fr * NCH;
        tt = (int)((f_lo + fr) % T);
    }
    const int nsteps = (nfr * NCH + NWV * 4 - 1) / (NWV * 4);
    const int lo_last = (8 * NCH - Fa) >> 1, w_last = Fa - 8;           // (dwords of the last window that belong to the window before it)
    const int f_lo32 = (int)f_lo;
    const int tot_b = (int)p.bt_elems;

    for (int it = 0; it < nsteps; it += U) {
        u32x4 ra[U][MTW][FW];
        u32x4 rbm[U][NWP][FW * NMAIN];
        u32x2 rbe[U][NWP][2];                                           // the dword (f32: the pair) before / after the main part
        int s_w[U], s_lo[U]; bool s_t0[U];
        // ---- loads of U steps ----
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool inr = fr < nfr;
            const int f = f_lo32 + (inr ? fr : nfr - 1);
            const bool last = c == NCH - 1;
            const int w = last ? w_last : 8 * c;
            s_w[u] = w; s_lo[u] = inr ? (last ? lo_last : 0) : 4; s_t0[u] = tt == 0;
            const int ao = f * rowa + w;
            if (a16) {
                const __bf16* ap = reinterpret_cast<const __bf16*>(p.a);
#pragma unroll
                for (int i = 0; i < MTW; ++i) ra[u][i][0] = *reinterpret_cast<const u32x4_a4*>(ap + (unsigned)(ao + a_row[i]));
            } else if constexpr (F32IN) {
                const float* ap = reinterpret_cast<const float*>(p.a);
#pragma unroll
                for (int i = 0; i < MTW; ++i) {
                    ra[u][i][0] = *reinterpret_cast<const u32x4_a4*>(ap + (unsigned)(ao + a_row[i]));
                    ra[u][i][1] = *reinterpret_cast<const u32x4_a4*>(ap + (unsigned)(ao + a_row[i] + 4));
                }
            }
            const int bo = f * rowb + S * w;
#pragma unroll
            for (int x = 0; x < NWP; ++x) {
                // main part: the S * 8 source bins of the position window, inside its row (frame -1 of the first clip: clamped, masked below);
                // the dwords before / after it are clamped into the tensor -- where they are outside the ROW they are zeroed below
                const int e = max(bo + b_const[x], 0);
                const int eb = max(e - 2, 0), ea = min(e + 8 * S, tot_b - 2);
                if (b16) {
                    const __bf16* bp = reinterpret_cast<const __bf16*>(p.bt);
#pragma unroll
                    for (int m = 0; m < NMAIN; ++m) rbm[u][x][m] = *reinterpret_cast<const u32x4_a4*>(bp + (unsigned)(e + 8 * m));
                    if constexpr (BEFORE) rbe[u][x][0].x = *reinterpret_cast<const unsigned*>(bp + (unsigned)eb);
                    if constexpr (AFTER) rbe[u][x][1].x = *reinterpret_cast<const unsigned*>(bp + (unsigned)ea);
                } else if constexpr (F32IN) {
                    const float* bp = reinterpret_cast<const float*>(p.bt);
#pragma unroll
                    for (int m = 0; m < 2 * NMAIN; ++m) rbm[u][x][m] = *reinterpret_cast<const u32x4_a4*>(bp + (unsigned)(e + 4 * m));
                    if constexpr (BEFORE) rbe[u][x][0] = *reinterpret_cast<const u32x2_a4*>(bp + (unsigned)eb);
                    if constexpr (AFTER) rbe[u][x][1] = *reinterpret_cast<const u32x2_a4*>(bp + (unsigned)ea);
                }
            }
            // advance by one step (32 pairs)
            c += p.c_inc;
            int adv = p.fr_inc;
            if (c >= NCH) { c -= NCH; ++adv; }
            fr += adv; tt += adv;
            while (tt >= T) tt -= T;
        }
        // ---- fragments and products ----
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // positions that are not this window's (the dwords the end-aligned last window shares with the one before it; lanes past the range)
            // are zeroed in the A fragments -- once per step instead of once per column tile
            const int nib = 0xf << s_lo[u];
            unsigned pm[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) pm[m] = (unsigned)__builtin_amdgcn_sbfe(nib, m, 1);
            bf16x8 fa[MTW];
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                u32x4 d = ra[u][i][0];
                if (!a16) {
                    const u32x4 d1 = ra[u][i][FW - 1];
                    d = (u32x4){pack_bf16(__uint_as_float(d.x), __uint_as_float(d.y)), pack_bf16(__uint_as_float(d.z), __uint_as_float(d.w)),
                                pack_bf16(__uint_as_float(d1.x), __uint_as_float(d1.y)), pack_bf16(__uint_as_float(d1.z), __uint_as_float(d1.w))};
                }
                d.x &= pm[0]; d.y &= pm[1]; d.z &= pm[2]; d.w &= pm[3];
                fa[i] = __builtin_bit_cast(bf16x8, d);
            }
            const int w = s_w[u];
#pragma unroll
            for (int x = 0; x < NWP; ++x) {
                // the window as dwords of two bf16: [before | main | after] (before: S1, S2 P1; after: S1, S2 P0)
                constexpr int NM = 4 * NMAIN, M0 = BEFORE ? 1 : 0, ND = NM + (BEFORE ? 1 : 0) + (AFTER ? 1 : 0);
                unsigned D[ND];
                if (b16) {
#pragma unroll
                    for (int m = 0; m < NMAIN; ++m) {
                        D[M0 + 4 * m] = rbm[u][x][m].x; D[M0 + 4 * m + 1] = rbm[u][x][m].y; D[M0 + 4 * m + 2] = rbm[u][x][m].z; D[M0 + 4 * m + 3] = rbm[u][x][m].w;
                    }
                    if constexpr (BEFORE) D[0] = rbe[u][x][0].x;
                    if constexpr (AFTER) D[ND - 1] = rbe[u][x][1].x;
                } else if constexpr (F32IN) {
#pragma unroll
                    for (int m = 0; m < 2 * NMAIN; ++m) {
                        D[M0 + 2 * m] = pack_bf16(__uint_as_float(rbm[u][x][m].x), __uint_as_float(rbm[u][x][m].y));
                        D[M0 + 2 * m + 1] = pack_bf16(__uint_as_float(rbm[u][x][m].z), __uint_as_float(rbm[u][x][m].w));
                    }
                    if constexpr (BEFORE) D[0] = pack_bf16(__uint_as_float(rbe[u][x][0].x), __uint_as_float(rbe[u][x][0].y));
                    if constexpr (AFTER) D[ND - 1] = pack_bf16(__uint_as_float(rbe[u][x][1].x), __uint_as_float(rbe[u][x][1].y));
                }
                // bins outside the row: bin -1 lives in the dword before the first window, bin Fb in the one after the last window
                if constexpr (BEFORE) { if (w == 0) D[0] = 0u; }
                if constexpr (AFTER) { if (w == w_last) D[ND - 1] = 0u; }
                // frame t - 1 of a clip's first frame is zero padding
                if (b_prev[x] && s_t0[u]) {
#pragma unroll
                    for (int m = 0; m < ND; ++m) D[m] = 0u;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {                          // tap kf = k: source bin S * position + k - PAD
                    unsigned out[4];
                    if constexpr (S == 1) {                            // elements w - 1 + k .. of [w - 2 | w .. w + 7 | w + 8]
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            out[m] = k == 1 ? D[m + 1] : __builtin_amdgcn_alignbit(D[m + (k == 0 ? 1 : 2)], D[m + (k == 0 ? 0 : 1)], 16);
                    } else if constexpr (PAD == 1) {                   // odd elements from the dword before (k = 0), even / odd elements of the main part
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            out[m] = k == 0 ? __builtin_amdgcn_perm(D[2 * m + 1], D[2 * m], 0x07060302u)
                                            : __builtin_amdgcn_perm(D[2 * m + 2], D[2 * m + 1], k == 1 ? 0x05040100u : 0x07060302u);
                    } else {                                           // even / odd elements of the main part, even elements one dword later (k = 2)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            out[m] = k == 2 ? __builtin_amdgcn_perm(D[2 * m + 2], D[2 * m + 1], 0x05040100u)
                                            : __builtin_amdgcn_perm(D[2 * m + 1], D[2 * m], k == 0 ? 0x05040100u : 0x07060302u);
                    }
                    const bf16x8 fb = __builtin_bit_cast(bf16x8, ((u32x4){out[0], out[1], out[2], out[3]}));
#pragma unroll
                    for (int i = 0; i < MTW; ++i) acc[i][x][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[i][x][k], 0, 0, 0);
                }
            }
        }
    }

    // ---- the 8 waves' accumulators -> one slab: fixed tree (4 + 2 + 1) through LDS ----
    constexpr int NTL = MTW * NWP * 3;
    auto put = [&](int slot) {
        float* r = red + (size_t)slot * NTL * 256;
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int x = 0; x < NWP; ++x)
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) r[(((i * NWP + x) * 3 + k) * 4 + q) * 64 + lane] = acc[i][x][k][q];
    };
    auto add = [&](int slot) {
        const float* r = red + (size_t)slot * NTL * 256;
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int x = 0; x < NWP; ++x)
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[i][x][k][q] += r[(((i * NWP + x) * 3 + k) * 4 + q) * 64 + lane];
    };
#pragma unroll
    for (int half = 4; half >= 1; half >>= 1) {
        if (wv >= half && wv < 2 * half) put(wv - half);
        __syncthreads();
        if (wv < half) add(wv);
        __syncthreads();
    }
    // ---- the slab of this (frame range, window group): the accumulator image itself ([tile][4][64 lanes], 256-byte stores); summed over the
    // slabs and scattered to dw[ca][cb][kt][kf] by wgrad_rd_reduce_kernel
    if (wv != 0) return;
    float* const my = p.partial + ((long long)slab_i * p.TG + tg) * (NTL * 256);
#pragma unroll
    for (int x = 0; x < NWP; ++x) {
        const int pr = tg * NWP + x;
        if (pr >= p.KT * ncbt || (pr % ncbt) * 16 + l15 >= Cb) continue;          // (padding columns: the reduction never reads them)
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (i * 16 + g * 4 + q < Ca) my[(((i * NWP + x) * 3 + k) * 4 + q) * 64 + lane] = acc[i][x][k][q];
    }
}

// dw[ca][cb][kt][kf] += sum over the slabs of the accumulator images; blockIdx.y owns a chunk of 16 slabs (one atomic per output and chunk,
// as wgrad_reduce_kernel: up to 16 slabs the sum has a fixed order)
__global__ __launch_bounds__(256) void wgrad_rd_reduce_kernel(const float* partial, int ns, int TG, int NWP, int ntl, int Ca, int Cb, int KT,
                                                              float* dw) {
    const int e = blockIdx.x * 256 + threadIdx.x;                      // element of a slab: [tg][tile][4][64]
    const int slab_elems = TG * ntl * 256;
    if (e >= slab_elems) return;
    const int tg = e / (ntl * 256), r = e - tg * (ntl * 256);
    const int tile = r >> 8, q = (r >> 6) & 3, ln = r & 63;
    const int i = tile / (NWP * 3), x = (tile / 3) % NWP, k = tile % 3;
    const int ncbt = (Cb + 15) >> 4, pr = tg * NWP + x;
    const int ca = i * 16 + (ln >> 4) * 4 + q;
    if (pr >= KT * ncbt || ca >= Ca) return;
    const int kt = pr / ncbt, cb = (pr - kt * ncbt) * 16 + (ln & 15);
    if (cb >= Cb) return;
    const int k0 = blockIdx.y * 16, k1 = min(ns, k0 + 16);
    float v[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = k0 + m < k1 ? partial[(long long)(k0 + m) * slab_elems + e] : 0.f;
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m) sum += v[m];
    atomicAdd(dw + (ca * Cb + cb) * (KT * 3) + kt * 3 + k, sum);
}

template <int MTW, int NWP, int U>
int launch_form(const WRArgs& p, int S, int pad, int grid, hipStream_t s) {
    const size_t lds = (size_t)4 * MTW * NWP * 3 * 256 * sizeof(float);
    int rc;
#define WR_LAUNCH(SV, PV)                                                                                                  \
    do {                                                                                                                   \
        if (p.a_bf16 && p.bt_bf16) {                                                                                       \
            if ((rc = cruse_ensure_dyn_lds(reinterpret_cast<const void*>(wgrad_rd_kernel<MTW, NWP, SV, PV, U, false>), lds, "wgrad_rd"))) return rc; \
            hipLaunchKernelGGL((wgrad_rd_kernel<MTW, NWP, SV, PV, U, false>), dim3(grid), dim3(512), lds, s, p);           \
        } else {                                                                                                           \
            if ((rc = cruse_ensure_dyn_lds(reinterpret_cast<const void*>(wgrad_rd_kernel<MTW, NWP, SV, PV, U, true>), lds, "wgrad_rd"))) return rc; \
            hipLaunchKernelGGL((wgrad_rd_kernel<MTW, NWP, SV, PV, U, true>), dim3(grid), dim3(512), lds, s, p);            \
        }                                                                                                                  \
    } while (0)
    if (S == 1) WR_LAUNCH(1, 1);
    else if (pad == 1) WR_LAUNCH(2, 1);
    else WR_LAUNCH(2, 0);
#undef WR_LAUNCH
    return CRUSE_OK;
}

}  // namespace

// 1 = handled (dw updated: *nblk_out = 0 slabs left to reduce), 0 = not eligible, < 0 error
int cruse_wgrad_rd_try(const float* a, const float* bt, float* partial, size_t ws_bytes, float* dw,
                       int B, int T, int Ca, int Fa, int Cb, int Fb, int KT, int S, int pad, int prec, int a_bf16, int bt_bf16,
                       int* nblk_out, hipStream_t stream) {
    if (prec != CRUSE_PREC_BF16 || cruse_opt("wg_rd", 1) == 0) return 0;
    const int ntaps = KT * 3;
    const int MT = (Ca + 15) / 16, npairs = KT * ((Cb + 15) / 16);        // row tiles; source windows (frame offset, 16 channels of bt)
    if (Ca > 64 || npairs > 4 || Fa < 8 || (Fa & 1) || Fb != S * Fa || !((S == 1 && pad == 1) || S == 2) || (KT != 1 && KT != 2)) return 0;
    if ((reinterpret_cast<uintptr_t>(a) & 3) || (reinterpret_cast<uintptr_t>(bt) & 3)) return 0;
    const long long nframes = (long long)B * T;
    const int mtw = MT <= 1 ? 1 : (MT == 2 ? 2 : 4);
    const int nwp = mtw == 4 ? 1 : (npairs >= 2 ? 2 : 1);                // 3 * mtw * nwp accumulator tiles per wave (<= 12)
    const int TG = (npairs + nwp - 1) / nwp;
    WRArgs p = {};
    p.a = a; p.bt = bt; p.partial = partial;
    p.T = T; p.Ca = Ca; p.Fa = Fa; p.Cb = Cb; p.Fb = Fb; p.KT = KT;
    p.NCH = (Fa + 7) / 8; p.ntaps = ntaps; p.TG = TG;
    p.a_bf16 = a_bf16 ? 1 : 0; p.bt_bf16 = bt_bf16 ? 1 : 0;
    p.nframes = nframes; p.bt_elems = nframes * Cb * Fb;
    p.fr_inc = 32 / p.NCH; p.c_inc = 32 % p.NCH;
    // ONE 8-wave workgroup per CU (256 in all; a slab is written by TG of them, one per group of source windows): measured at the bench shapes,
    // 128 / 256 / 512 / 1024 workgroups: 363 / 266 / 322 / 447 us for the twelve launches of a step (tools/wgrad_probe.py) -- the per-workgroup
    // costs (index set-up, the LDS tree, the slab and its reduction) outweigh what more waves hide; deeper unrolling (U x 2) measured equal
    int ns = cruse_opt("wg_grid", 0) > 0 ? cruse_opt("wg_grid", 0) : 256 / TG;
    const size_t slab_bytes = (size_t)TG * mtw * nwp * 3 * 256 * sizeof(float);
    if (slab_bytes > ws_bytes) return 0;
    if ((size_t)ns * slab_bytes > ws_bytes) ns = (int)(ws_bytes / slab_bytes);
    if ((long long)ns > nframes / 4) ns = (int)(nframes / 4);   // (small problems: <= 16 slabs, i.e. one chunk of the reduction -- a fixed summation order)
    if (ns < 1) ns = 1;
    p.fpw = (int)((nframes + ns - 1) / ns);
    ns = (int)((nframes + p.fpw - 1) / p.fpw);
    if ((long long)p.fpw * p.NCH > (1ll << 30) || nframes * Ca * Fa >= (1ll << 31) - 64 || p.bt_elems >= (1ll << 31) - 64 || p.bt_elems < 16) return 0;
    p.ns = ns;
    const int grid = ((ns + 7) / 8) * 8 * TG;                  // (slab = (block / 8 / TG) * 8 + block % 8: whole rounds of the 8 XCDs)
    int rc;
    if (mtw == 1 && nwp == 1) rc = launch_form<1, 1, 4>(p, S, pad, grid, stream);
    else if (mtw == 1) rc = launch_form<1, 2, 2>(p, S, pad, grid, stream);
    else if (mtw == 2 && nwp == 1) rc = launch_form<2, 1, 2>(p, S, pad, grid, stream);
    else if (mtw == 2) rc = launch_form<2, 2, 1>(p, S, pad, grid, stream);
    else rc = launch_form<4, 1, 1>(p, S, pad, grid, stream);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cruse_set_error("wgrad_rd: HIP launch failed: %s", hipGetErrorString(e)); return CRUSE_E_HIP; }
    const int ntl = mtw * nwp * 3;
    hipLaunchKernelGGL(wgrad_rd_reduce_kernel, dim3((TG * ntl * 256 + 255) / 256, (ns + 15) / 16), dim3(256), 0, stream,
                       (const float*)partial, ns, TG, nwp, ntl, Ca, Cb, KT, dw);
    e = hipGetLastError();
    if (e != hipSuccess) { cruse_set_error("wgrad_rd_reduce: HIP launch failed: %s", hipGetErrorString(e)); return CRUSE_E_HIP; }
    *nblk_out = 0;                                             // (nothing left for wgrad_reduce_kernel)
    return 1;
}
