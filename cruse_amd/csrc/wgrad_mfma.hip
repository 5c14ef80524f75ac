// Convolution weight gradients on MFMA (backward-weights of nn.Conv2d / nn.ConvTranspose2d,
// model/cruse_net.py:138-143):
//
//   dW[ca, j] = sum_n A[ca, n] * Patch[j, n],   n = (frame, bin fa),  j = (tap, cb)
//   Patch[(kt,kf,cb), (t,fa)] = bt[t - (KT-1) + kt][cb][fa*S - pad + kf]
//
// The contraction runs over POSITIONS, so both MFMA operands need 8 consecutive positions per lane:
//   A  : the `a` rows are already bin-contiguous -> LDS [ca][frame][FaP] (FaP = Fa rounded up to 8, zeros)
//   B  : the im2col patch matrix of the tile is materialised IN LDS, one bin-contiguous row per
//        (tap, cb, frame), built from the raw `bt` frames staged once from HBM.
// M = Ca (16-row tiles), N = taps*Cb (16-column tiles, split over the 4 wavefronts), K = positions in
// steps of 32 (4 lane groups x 8 bins).  A workgroup is persistent over frame tiles and keeps its
// accumulators in registers; it writes one partial slab at the end, summed by wgrad_reduce_kernel.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int MAXV = 8;
constexpr int RPAD = 8;

struct WMArgs {
    const float* a; const float* bt; float* partial;
    int B, T, Ca, Fa, Cb, Fb, KT, S, pad;
    int FaP, NCH, ntaps, nrows, ntiles_total;
    int a_bf16, bt_bf16;                   // operand tensors hold bf16 elements (backward-only tensors stored in bf16; PREC == bf16 only)
    int dbg;                               // profiling only (CRUSE_WG_DBG bit mask: skip 1 patch build, 2 MFMA loop, 4 loads, 8 A/raw image)
};

template <int PREC> struct WStore {
    typedef __bf16 elem;
    static constexpr int NPL = (PREC == CRUSE_PREC_BF16X3) ? 2 : 1;
};
template <> struct WStore<CRUSE_PREC_F32> {
    typedef float elem;
    static constexpr int NPL = 1;
};

template <int PREC>
__device__ __forceinline__ void wput(typename WStore<PREC>::elem* base, size_t plane, size_t off, float v) {
    if constexpr (PREC == CRUSE_PREC_F32) base[off] = v;
    else if constexpr (PREC == CRUSE_PREC_BF16) base[off] = (__bf16)v;
    else { __bf16 h, l; split_bf16(v, h, l); base[off] = h; base[plane + off] = l; }
}
// two consecutive elements (even offset) in one LDS store
template <int PREC>
__device__ __forceinline__ void wput2(typename WStore<PREC>::elem* base, size_t plane, size_t off, float v0, float v1) {
    if constexpr (PREC == CRUSE_PREC_F32) {
        *reinterpret_cast<float2*>(base + off) = make_float2(v0, v1);
    } else {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        bf16x2 h;
        h[0] = (__bf16)v0; h[1] = (__bf16)v1;
        *reinterpret_cast<bf16x2*>(base + off) = h;
        if constexpr (PREC == CRUSE_PREC_BF16X3) {
            bf16x2 l;
            l[0] = (__bf16)(v0 - (float)h[0]); l[1] = (__bf16)(v1 - (float)h[1]);
            *reinterpret_cast<bf16x2*>(base + plane + off) = l;
        }
    }
}
template <int PREC>
__device__ __forceinline__ Frag<PREC> wget(const typename WStore<PREC>::elem* base, size_t plane, size_t off) {
    Frag<PREC> f;
    if constexpr (PREC == CRUSE_PREC_F32) {
        const float4 a0 = *reinterpret_cast<const float4*>(base + off);
        const float4 a1 = *reinterpret_cast<const float4*>(base + off + 4);
        f.v[0] = a0.x; f.v[1] = a0.y; f.v[2] = a0.z; f.v[3] = a0.w;
        f.v[4] = a1.x; f.v[5] = a1.y; f.v[6] = a1.z; f.v[7] = a1.w;
    } else {
        f.h = *reinterpret_cast<const bf16x8*>(base + off);
        if constexpr (PREC == CRUSE_PREC_BF16X3) f.l = *reinterpret_cast<const bf16x8*>(base + plane + off);
    }
    return f;
}

// TFW frames per tile; MT = ceil(Ca/16) row tiles; NTW = column tiles per wavefront
// (The plain-bf16 mode of the training step runs wgrad_rd.hip -- fragments straight from global memory, no LDS image; this kernel serves the
// f32 / split-bf16 modes and the bf16 shapes that one does not take: rows of fewer than 8 or an odd number of positions, Fb != S * Fa.)
template <int PREC, int TFW, int MT, int NTW>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WMArgs p) {
    typedef typename WStore<PREC>::elem elem;
    constexpr int NPL = WStore<PREC>::NPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int FaP = p.FaP, NJ = NTW * 4 * 16;               // padded patch rows
    // operand row stride: TFW*FaP is a multiple of 64 elements, so unpadded rows put the 16 rows of a fragment read
    // on the same LDS banks (8-16-way conflicts on every ds_read_b128); RPAD elements (16 bytes) de-phase them
    const int RS = TFW * FaP + RPAD;
    const size_t aplane = (size_t)MT * 16 * RS;             // elements per A plane
    const size_t bplane = (size_t)NJ * RS;                  // elements per patch plane
    elem* al = reinterpret_cast<elem*>(smem_raw);           // [NPL][MT*16][TFW][FaP]
    elem* pl = al + NPL * aplane;                            // [NPL][NJ][TFW][FaP]
    float* rawl = reinterpret_cast<float*>(pl + NPL * bplane);   // [nrows][Cb][Fb]
    const int rowa = p.Ca * p.Fa, rowb = p.Cb * p.Fb;
    const int ntile_t = (p.T + TFW - 1) / TFW;
    const int nva = TFW * rowa / 4, nvb = p.nrows * rowb / 4;

    // zero both operand images once: pad bins, pad rows and pad columns stay zero for good
    {
        uint4* z = reinterpret_cast<uint4*>(al);             // both images are multiples of 16 bytes
        const size_t n16 = NPL * (aplane + bplane) * sizeof(elem) / 16;
        for (size_t i = tid; i < n16; i += 256) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- tile-invariant coordinates, computed once (the runtime integer divisions they need used to sit
    //      in the per-element loops and dominated the kernel) -------------------------------------------------
    // A staging: slot q covers 4 consecutive floats of the [TFW][Ca][Fa] tile = two bin PAIRS (Fa is even, so a pair
    // never straddles a row); LDS element offset per pair
    int a_off[MAXV][2];
    bool a_ok[MAXV];
#pragma unroll
    for (int q = 0; q < MAXV; ++q) {
        const int i = tid + 256 * q;
        a_ok[q] = i < nva;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = min(i, nva - 1) * 4 + 2 * u;
            const int r = e / rowa, j = e - r * rowa;
            const int ca = j / p.Fa, fa = j - ca * p.Fa;
            a_off[q][u] = ca * RS + r * FaP + fa;
        }
    }
    // patch build: 768 virtual threads (3 slots per thread); virtual thread vt owns the bin PAIR vt % P2 of the
    // TFW x Fa tile and the channels cb = vt / P2, + ngroups, ...  A pair's three taps read the 4 (S = 1) or 5
    // (S = 2) consecutive source bins fb0 .. fb0+NSRC-1 and write three packed bf16x2 / float2 entries.
    constexpr int MAXP = 3;
    const int P2 = TFW * p.Fa / 2;
    // alignment class of the patch source reads (the staged rows start 16-byte aligned when Fb % 4 == 0)
    const int src_al = (p.Fb % 4 != 0) ? 0 : (p.S == 2 && p.pad == 1) ? 2 : (p.S == 1 && p.pad == 1) ? 1 :
                       (p.S == 2 && p.pad == 0) ? 3 : 0;
    const int ngroups = max(1, (256 * MAXP) / P2);
    const int tap_stride = p.Cb * RS, cb_stride = RS;
    int p_dst[MAXP], p_src[MAXP];                            // dst = cb0*cb_stride + tl*FaP + fa; src = cb0*Fb + tl*rowb + fb0
    unsigned p_ok[MAXP];                                     // bit i: source bin fb0 + i is inside [0, Fb)
#pragma unroll
    for (int sidx = 0; sidx < MAXP; ++sidx) {
        const int vt = tid + 256 * sidx;
        const int grp = vt / P2, pos = vt - grp * P2;
        p_dst[sidx] = -1; p_src[sidx] = 0; p_ok[sidx] = 0;
        if (grp < ngroups) {
            const int tl = (2 * pos) / p.Fa, fa = 2 * pos - tl * p.Fa;
            const int fb0 = fa * p.S - p.pad;
            p_dst[sidx] = grp * cb_stride + tl * FaP + fa;
            p_src[sidx] = grp * p.Fb + tl * rowb + fb0;
            for (int i = 0; i < 5; ++i)
                if (fb0 + i >= 0 && fb0 + i < p.Fb) p_ok[sidx] |= 1u << i;
        }
    }
    // frame index of every load slot inside the tile, 4 bits each (tile-invariant; the divisions used to sit in
    // prefetch() and cost ~2 us per tile)
    unsigned fr_a = 0, fr_b = 0;
#pragma unroll
    for (int q = 0; q < MAXV; ++q) {
        const int i = tid + 256 * q;
        fr_a |= (unsigned)min(15, (i * 4) / rowa) << (4 * q);
        fr_b |= (unsigned)min(15, (i * 4) / rowb) << (4 * q);
    }
    float4 pa[MAXV], pb[MAXV];
    auto prefetch = [&](int tile) {
        const int b = tile / ntile_t;
        const int t0 = (tile - b * ntile_t) * TFW;
        const float* srca = p.a + ((long long)b * p.T + t0) * rowa;
        const float* srcb = p.bt + ((long long)b * p.T + (t0 - (p.KT - 1))) * rowb;
        // (dtype branches OUTSIDE the slot loops: see conv_mfma.hip)
        const bool ld = !(p.dbg & 4);
        if (p.a_bf16) {                                      // 4 bf16 = 8 bytes per slot, the bit patterns travel in .x / .y
            const __bf16* sa = reinterpret_cast<const __bf16*>(p.a) + ((long long)b * p.T + t0) * rowa;
#pragma unroll
            for (int q = 0; q < MAXV; ++q) {
                const int i = tid + 256 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ld && i < nva && t0 + (int)((fr_a >> (4 * q)) & 15u) < p.T) {
                    const float2 w2 = *reinterpret_cast<const float2*>(sa + i * 4);
                    v.x = w2.x; v.y = w2.y;
                }
                pa[q] = v;
            }
        } else {
#pragma unroll
            for (int q = 0; q < MAXV; ++q) {
                const int i = tid + 256 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ld && i < nva && t0 + (int)((fr_a >> (4 * q)) & 15u) < p.T) v = *reinterpret_cast<const float4*>(srca + i * 4);
                pa[q] = v;
            }
        }
        if (p.bt_bf16) {
            const __bf16* sb = reinterpret_cast<const __bf16*>(p.bt) + ((long long)b * p.T + (t0 - (p.KT - 1))) * rowb;
#pragma unroll
            for (int q = 0; q < MAXV; ++q) {
                const int i = tid + 256 * q;
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < nvb) {
                    const int t = t0 - (p.KT - 1) + (int)((fr_b >> (4 * q)) & 15u);
                    if (ld && t >= 0 && t < p.T) {
                        const float2 w2 = *reinterpret_cast<const float2*>(sb + i * 4);
                        w.x = w2.x; w.y = w2.y;
                    }
                }
                pb[q] = w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < MAXV; ++q) {
                const int i = tid + 256 * q;
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < nvb) {
                    const int t = t0 - (p.KT - 1) + (int)((fr_b >> (4 * q)) & 15u);
                    if (ld && t >= 0 && t < p.T) w = *reinterpret_cast<const float4*>(srcb + i * 4);
                }
                pb[q] = w;
            }
        }
    };
    if ((int)blockIdx.x < p.ntiles_total) prefetch(blockIdx.x);
    const int nks = TFW * p.NCH / 4;                         // K steps per tile (TFW*NCH % 4 == 0, host-checked)
    for (int tile = blockIdx.x; tile < p.ntiles_total; tile += gridDim.x) {
        __syncthreads();                                     // previous tile's fragment reads are done
        // A image: [t][ca][fa] -> al[ca][tl][fa];  raw bt frames -> rawl (straight copy)
        if (PREC == CRUSE_PREC_BF16 && p.a_bf16) {           // already the operand's type: two 4-byte stores, no conversion
#pragma unroll
            for (int q = 0; q < MAXV; ++q) {
                if (a_ok[q] && !(p.dbg & 8)) {
                    *reinterpret_cast<float*>(al + a_off[q][0]) = pa[q].x;
                    *reinterpret_cast<float*>(al + a_off[q][1]) = pa[q].y;
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < MAXV; ++q) {
                if (a_ok[q] && !(p.dbg & 8)) {
                    wput2<PREC>(al, aplane, (size_t)a_off[q][0], pa[q].x, pa[q].y);
                    wput2<PREC>(al, aplane, (size_t)a_off[q][1], pa[q].z, pa[q].w);
                }
            }
        }
        if (p.bt_bf16) {
#pragma unroll
            for (int q = 0; q < MAXV; ++q) {
                const int i = tid + 256 * q;
                if (i < nvb) {
                    const unsigned w0 = __float_as_uint(pb[q].x), w1 = __float_as_uint(pb[q].y);
                    *reinterpret_cast<float4*>(rawl + i * 4) = make_float4(__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u),
                                                                           __uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u));
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < MAXV; ++q) {
                const int i = tid + 256 * q;
                if (i < nvb) *reinterpret_cast<float4*>(rawl + i * 4) = pb[q];
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < p.ntiles_total) prefetch(tile + gridDim.x);
        // patch image: pl[j = tap*Cb + cb][tl][fa] = bt[tl + kt][cb][fa*S - pad + kf]
#pragma unroll
        for (int sidx = 0; sidx < MAXP; ++sidx) {
            if (p_dst[sidx] < 0 || (p.dbg & 1)) continue;
            const unsigned ok = p_ok[sidx];
            int dst = p_dst[sidx], src = p_src[sidx];
            for (int cb = (tid + 256 * sidx) / P2; cb < p.Cb; cb += ngroups) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    if (kt < p.KT) {
                        const float* rs_ = rawl + src + kt * rowb;      // frame tl + kt of the staged rows
                        // fb0 is odd (2*fa - 1 or fa - 1 with fa even) except for the pad-0 decoder form: the aligned
                        // part of the 4-5 source bins comes in one or two wide LDS reads
                        float v[5];
                        if (src_al == 2) {                               // S = 2, pad = 1: bins fb0+1 .. fb0+4 are 16-byte aligned
                            const float4 w = *reinterpret_cast<const float4*>(rs_ + 1);
                            v[0] = rs_[0]; v[1] = w.x; v[2] = w.y; v[3] = w.z; v[4] = w.w;
                        } else if (src_al == 1) {                        // S = 1, pad = 1: bins fb0+1, fb0+2 are 8-byte aligned
                            const float2 w = *reinterpret_cast<const float2*>(rs_ + 1);
                            v[0] = rs_[0]; v[1] = w.x; v[2] = w.y; v[3] = rs_[3]; v[4] = 0.f;
                        } else if (src_al == 3) {                        // S = 2, pad = 0: bins fb0 .. fb0+3 are 16-byte aligned
                            const float4 w = *reinterpret_cast<const float4*>(rs_);
                            v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w; v[4] = rs_[4];
                        } else {
#pragma unroll
                            for (int i = 0; i < 5; ++i) v[i] = rs_[i];
                        }
#pragma unroll
                        for (int i = 0; i < 5; ++i) v[i] = (ok >> i) & 1u ? v[i] : 0.f;
                        elem* d = pl + dst + kt * 3 * tap_stride;
                        if (p.S == 2) {
                            wput2<PREC>(d, bplane, 0, v[0], v[2]);
                            wput2<PREC>(d, bplane, (size_t)tap_stride, v[1], v[3]);
                            wput2<PREC>(d, bplane, (size_t)2 * tap_stride, v[2], v[4]);
                        } else {
                            wput2<PREC>(d, bplane, 0, v[0], v[1]);
                            wput2<PREC>(d, bplane, (size_t)tap_stride, v[1], v[2]);
                            wput2<PREC>(d, bplane, (size_t)2 * tap_stride, v[2], v[3]);
                        }
                    }
                }
                dst += ngroups * cb_stride;
                src += ngroups * p.Fb;
            }
        }
        __syncthreads();
        // K steps: lane group (lane >> 4) takes chunk ks*4 + group; (frame, bin-chunk) advance without dividing
        int tl = (lane >> 4) / p.NCH, fc = (lane >> 4) - tl * p.NCH;
        for (int ks = 0; ks < ((p.dbg & 2) ? 0 : nks); ++ks) {
            const size_t poff = (size_t)(tl * FaP + fc * 8);
            Frag<PREC> fa_[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                fa_[i] = wget<PREC>(al, aplane, (size_t)(i * 16 + (lane & 15)) * RS + poff);
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int col = (wv * NTW + j) * 16 + (lane & 15);
                const Frag<PREC> fb_ = wget<PREC>(pl, bplane, (size_t)col * RS + poff);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j] = mma(fa_[i], fb_, acc[i][j]);
            }
            fc += 4;
            while (fc >= p.NCH) { fc -= p.NCH; ++tl; }
        }
    }
    // partial slab [Ca][Cb][KT*3]
    const int nout = p.Ca * p.Cb * p.KT * 3;
    float* slab = p.partial + (long long)blockIdx.x * nout;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int col = (wv * NTW + j) * 16 + (lane & 15);
            if (col < p.ntaps * p.Cb) {
                const int tap = col / p.Cb, cb = col - tap * p.Cb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ca = i * 16 + (lane >> 4) * 4 + r;
                    if (ca < p.Ca) slab[((long long)ca * p.Cb + cb) * (p.KT * 3) + tap] = acc[i][j][r];
                }
            }
        }
}

template <int PREC, int TFW, int MT>
int launch_ntw(const WMArgs& p, int ntw, int grid, size_t lds, hipStream_t s) {
    int rc;
#define WM_LAUNCH(NTWV)                                                                                       \
    do {                                                                                                      \
        if ((rc = cruse_ensure_dyn_lds(reinterpret_cast<const void*>(wgrad_mfma_kernel<PREC, TFW, MT, NTWV>), lds, \
                                       "wgrad_mfma"))) return rc;                                             \
        hipLaunchKernelGGL((wgrad_mfma_kernel<PREC, TFW, MT, NTWV>), dim3(grid), dim3(256), lds, s, p);       \
    } while (0)
    if (ntw <= 1) WM_LAUNCH(1);
    else if (ntw == 2) WM_LAUNCH(2);
    else WM_LAUNCH(3);
#undef WM_LAUNCH
    return CRUSE_OK;
}

template <int PREC, int TFW>
int launch_mt(const WMArgs& p, int mt, int ntw, int grid, size_t lds, hipStream_t s) {
    if (mt <= 1) return launch_ntw<PREC, TFW, 1>(p, ntw, grid, lds, s);
    if (mt == 2) return launch_ntw<PREC, TFW, 2>(p, ntw, grid, lds, s);
    return launch_ntw<PREC, TFW, 4>(p, ntw, grid, lds, s);
}

}  // namespace

// 1 = handled (partial slabs written, *nblk_out = number of slabs), 0 = not eligible, < 0 error
int cruse_wgrad_mfma_try(const float* a, const float* bt, float* partial, int max_slabs,
                         int B, int T, int Ca, int Fa, int Cb, int Fb, int KT, int S, int pad, int prec, int a_bf16, int bt_bf16,
                         int* nblk_out, hipStream_t stream) {
    if ((a_bf16 || bt_bf16) && prec != CRUSE_PREC_BF16) return 0;      // (bf16-stored tensors belong to the bf16 mode)
    const int FaP = (Fa + 7) / 8 * 8, NCH = FaP / 8;
    const int ntaps = KT * 3;
    const int mt = Ca <= 16 ? 1 : (Ca <= 32 ? 2 : 4);
    const int ntiles_n = (ntaps * Cb + 15) / 16;
    const int ntw = (ntiles_n + 3) / 4;
    if (Ca > 64 || ntw > 3) return 0;
    if (Fa & 1) return 0;            // the staging works on PAIRS of bins of one row (kernel: "Fa is even"): odd widths go to the exact VALU kernel
    if ((Ca * Fa) % 4 != 0 || (Cb * Fb) % 4 != 0 || ((uintptr_t)a % 16) != 0 || ((uintptr_t)bt % 16) != 0) return 0;
    const int esz = (prec == CRUSE_PREC_F32) ? 4 : 2, npl = (prec == CRUSE_PREC_BF16X3) ? 2 : 1;
    auto lds_of = [&](int tf) {
        return ((size_t)mt * 16 + (size_t)ntw * 64) * (tf * FaP + RPAD) * esz * npl + (size_t)(tf + KT - 1) * Cb * Fb * 4;
    };
    auto fits = [&](int tf) {
        return (tf * NCH) % 4 == 0 && tf * Ca * Fa <= MAXV * 1024 && (tf + KT - 1) * Cb * Fb <= MAXV * 1024 && tf * Fa <= 768 &&
               lds_of(tf) <= 150 * 1024;
    };
    // Frames per tile and workgroups per CU.  The kernel runs one wave per SIMD per workgroup and is latency-bound in
    // every phase (staging, patch build, K loop: SQ_WAIT_ANY 47 % of wave time), so TWO workgroups per CU are worth more
    // than long tiles: 8-frame tiles where two of them fit the CU's LDS (12 launches alone: 766 -> 588 us in total),
    // else 4-frame tiles; one 8-frame workgroup per CU only where neither fits twice.
    const size_t two_wg = 156 * 1024;
    int tfw = 4, gcap = 256;
    if (prec == CRUSE_PREC_BF16) {
        if (fits(8) && 2 * lds_of(8) <= two_wg) { tfw = 8; gcap = 512; }
        else if (fits(4) && 2 * lds_of(4) <= two_wg) { tfw = 4; gcap = 512; }
        else if (fits(8)) { tfw = 8; gcap = 256; }
    } else if (fits(4) && 2 * lds_of(4) <= two_wg) {
        gcap = 512;
    }
    if (!fits(tfw)) return 0;
    const size_t lds = lds_of(tfw);
    WMArgs p = {};
    p.a = a; p.bt = bt; p.partial = partial;
    p.B = B; p.T = T; p.Ca = Ca; p.Fa = Fa; p.Cb = Cb; p.Fb = Fb; p.KT = KT; p.S = S; p.pad = pad;
    p.FaP = FaP; p.NCH = NCH; p.ntaps = ntaps; p.nrows = tfw + KT - 1;
    p.ntiles_total = B * ((T + tfw - 1) / tfw);
    p.dbg = cruse_opt("wg_dbg", 0);
    p.a_bf16 = a_bf16 ? 1 : 0; p.bt_bf16 = bt_bf16 ? 1 : 0;
    int grid = p.ntiles_total < max_slabs ? p.ntiles_total : max_slabs;
    if (grid > gcap) grid = gcap;               // resident blocks only; fewer partial slabs to reduce
    int rc;
    if (prec == CRUSE_PREC_F32) rc = launch_mt<CRUSE_PREC_F32, 4>(p, mt, ntw, grid, lds, stream);
    else if (prec == CRUSE_PREC_BF16X3) rc = launch_mt<CRUSE_PREC_BF16X3, 4>(p, mt, ntw, grid, lds, stream);
    else if (tfw == 4) rc = launch_mt<CRUSE_PREC_BF16, 4>(p, mt, ntw, grid, lds, stream);
    else rc = launch_mt<CRUSE_PREC_BF16, 8>(p, mt, ntw, grid, lds, stream);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cruse_set_error("wgrad_mfma: HIP launch failed: %s", hipGetErrorString(e)); return CRUSE_E_HIP; }
    *nblk_out = grid;
    return 1;
}
