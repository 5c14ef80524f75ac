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

namespace {

constexpr int MAXV = 8;

struct WMArgs {
    const float* a; const float* bt; float* partial;
    int B, T, Ca, Fa, Cb, Fb, KT, S, pad;
    int FaP, NCH, ntaps, nrows, ntiles_total;
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
template <int PREC, int TFW, int MT, int NTW>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WMArgs p) {
    typedef typename WStore<PREC>::elem elem;
    constexpr int NPL = WStore<PREC>::NPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int FaP = p.FaP, NJ = NTW * 4 * 16;               // padded patch rows
    const size_t aplane = (size_t)MT * 16 * TFW * FaP;      // elements per A plane
    const size_t bplane = (size_t)NJ * TFW * FaP;           // elements per patch plane
    elem* al = reinterpret_cast<elem*>(smem_raw);           // [NPL][MT*16][TFW][FaP]
    elem* pl = al + NPL * aplane;                            // [NPL][NJ][TFW][FaP]
    float* rawl = reinterpret_cast<float*>(pl + NPL * bplane);   // [nrows][Cb][Fb]
    const int rowa = p.Ca * p.Fa, rowb = p.Cb * p.Fb;
    const int ntile_t = (p.T + TFW - 1) / TFW;
    const int nva = TFW * rowa / 4, nvb = p.nrows * rowb / 4;

    // zero both operand images once: pad bins, pad rows and pad columns stay zero for good
    for (size_t i = tid; i < NPL * (aplane + bplane); i += 256) al[i] = (elem)0.f;

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- tile-invariant coordinates, computed once (the runtime integer divisions they need used to sit
    //      in the per-element loops and dominated the kernel) -------------------------------------------------
    // A staging: slot q covers 4 consecutive floats of the [TFW][Ca][Fa] tile; LDS offset per element
    int a_off[MAXV][4];
    bool a_ok[MAXV];
#pragma unroll
    for (int q = 0; q < MAXV; ++q) {
        const int i = tid + 256 * q;
        a_ok[q] = i < nva;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = min(i, nva - 1) * 4 + u;
            const int r = e / rowa, j = e - r * rowa;
            const int ca = j / p.Fa, fa = j - ca * p.Fa;
            a_off[q][u] = (ca * TFW + r) * FaP + fa;
        }
    }
    // patch build: 768 virtual threads (3 slots per thread); virtual thread vt owns position vt % P of the
    // TFW x Fa tile and the channels cb = vt / P, + ngroups, ... -- all offsets fixed per slot
    constexpr int MAXP = 3;
    const int P = TFW * p.Fa;
    const int ngroups = max(1, (256 * MAXP) / P);
    int p_dst[MAXP], p_cb0[MAXP], p_src[MAXP][3];            // dst = tl*FaP + fa; src[kf] = tl*Cb*Fb + fb (or -1)
#pragma unroll
    for (int sidx = 0; sidx < MAXP; ++sidx) {
        const int vt = tid + 256 * sidx;
        const int grp = vt / P, pos = vt - grp * P;
        p_dst[sidx] = -1; p_cb0[sidx] = 0;
#pragma unroll
        for (int kf = 0; kf < 3; ++kf) p_src[sidx][kf] = -1;
        if (grp < ngroups) {
            const int tl = pos / p.Fa, fa = pos - tl * p.Fa;
            p_dst[sidx] = tl * FaP + fa;
            p_cb0[sidx] = grp;
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) {
                const int fb = fa * p.S - p.pad + kf;
                p_src[sidx][kf] = (fb >= 0 && fb < p.Fb) ? tl * rowb + fb : -1;
            }
        }
    }
    float4 pa[MAXV], pb[MAXV];
    auto prefetch = [&](int tile) {
        const int b = tile / ntile_t;
        const int t0 = (tile - b * ntile_t) * TFW;
        const float* srca = p.a + ((long long)b * p.T + t0) * rowa;
        const float* srcb = p.bt + ((long long)b * p.T + (t0 - (p.KT - 1))) * rowb;
#pragma unroll
        for (int q = 0; q < MAXV; ++q) {
            const int i = tid + 256 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < nva && t0 + (i * 4) / rowa < p.T) v = *reinterpret_cast<const float4*>(srca + i * 4);
            pa[q] = v;
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < nvb) {
                const int t = t0 - (p.KT - 1) + (i * 4) / rowb;
                if (t >= 0 && t < p.T) w = *reinterpret_cast<const float4*>(srcb + i * 4);
            }
            pb[q] = w;
        }
    };
    if ((int)blockIdx.x < p.ntiles_total) prefetch(blockIdx.x);
    const int nks = TFW * p.NCH / 4;                         // K steps per tile (TFW*NCH % 4 == 0, host-checked)
    for (int tile = blockIdx.x; tile < p.ntiles_total; tile += gridDim.x) {
        __syncthreads();                                     // previous tile's fragment reads are done
        // A image: [t][ca][fa] -> al[ca][tl][fa];  raw bt frames -> rawl (straight copy)
#pragma unroll
        for (int q = 0; q < MAXV; ++q) {
            if (a_ok[q]) {
                const float vv[4] = {pa[q].x, pa[q].y, pa[q].z, pa[q].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) wput<PREC>(al, aplane, (size_t)a_off[q][u], vv[u]);
            }
            const int i = tid + 256 * q;
            if (i < nvb) *reinterpret_cast<float4*>(rawl + i * 4) = pb[q];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < p.ntiles_total) prefetch(tile + gridDim.x);
        // patch image: pl[j = tap*Cb + cb][tl][fa] = bt[tl + kt][cb][fa*S - pad + kf]
#pragma unroll
        for (int sidx = 0; sidx < MAXP; ++sidx) {
            if (p_dst[sidx] < 0) continue;
            for (int cb = p_cb0[sidx]; cb < p.Cb; cb += ngroups) {
                for (int kt = 0; kt < p.KT; ++kt) {
                    const float* rsrc = rawl + (kt * p.Cb + cb) * p.Fb;
#pragma unroll
                    for (int kf = 0; kf < 3; ++kf) {
                        const int so = p_src[sidx][kf];
                        wput<PREC>(pl, bplane, (size_t)((kt * 3 + kf) * p.Cb + cb) * TFW * FaP + p_dst[sidx],
                                   so >= 0 ? rsrc[so] : 0.f);
                    }
                }
            }
        }
        __syncthreads();
        for (int ks = 0; ks < nks; ++ks) {
            const int chunk = ks * 4 + (lane >> 4);
            const int tl = chunk / p.NCH, fc = chunk - tl * p.NCH;
            const size_t poff = (size_t)tl * FaP + fc * 8;
            Frag<PREC> fa_[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                fa_[i] = wget<PREC>(al, aplane, (size_t)(i * 16 + (lane & 15)) * TFW * FaP + poff);
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int col = (wv * NTW + j) * 16 + (lane & 15);
                const Frag<PREC> fb_ = wget<PREC>(pl, bplane, (size_t)col * TFW * FaP + poff);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j] = mma(fa_[i], fb_, acc[i][j]);
            }
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
                         int B, int T, int Ca, int Fa, int Cb, int Fb, int KT, int S, int pad, int prec,
                         int* nblk_out, hipStream_t stream) {
    const int tfw = (prec == CRUSE_PREC_BF16) ? 8 : 4;
    const int FaP = (Fa + 7) / 8 * 8, NCH = FaP / 8;
    const int ntaps = KT * 3;
    const int mt = Ca <= 16 ? 1 : (Ca <= 32 ? 2 : 4);
    const int ntiles_n = (ntaps * Cb + 15) / 16;
    const int ntw = (ntiles_n + 3) / 4;
    if (Ca > 64 || ntw > 3) return 0;
    if ((tfw * NCH) % 4 != 0) return 0;
    if ((Ca * Fa) % 4 != 0 || (Cb * Fb) % 4 != 0 || ((uintptr_t)a % 16) != 0 || ((uintptr_t)bt % 16) != 0) return 0;
    if (tfw * Ca * Fa > MAXV * 1024 || (tfw + KT - 1) * Cb * Fb > MAXV * 1024) return 0;
    if (tfw * Fa > 768) return 0;
    const int esz = (prec == CRUSE_PREC_F32) ? 4 : 2, npl = (prec == CRUSE_PREC_BF16X3) ? 2 : 1;
    const size_t lds = ((size_t)mt * 16 + (size_t)ntw * 64) * tfw * FaP * esz * npl +
                       (size_t)(tfw + KT - 1) * Cb * Fb * 4;
    if (lds > 150 * 1024) return 0;
    WMArgs p = {};
    p.a = a; p.bt = bt; p.partial = partial;
    p.B = B; p.T = T; p.Ca = Ca; p.Fa = Fa; p.Cb = Cb; p.Fb = Fb; p.KT = KT; p.S = S; p.pad = pad;
    p.FaP = FaP; p.NCH = NCH; p.ntaps = ntaps; p.nrows = tfw + KT - 1;
    p.ntiles_total = B * ((T + tfw - 1) / tfw);
    int grid = p.ntiles_total < max_slabs ? p.ntiles_total : max_slabs;
    if (grid > 512) grid = 512;
    int rc;
    if (prec == CRUSE_PREC_F32) rc = launch_mt<CRUSE_PREC_F32, 4>(p, mt, ntw, grid, lds, stream);
    else if (prec == CRUSE_PREC_BF16X3) rc = launch_mt<CRUSE_PREC_BF16X3, 4>(p, mt, ntw, grid, lds, stream);
    else rc = launch_mt<CRUSE_PREC_BF16, 8>(p, mt, ntw, grid, lds, stream);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cruse_set_error("wgrad_mfma: HIP launch failed: %s", hipGetErrorString(e)); return CRUSE_E_HIP; }
    *nblk_out = grid;
    return 1;
}
