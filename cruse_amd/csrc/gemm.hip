// MFMA GEMM for the GRU gate projections (nn.GRU at model/cruse_net.py:23-31,44,50):
//   gi = x W_ih^T + b_ih          (NT)     dX   = dgi W_ih        (NN)
//   dW_ih += dgi^T x              (TN)     dW_hh += dgh^T h_{t-1} (TN, row-shifted B)
// f32 storage in HBM; tiles are staged through LDS as f32 and turned into MFMA operand
// fragments per precision mode (exact f32 MFMA, split-bf16 x3, or bf16) -- see common.h.
// 128x128x32 block tile, 4 wavefronts as 2x2, each 64x64 = 4x4 MFMA tiles of 16x16.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128;

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc;
    int accumulate, splitk, kchunk, shiftT, vecA, vecB;
    int tiles_m, tiles_n, nunits, inner;     // XCD-aware 1-D grid (see gemm_kernel)
};

// LDS operand storage: f32 keeps floats (BK = 32, 8 x f32 MFMA per fragment pair); bf16 / bf16x3 keep
// 1 / 2 planes of bf16 converted once while staging (BK = 64), so a fragment is one ds_read_b128 per plane.
template <int PREC> struct Tile {
    typedef __bf16 elem;
    static constexpr int BK = 64, PADK = 8, NPL = (PREC == CRUSE_PREC_BF16X3) ? 2 : 1;
};
template <> struct Tile<CRUSE_PREC_F16> {
    typedef _Float16 elem;
    static constexpr int BK = 64, PADK = 8, NPL = 1;
};
template <> struct Tile<CRUSE_PREC_F32> {
    typedef float elem;
    static constexpr int BK = 32, PADK = 4, NPL = 1;
};

template <int PREC>
__device__ __forceinline__ void put4(typename Tile<PREC>::elem* s, int plane, int off, const float4& v) {
    if constexpr (PREC == CRUSE_PREC_F32) {
        *reinterpret_cast<float4*>(s + off) = v;
    } else if constexpr (PREC == CRUSE_PREC_F16) {
        typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
        f16x4 h;
        h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
        *reinterpret_cast<f16x4*>(s + off) = h;
    } else {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        bf16x4 h;
        h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(s + off) = h;
        if constexpr (PREC == CRUSE_PREC_BF16X3) {
            bf16x4 l;
            l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]);
            l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
            *reinterpret_cast<bf16x4*>(s + plane + off) = l;
        }
    }
}

template <int PREC>
__device__ __forceinline__ Frag<PREC> get8(const typename Tile<PREC>::elem* s, int plane, int off) {
    Frag<PREC> f;
    if constexpr (PREC == CRUSE_PREC_F32) {
        const float4 a0 = *reinterpret_cast<const float4*>(s + off);
        const float4 a1 = *reinterpret_cast<const float4*>(s + off + 4);
        f.v[0] = a0.x; f.v[1] = a0.y; f.v[2] = a0.z; f.v[3] = a0.w;
        f.v[4] = a1.x; f.v[5] = a1.y; f.v[6] = a1.z; f.v[7] = a1.w;
    } else if constexpr (PREC == CRUSE_PREC_F16) {
        f.h = *reinterpret_cast<const f16x8_t*>(s + off);
    } else {
        f.h = *reinterpret_cast<const bf16x8*>(s + off);
        if constexpr (PREC == CRUSE_PREC_BF16X3) f.l = *reinterpret_cast<const bf16x8*>(s + plane + off);
    }
    return f;
}

// [128 rows][BK k] tile of an operand whose K index is CONTIGUOUS (element (row,k) at p[row*ld + k]):
// NV float4 per thread, 16 B along k.
template <int BK, int NV>
__device__ __forceinline__ void load_kmajor(const float* p, int ld, int row0, int nrows, int k0, int kend,
                                            int vec, int tid, float4 (&r)[NV]) {
    constexpr int C4 = BK / 4;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int idx = tid + 256 * q;
        const int row = idx / C4, c4 = idx % C4;
        const int gr = row0 + row, gk = k0 + c4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < nrows && gk < kend) {
            const float* src = p + (long long)gr * ld + gk;
            if (vec && gk + 3 < kend) {
                v = *reinterpret_cast<const float4*>(src);
            } else {
                v.x = src[0];
                if (gk + 1 < kend) v.y = src[1];
                if (gk + 2 < kend) v.z = src[2];
                if (gk + 3 < kend) v.w = src[3];
            }
        }
        r[q] = v;
    }
}
template <int PREC, int NV>
__device__ __forceinline__ void store_kmajor(typename Tile<PREC>::elem* s, int plane, int tid, const float4 (&r)[NV]) {
    constexpr int BK = Tile<PREC>::BK, BKP = BK + Tile<PREC>::PADK, C4 = BK / 4;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int idx = tid + 256 * q;
        const int row = idx / C4, c4 = idx % C4;
        put4<PREC>(s, plane, row * BKP + c4 * 4, r[q]);
    }
}

// operand whose ROW index is contiguous (element (row,k) at p[k*ld + row]): a thread owns 4x4 blocks
// (4 rows x 4 k): four 16-byte loads along the rows (8 lanes = one 128-byte line), transposed in
// registers, stored as four 4-k groups.  shiftT > 0: k-row kk is read from kk-1 and is zero when
// kk % shiftT == 0 (the h_{t-1} operand of dW_hh).
template <int BK>
__device__ __forceinline__ void rm_block(int blk, int& rq, int& kq) {
    constexpr int KQ = BK / 4;
    rq = (blk & 7) + 8 * (blk / (8 * KQ));
    kq = (blk >> 3) % KQ;
}
template <int BK, int NV>
__device__ __forceinline__ void load_rowmajor(const float* p, int ld, int row0, int nrows, int k0, int kend,
                                              int shiftT, int vec, int tid, float4 (&r)[NV]) {
#pragma unroll
    for (int q = 0; q < NV / 4; ++q) {
        int rq, kq;
        rm_block<BK>(tid + 256 * q, rq, kq);
        const int gr = row0 + rq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int gk = k0 + kq * 4 + j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            bool ok = gk < kend && gr < nrows;
            if (ok && shiftT > 0) {
                if (gk % shiftT == 0) ok = false;
                gk -= 1;
            }
            if (ok) {
                const float* src = p + (long long)gk * ld + gr;
                if (vec && gr + 3 < nrows) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0];
                    if (gr + 1 < nrows) v.y = src[1];
                    if (gr + 2 < nrows) v.z = src[2];
                    if (gr + 3 < nrows) v.w = src[3];
                }
            }
            r[q * 4 + j] = v;
        }
    }
}
template <int PREC, int NV>
__device__ __forceinline__ void store_rowmajor(typename Tile<PREC>::elem* s, int plane, int tid, const float4 (&r)[NV]) {
    constexpr int BK = Tile<PREC>::BK, BKP = BK + Tile<PREC>::PADK;
#pragma unroll
    for (int q = 0; q < NV / 4; ++q) {
        int rq, kq;
        rm_block<BK>(tid + 256 * q, rq, kq);
        const float4 a = r[q * 4 + 0], b = r[q * 4 + 1], c = r[q * 4 + 2], d = r[q * 4 + 3];
        put4<PREC>(s, plane, (rq * 4 + 0) * BKP + kq * 4, make_float4(a.x, b.x, c.x, d.x));
        put4<PREC>(s, plane, (rq * 4 + 1) * BKP + kq * 4, make_float4(a.y, b.y, c.y, d.y));
        put4<PREC>(s, plane, (rq * 4 + 2) * BKP + kq * 4, make_float4(a.z, b.z, c.z, d.z));
        put4<PREC>(s, plane, (rq * 4 + 3) * BKP + kq * 4, make_float4(a.w, b.w, c.w, d.w));
    }
}

template <int PREC, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    typedef typename Tile<PREC>::elem elem;
    constexpr int BK = Tile<PREC>::BK, BKP = BK + Tile<PREC>::PADK, NPL = Tile<PREC>::NPL;
    constexpr int NV = BM * BK / 4 / 256;
    constexpr int PLANE = BM * BKP;
    __shared__ __attribute__((aligned(16))) elem As[NPL * PLANE];
    __shared__ __attribute__((aligned(16))) elem Bs[NPL * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    // XCD-aware tile order (block id % 8 -> XCD is the observed dispatch; speed only).  A UNIT is the set of
    // tiles that share an operand panel: the n-tiles of one m-tile (A panel) or, with split-K, all tiles of one
    // k-slice.  Unit u runs entirely on XCD u % 8, its tiles back to back, so the shared panel is fetched into
    // that XCD's L2 once instead of once per tile (PMC: 1.0 GB -> see profiles/ for the NT gate projection).
    const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3;
    const int inner = qq % g.inner, unit = (qq / g.inner) * 8 + xcd;
    if (unit >= g.nunits) return;
    int tm, tn, tz;
    if (g.splitk > 1) { tz = unit; tm = inner % g.tiles_m; tn = inner / g.tiles_m; }
    else { tz = 0; tm = unit; tn = inner; }
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = tz * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 ra[NV], rb[NV];
    auto gload = [&](int k0) {
        if (TA) load_rowmajor<BK, NV>(g.A, g.lda, m0, g.M, k0, kend, 0, g.vecA, tid, ra);
        else load_kmajor<BK, NV>(g.A, g.lda, m0, g.M, k0, kend, g.vecA, tid, ra);
        if (TB) load_kmajor<BK, NV>(g.B, g.ldb, n0, g.N, k0, kend, g.vecB, tid, rb);
        else load_rowmajor<BK, NV>(g.B, g.ldb, n0, g.N, k0, kend, g.shiftT, g.vecB, tid, rb);
    };
    if (kbeg < kend) gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        if (TA) store_rowmajor<PREC, NV>(As, PLANE, tid, ra); else store_kmajor<PREC, NV>(As, PLANE, tid, ra);
        if (TB) store_kmajor<PREC, NV>(Bs, PLANE, tid, rb); else store_rowmajor<PREC, NV>(Bs, PLANE, tid, rb);
        __syncthreads();
        if (k0 + BK < kend) gload(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 32) {
            Frag<PREC> fa[4], fb[4];
            const int ko = kk + (lane >> 4) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = get8<PREC>(As, PLANE, (wm * 64 + i * 16 + (lane & 15)) * BKP + ko);
                fb[i] = get8<PREC>(Bs, PLANE, (wn * 64 + i * 16 + (lane & 15)) * BKP + ko);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma(fa[i], fb[j], acc[i][j]);
        }
        __syncthreads();
    }

    const bool add_bias = g.bias != nullptr && tz == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane & 15);
            if (n >= g.N) continue;
            const float bv = add_bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                if (m >= g.M) continue;
                float* c = g.C + (long long)m * g.ldc + n;
                const float v = acc[i][j][r] + bv;
                if (g.splitk > 1) atomicAdd(c, v);
                else if (g.accumulate) *c += v;
                else *c = v;
            }
        }
    }
}

template <int PREC>
void launch_prec(const GemmArgs& g, int transA, int transB, dim3 grid, hipStream_t s) {
    if (!transA && transB) hipLaunchKernelGGL((gemm_kernel<PREC, false, true>), grid, dim3(256), 0, s, g);
    else if (!transA && !transB) hipLaunchKernelGGL((gemm_kernel<PREC, false, false>), grid, dim3(256), 0, s, g);
    else if (transA && !transB) hipLaunchKernelGGL((gemm_kernel<PREC, true, false>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_kernel<PREC, true, true>), grid, dim3(256), 0, s, g);
}

}  // namespace

extern "C" int cruse_gemm(int transA, int transB, int M, int N, int K,
                          const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                          const float* bias, int accumulate, int splitk, int b_shift_T, int prec, void* stream) {
    CRUSE_REQUIRE(M > 0 && N > 0 && K > 0, CRUSE_E_SHAPE, "gemm: empty shape M=%d N=%d K=%d", M, N, K);
    CRUSE_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, CRUSE_E_SHAPE,
                  "gemm: leading dimensions lda=%d ldb=%d ldc=%d too small", lda, ldb, ldc);
    CRUSE_REQUIRE(prec == CRUSE_PREC_F32 || prec == CRUSE_PREC_BF16X3 || prec == CRUSE_PREC_BF16 || prec == CRUSE_PREC_F16, CRUSE_E_DTYPE,
                  "gemm: unknown precision %d", prec);
    CRUSE_REQUIRE(b_shift_T == 0 || !transB, CRUSE_E_SHAPE, "gemm: b_shift_T needs transB == 0");
    if (splitk < 1) splitk = 1;
    const int BK = 64;   // multiple of both tile depths (32 for f32, 64 for bf16)
    int kchunk = ((K + splitk - 1) / splitk + BK - 1) / BK * BK;
    splitk = (K + kchunk - 1) / kchunk;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.accumulate = accumulate; g.splitk = splitk; g.kchunk = kchunk; g.shiftT = b_shift_T;
    g.vecA = (lda % 4 == 0 && ((uintptr_t)A % 16) == 0) ? 1 : 0;   // 16-byte loads along the contiguous index
    g.vecB = (ldb % 4 == 0 && ((uintptr_t)B % 16) == 0) ? 1 : 0;
    g.tiles_m = cdiv(M, BM); g.tiles_n = cdiv(N, BN);
    if (splitk > 1) { g.nunits = splitk; g.inner = g.tiles_m * g.tiles_n; }
    else { g.nunits = g.tiles_m; g.inner = g.tiles_n; }
    const long long nblk = (long long)cdiv(g.nunits, 8) * 8 * g.inner;
    CRUSE_REQUIRE(nblk < (1ll << 31), CRUSE_E_SHAPE, "gemm: grid too large");
    dim3 grid((unsigned)nblk);
    hipStream_t s = (hipStream_t)stream;
    if (prec == CRUSE_PREC_F32) launch_prec<CRUSE_PREC_F32>(g, transA, transB, grid, s);
    else if (prec == CRUSE_PREC_BF16X3) launch_prec<CRUSE_PREC_BF16X3>(g, transA, transB, grid, s);
    else if (prec == CRUSE_PREC_F16) launch_prec<CRUSE_PREC_F16>(g, transA, transB, grid, s);
    else launch_prec<CRUSE_PREC_BF16>(g, transA, transB, grid, s);
    CRUSE_LAUNCH_CHECK("gemm");
    return CRUSE_OK;
}
