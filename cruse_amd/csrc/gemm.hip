// MFMA GEMM for the GRU gate projections (nn.GRU at model/cruse_net.py:23-31,44,50):
//   gi = x W_ih^T + b_ih          (NT)     dX   = dgi W_ih        (NN)
//   dW_ih += dgi^T x              (TN)     dW_hh += dgh^T h_{t-1} (TN, row-shifted B)
// f32 storage in HBM; tiles are staged through LDS as f32 and turned into MFMA operand
// fragments per precision mode (exact f32 MFMA, split-bf16 x3, or bf16) -- see common.h.
// 128x128x32 block tile, 4 wavefronts as 2x2, each 64x64 = 4x4 MFMA tiles of 16x16.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, BKP = 36;  // BKP: 144-byte rows keep float4 alignment

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc;
    int accumulate, splitk, kchunk, shiftT, vecA, vecB;
};

// stage a [128 rows][32 k] tile of an operand whose K index is CONTIGUOUS in memory
// (element (row, k) at p[row*ld + k]) into registers: 4 float4 per thread.
__device__ __forceinline__ void load_kmajor(const float* p, int ld, int row0, int nrows, int k0, int kend,
                                            int vec, int tid, float4 (&r)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + 256 * q;
        const int row = idx >> 3, c4 = idx & 7;
        const int gr = row0 + row, gk = k0 + c4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < nrows) {
            const float* src = p + (long long)gr * ld + gk;
            if (vec && gk + 3 < kend) {
                v = *reinterpret_cast<const float4*>(src);
            } else {
                if (gk + 0 < kend) v.x = src[0];
                if (gk + 1 < kend) v.y = src[1];
                if (gk + 2 < kend) v.z = src[2];
                if (gk + 3 < kend) v.w = src[3];
            }
        }
        r[q] = v;
    }
}
__device__ __forceinline__ void store_kmajor(float* s, int tid, const float4 (&r)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + 256 * q;
        const int row = idx >> 3, c4 = idx & 7;
        *reinterpret_cast<float4*>(s + row * BKP + c4 * 4) = r[q];
    }
}

// stage a tile of an operand whose ROW index is contiguous (element (row, k) at p[k*ld + row]):
// each thread reads 4 consecutive k for one row (4 coalesced dword loads) and stores one float4.
// shiftT > 0: k-row kk is read from kk-1 and is zero when kk % shiftT == 0.
__device__ __forceinline__ void load_rowmajor(const float* p, int ld, int row0, int nrows, int k0, int kend,
                                              int shiftT, int tid, float4 (&r)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + 256 * q;
        const int row = idx & 127, kq = idx >> 7;      // kq 0..7
        const int gr = row0 + row;
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if (gr < nrows) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int gk = k0 + kq * 4 + j;
                if (gk < kend) {
                    if (shiftT > 0) {
                        if (gk % shiftT == 0) continue;
                        gk -= 1;
                    }
                    e[j] = p[(long long)gk * ld + gr];
                }
            }
        }
        r[q] = make_float4(e[0], e[1], e[2], e[3]);
    }
}
__device__ __forceinline__ void store_rowmajor(float* s, int tid, const float4 (&r)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + 256 * q;
        const int row = idx & 127, kq = idx >> 7;
        *reinterpret_cast<float4*>(s + row * BKP + kq * 4) = r[q];
    }
}

template <int PREC, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[BM * BKP];
    __shared__ __attribute__((aligned(16))) float Bs[BN * BKP];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 ra[4], rb[4];
    auto gload = [&](int k0) {
        if (TA) load_rowmajor(g.A, g.lda, m0, g.M, k0, kend, 0, tid, ra);
        else load_kmajor(g.A, g.lda, m0, g.M, k0, kend, g.vecA, tid, ra);
        if (TB) load_kmajor(g.B, g.ldb, n0, g.N, k0, kend, g.vecB, tid, rb);
        else load_rowmajor(g.B, g.ldb, n0, g.N, k0, kend, g.shiftT, tid, rb);
    };
    if (kbeg < kend) gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        if (TA) store_rowmajor(As, tid, ra); else store_kmajor(As, tid, ra);
        if (TB) store_kmajor(Bs, tid, rb); else store_rowmajor(Bs, tid, rb);
        __syncthreads();
        if (k0 + BK < kend) gload(k0 + BK);
        Frag<PREC> fa[4], fb[4];
        const int ko = (lane >> 4) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* pa = As + (wm * 64 + i * 16 + (lane & 15)) * BKP + ko;
            const float4 a0 = *reinterpret_cast<const float4*>(pa);
            const float4 a1 = *reinterpret_cast<const float4*>(pa + 4);
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            fa[i].set(av);
            const float* pb = Bs + (wn * 64 + i * 16 + (lane & 15)) * BKP + ko;
            const float4 b0 = *reinterpret_cast<const float4*>(pb);
            const float4 b1 = *reinterpret_cast<const float4*>(pb + 4);
            const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            fb[i].set(bv);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mma(fa[i], fb[j], acc[i][j]);
        __syncthreads();
    }

    const bool add_bias = g.bias != nullptr && blockIdx.z == 0;
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
    CRUSE_REQUIRE(prec == CRUSE_PREC_F32 || prec == CRUSE_PREC_BF16X3 || prec == CRUSE_PREC_BF16, CRUSE_E_DTYPE,
                  "gemm: unknown precision %d", prec);
    CRUSE_REQUIRE(b_shift_T == 0 || !transB, CRUSE_E_SHAPE, "gemm: b_shift_T needs transB == 0");
    if (splitk < 1) splitk = 1;
    int kchunk = ((K + splitk - 1) / splitk + BK - 1) / BK * BK;
    splitk = (K + kchunk - 1) / kchunk;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.accumulate = accumulate; g.splitk = splitk; g.kchunk = kchunk; g.shiftT = b_shift_T;
    g.vecA = (!transA && lda % 4 == 0 && ((uintptr_t)A % 16) == 0) ? 1 : 0;
    g.vecB = (transB && ldb % 4 == 0 && ((uintptr_t)B % 16) == 0) ? 1 : 0;
    dim3 grid(cdiv(M, BM), cdiv(N, BN), splitk);
    CRUSE_REQUIRE(grid.y <= 65535 && grid.z <= 65535, CRUSE_E_SHAPE, "gemm: grid too large");
    hipStream_t s = (hipStream_t)stream;
    if (prec == CRUSE_PREC_F32) launch_prec<CRUSE_PREC_F32>(g, transA, transB, grid, s);
    else if (prec == CRUSE_PREC_BF16X3) launch_prec<CRUSE_PREC_BF16X3>(g, transA, transB, grid, s);
    else launch_prec<CRUSE_PREC_BF16>(g, transA, transB, grid, s);
    CRUSE_LAUNCH_CHECK("gemm");
    return CRUSE_OK;
}
