// bf16-operand MFMA GEMM for the GRU gate projections in CRUSE_PREC_BF16 mode (nn.GRU at
// model/cruse_net.py:23-31,44,50), plus the layout kernels that feed it:
//   C[M,N] (+)= A[M,K] . B[N,K]^T (+ bias[n])          A, B bf16 with K contiguous, C f32
// Every gate GEMM of the step is brought to this one NT form by keeping bf16 copies of the operands in the
// orientation the product needs (row-major for gi / dX, time-major transposes for the weight gradients):
//   gi     = x_bf   . W_ih_bf^T            dX     = dgi_bf . (W_ih^T)_bf^T
//   dW_ih += dgT    . (x^T)_bf^T           dW_hh += dgT    . (h_{t-1}^T)_bf^T
// The values are the same RNE-rounded bf16 operands the f32-input kernel in gemm.hip forms on the fly, so
// both kernels produce the same products; this one moves half the bytes and stages tiles with LDS-DMA.
//
// 128x128x64 block tile, 4 wavefronts as 2x2 (64x64 = 4x4 MFMA 16x16x32 tiles each), two LDS buffers filled by
// global_load_lds (16 B per lane, 1 KiB per wave instruction).  The LDS image of a tile is [128 rows][8 chunks
// of 16 B]; chunk c of row r sits at position c ^ (r & 7) -- the swizzle is applied on the per-lane global
// SOURCE address (the LDS-DMA destination is lane-linear) and undone in the fragment read address.
#include "common.h"
#include <stdlib.h>

extern "C" int cruse_cast_bf16_split(const float* x, void* y, void* y_lo, long long n, void* stream);

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;      // 16 KiB

typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef const __attribute__((address_space(1))) void glb_ptr_t;

struct GbArgs {
    const __bf16* A; const __bf16* B; float* C; const float* bias;
    long long a_lo, b_lo;                    // element offsets of the LOW planes (split-bf16 x3 form), 0 = plain bf16
    int M, N, K;
    long long lda, ldb, ldc, a_ks, b_ks;     // *_ks: elements between consecutive 64-deep k-tiles of one row
    int accumulate, splitk, kt_chunk;
    int tiles_m, tiles_n, nunits, inner;
    int xcdk;                                // split-K with the k-slices PINNED to XCDs (see the kernel's tile order)
    long long slab;                          // MODE 4: elements between the partial-sum slabs of consecutive k-slices
    // ROW SEGMENTS on the M side: logical row m of A and C is physical row (m / seg_len) * seg_stride + seg_off + m % seg_len --
    // frames [seg_off, seg_off + seg_len) of every clip of a [B, T = seg_stride] tensor, i.e. a TIME CHUNK of the batch
    // (the gate projections of one chunk run beside the recurrence of the previous one).  seg_len == 0: identity.
    int seg_len; long long seg_stride, seg_off;
    // CONCATENATED products (cruse_gemm_bf16_nt_slabs_cat): row tiles [0, tm_b1) are product 0 (A, B as above), [tm_b1, tm_b2) product 1,
    // [tm_b2, ..) product 2 -- each with its own B operand and its own first A row (a_shift*: elements added to the A address of the
    // VIRTUAL row), all sharing N, K, the strides and one output [sum of the M_i, N].  tm_b1 = tm_b2 = INT_MAX: one product.
    int tm_b1, tm_b2; long long a_shift1, a_shift2; const __bf16* B1; const __bf16* B2;
    // (products whose M_i is not a multiple of 128 occupy whole row tiles all the same: m_end* = end of product's VIRTUAL rows -- rows past it are
    // computed and dropped --, c_rows* = output row of a virtual row minus that virtual row)
    int m_end0, m_end1, m_end2; int c_rows1, c_rows2;
    // GROUPS along N (cruse_gemm_bf16_nt_groups): the column tiles [q * tn_per_g, (q + 1) * tn_per_g) are product q -- N columns each, A columns
    // a_gstep further right, B / bias / C columns b_gstep / bias_gstep / c_gstep elements further on -- the GRU groups of one layer in ONE launch.
    int tn_per_g; long long a_gstep, b_gstep, c_gstep, bias_gstep;
    // ATR (cruse_gemm_bf16_nt_atr): A is read from its TIME-MAJOR K-tiled image -- element (m, k) at A[(m / 64) * a_mbs + k * 64 + m % 64], the
    // layout of the gate-gradient tensor dgT the weight-gradient GEMMs consume -- so the row-major copy dgi need not exist.  a_ks = 64 * 64.
    long long a_mbs; int a_mb_last;
    int c_f16;                               // MODE 3: the 2-byte result rows are IEEE f16 (0: bf16)
};

__device__ __forceinline__ long long seg_row(const GbArgs& g, int m) {
    if (g.seg_len == 0) return m;
    const int q = m / g.seg_len;
    return (long long)q * g.seg_stride + g.seg_off + (m - q * g.seg_len);
}

// MODE: C update -- 0 store, 1 read-add-store, 2 atomic add (split-K), 3 store as bf16 (C is then a bf16 tensor, ldc in its elements),
// 4 split-K with every k-slice STORING its partial sums to its own slab C + tz * slab (summed by gemm_slab_reduce_kernel): the
// f32 atomics of MODE 2 from the 8 XCDs meet at the memory side -- 9.8 M of them per weight-gradient product, 0.14 ms per step.
// NST: LDS stages.  2 = one k-tile of prefetch, two blocks per CU (the short-K products, whose epilogues then
// overlap the other block's k-loop); 3 = two k-tiles in flight behind counted vmcnt waits and one raw barrier per
// k-tile, one block per CU (the split-K weight gradients: hundreds of k-tiles streamed once, where the k-loop is
// bound by the latency of the next tile's loads).
// BMT: 1 = 128 x 128 tiles on 4 wavefronts; 2 = 256 x 128 tiles on 8 wavefronts (4 x 2, the same 64 x 64 block per wave) with THREE
// stages -- 96 KB of operands in flight per CU instead of 64, 48 KB instead of 64 KB per 256 x 128 x 64 of work.  MEASURED SLOWER and off
// (library option gb_bm256): gi 96 -> 118 us, dX 85 -> 105 us alone (r4): one 8-wave block per CU loses more at its barriers than two
// independent 4-wave blocks that fill each other's stalls.  Alone these two products run at 0.66 / 0.74 PFLOP/s; their 150-165 us in the
// step are cold operands and the side stream's traffic, not this loop.
// F16: the operands are IEEE f16 (same bytes per element, same staging and fragment layout; v_mfma_f32_16x16x32_f16): the forward gate
// projection as ONE pass -- 11 significant bits on both operands against 8 + the W_ih low-plane pass of the split-bf16 form (cruse_gemm_f16_nt).
// ATR: the A tile is staged from a k-major source ([64 k][64 m] per 64-row block: 8 KB contiguous per block and k-tile, lane-linear LDS-DMA as
// ever) and its MFMA fragments are TRANSPOSING LDS reads: ds_read_b64_tr_b16 hands lane c of a 16-lane group the four k of column c of a [4 k][16 m]
// block whose rows the lanes 4 r .. 4 r + 3 address (measured, tools/probes/tr16_probe.hip) -- two reads per fragment.  The k-rows of the image are 128
// bytes (32 banks) apart, and a 32-lane service group reads rows {0..3} + 8 kg of one parity class each: the 16-byte chunks of row k are XOR-swizzled
// (on the global SOURCE address, as for the row-major image) by ((k >> 1 & 1) | (k >> 3 & 1) << 1) << 1, which puts the four same-parity rows of a
// group on four different 32-byte bank windows -- conflict-free.
template <int MODE, int NST, int BMT = 1, bool F16 = false, bool ATR = false, bool PIPE = false>
__global__ __launch_bounds__(256 * BMT, (NST == 2 && BMT == 1) ? 2 : 1) void gemm_bf16_nt_kernel(const GbArgs g) {
    static_assert(!ATR || (BMT == 1 && !F16), "transposed-A staging: 128-row tiles, bf16");
    constexpr int BM_ = BM * BMT, A_BYTES = TILE_BYTES * BMT, STAGE_BYTES = A_BYTES + TILE_BYTES;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_dyn[];
    auto As_ = [&](int buf) -> unsigned char* { return smem_dyn + (size_t)buf * STAGE_BYTES; };            // [stage][A | B]
    auto Bs_ = [&](int buf) -> unsigned char* { return smem_dyn + (size_t)buf * STAGE_BYTES + A_BYTES; };
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    // XCD-aware tile order, as in gemm.hip: a unit -- the n-tiles of one m-tile, or with split-K the m-tiles of one
    // (k-slice, n-tile) -- stays on one XCD, so the operand panel its tiles share is fetched into that L2 once
    // xcdk (the K = 25 664 weight gradients): k-slice z is computed ENTIRELY on XCD z % 8 -- all its output tiles, m-major, so
    // the blocks resident on that XCD at one time walk the same k-range together and share every operand k-tile through
    // that XCD's L2: each operand byte leaves HBM about once.  (With (k-slice, n-tile) units spread round-robin the big
    // operand was fetched once per n-tile: 506 MB per launch for 131 MB of operands, PMC r01.)
    const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3;
    int tm, tn, tz;
    if (g.xcdk) {
        const int ntile = g.tiles_m * g.tiles_n;
        tz = (qq / ntile) * 8 + xcd;
        if (tz >= g.splitk) return;
        const int tl = qq % ntile;
        tm = tl / g.tiles_n; tn = tl % g.tiles_n;
    } else {
        const int inner = qq % g.inner, unit = (qq / g.inner) * 8 + xcd;
        if (unit >= g.nunits) return;
        if (g.splitk > 1) { tz = unit / g.tiles_n; tn = unit % g.tiles_n; tm = inner; }
        else { tz = 0; tm = unit; tn = inner; }
    }
    const int gq = g.tn_per_g > 0 ? tn / g.tn_per_g : 0;                  // product (GRU group) of this column tile
    if (g.tn_per_g > 0) tn -= gq * g.tn_per_g;
    const int m0 = tm * BM_, n0 = tn * BN;
    const __bf16* Abase = g.A + gq * g.a_gstep;
    const __bf16* Bbase = g.B + gq * g.b_gstep;
    int m_lim = g.m_end0, c_sh = 0;
    if (tm >= g.tm_b2) { Abase += g.a_shift2; Bbase = g.B2; m_lim = g.m_end2; c_sh = g.c_rows2; }
    else if (tm >= g.tm_b1) { Abase += g.a_shift1; Bbase = g.B1; m_lim = g.m_end1; c_sh = g.c_rows1; }
    const int nkt = g.K / BK;
    const int kt0 = tz * g.kt_chunk, kt1 = min(nkt, kt0 + g.kt_chunk);
    // split-bf16 forms: the k-range is walked again on the same accumulators for every correction term -- (A_hi, B_lo)
    // when B has a low plane, (A_lo, B_hi) when A has one too -- as extra virtual k-tiles of ONE staging pipeline
    const int nreal = kt1 - kt0;
    const int nvirt = (g.b_lo ? (g.a_lo ? 3 : 2) : 1) * nreal;      // (hi,hi) [, (hi,lo) [, (lo,hi)]]

    // staging: wave wv fills rows [wv*32, wv*32+32) of the A tile and its share of the B tile (32 rows of 128 on 4 waves, 16 on 8),
    // 8 rows (1 KiB) per instruction
    constexpr int NB_I = 4 / BMT;
    const __bf16* ap[4];
    const __bf16* bp[NB_I];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (ATR) {
            const int j = wv * 4 + i;                            // instruction j fills k-rows (j & 7) * 8 .. + 8 of 64-row block j >> 3
            const int krow = (j & 7) * 8 + (lane >> 3);
            const int ch = (lane & 7) ^ ((((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 1);
            const int mb = min(2 * tm + (j >> 3), g.a_mb_last);  // (rows past M: the last block again -- computed, never stored)
            ap[i] = Abase + (long long)mb * g.a_mbs + krow * 64 + ch * 8;
        } else {
            const int r = (wv * 4 + i) * 8 + (lane >> 3);
            const int ch = (lane & 7) ^ (lane >> 3);
            ap[i] = Abase + seg_row(g, min(m0 + r, m_lim - 1)) * g.lda + ch * 8;
        }
    }
#pragma unroll
    for (int i = 0; i < NB_I; ++i) {
        const int r = (wv * NB_I + i) * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ (lane >> 3);
        bp[i] = Bbase + (long long)min(n0 + r, g.N - 1) * g.ldb + ch * 8;
    }
    auto stage = [&](int v, int buf) {               // v: virtual k-tile index in [0, nvirt)
        const int seg = v >= 2 * nreal ? 2 : (v >= nreal ? 1 : 0);
        const int kt = kt0 + v - seg * nreal;
        const long long oa = (long long)kt * g.a_ks + (seg == 2 ? g.a_lo : 0);
        const long long ob = (long long)kt * g.b_ks + (seg == 1 ? g.b_lo : 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(ap[i] + oa), (lds_ptr_t*)(As_(buf) + (wv * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NB_I; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(bp[i] + ob), (lds_ptr_t*)(Bs_(buf) + (wv * NB_I + i) * 1024), 16, 0, 0);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment byte offsets inside a tile: row (lane & 15) of each 16-row sub-tile, chunk kk*4 + (lane >> 4)
    int offa[2], offb[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int c = (kk * 4 + (lane >> 4)) ^ (lane & 7);
        offa[kk] = (wm * 64 + (lane & 15)) * 128 + c * 16;
        offb[kk] = (wn * 64 + (lane & 15)) * 128 + c * 16;
    }

    // ATR: byte offset of this lane's 8-byte piece for sub-tile i (k-row kg * 8 + q of the half (kk, h) added as a constant)
    int offt[4];
    if constexpr (ATR) {
        const int c15 = lane & 15, q = c15 >> 2, kg = lane >> 4;
        const int gsw = ((q >> 1) & 1) | ((kg & 1) << 1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            offt[i] = wm * 8192 + (kg * 8 + q) * 128 + (((i ^ gsw) * 2 + ((c15 & 3) >> 1)) * 16) + (c15 & 1) * 8;
    }

    // One k-tile = two 32-deep halves of 16 MFMAs per wave.  CRUSE_GB_PIPE: the fragment reads of both halves first, the order then pinned with
    // sched_group_barrier (reads of half 0, one read of half 1 behind each of the first MFMAs of half 0, the remaining MFMAs) -- left to itself the
    // scheduler re-uses the A fragment registers and waits with lgkmcnt(0) in front of every group of 4 to 8 MFMAs, which measured no slower.
    auto compute_pipe = [&](int buf) {
        const unsigned char* As = As_(buf);
        const unsigned char* Bs = Bs_(buf);
        bf16x8 fa[2][4], fb[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (ATR) {
                    typedef __attribute__((ext_vector_type(4))) short s16x4_;
                    typedef __attribute__((address_space(3))) s16x4_ lds_s16x4;
                    typedef __attribute__((ext_vector_type(8))) short s16x8_;
                    const s16x4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(As + offt[i] + kk * 32 * 128));
                    const s16x4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(As + offt[i] + kk * 32 * 128 + 4 * 128));
                    const s16x8_ both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    fa[kk][i] = __builtin_bit_cast(bf16x8, both);
                } else {
                    fa[kk][i] = *reinterpret_cast<const bf16x8*>(As + offa[kk] + i * 16 * 128);
                }
                fb[kk][i] = *reinterpret_cast<const bf16x8*>(Bs + offb[kk] + i * 16 * 128);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if constexpr (F16) {
                        typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_, fa[kk][i]), __builtin_bit_cast(f16x8_, fb[kk][j]),
                                                                           acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
                    }
        constexpr int NLH = ATR ? 12 : 8;                    // LDS reads per half
        __builtin_amdgcn_sched_group_barrier(0x100, NLH, 0);
#pragma unroll
        for (int r = 0; r < NLH; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 32 - NLH, 0);
    };
    auto compute_plain = [&](int buf) {
        const unsigned char* As = As_(buf);
        const unsigned char* Bs = Bs_(buf);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (ATR) {
                    typedef __attribute__((ext_vector_type(4))) short s16x4_;
                    typedef __attribute__((address_space(3))) s16x4_ lds_s16x4;
                    typedef __attribute__((ext_vector_type(8))) short s16x8_;
                    const s16x4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(As + offt[i] + kk * 32 * 128));
                    const s16x4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(As + offt[i] + kk * 32 * 128 + 4 * 128));
                    const s16x8_ both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    fa[i] = __builtin_bit_cast(bf16x8, both);
                } else {
                    fa[i] = *reinterpret_cast<const bf16x8*>(As + offa[kk] + i * 16 * 128);
                }
                fb[i] = *reinterpret_cast<const bf16x8*>(Bs + offb[kk] + i * 16 * 128);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if constexpr (F16) {
                        typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_, fa[i]), __builtin_bit_cast(f16x8_, fb[j]),
                                                                           acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                    }
        }
    };
    auto compute = [&](int buf) { if constexpr (PIPE) compute_pipe(buf); else compute_plain(buf); };
    if constexpr (NST == 2) {
        if (nvirt > 0) stage(0, 0);
        for (int v = 0; v < nvirt; ++v) {
            const int buf = v & 1;
            __syncthreads();                   // tile v landed (vmcnt(0) + barrier); buffer buf^1 is free again
            if (v + 1 < nvirt) stage(v + 1, buf ^ 1);
            compute(buf);
        }
    } else {
        // each stage() is 8 (BMT 2: 6) LDS-DMA loads per wave, retired in order: vmcnt(8 / 6) == "all but the newest tile landed"
        if (nvirt > 0) stage(0, 0);
        if (nvirt > 1) stage(1, 1);
        int buf = 0;
        for (int v = 0; v < nvirt; ++v) {
            if (v + 1 < nvirt) { if constexpr (BMT == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // tile v is in LDS for every wave; stage (v+2)%3 == (v-1)%3 is drained
            if (v + 2 < nvirt) stage(v + 2, buf == 0 ? 2 : buf - 1);
            compute(buf);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            buf = buf == 2 ? 0 : buf + 1;
        }
    }

    // epilogue: each wave turns its 64x64 accumulator block into row-contiguous float4 accesses through its own
    // 32 x 68-float LDS patch (two halves), so C is written in 256-byte row segments
    __syncthreads();                           // everyone is done with the operand tiles
    float* patch = reinterpret_cast<float*>(smem_dyn) + wv * (32 * 68);
    const bool add_bias = g.bias != nullptr && tz == 0;
    const int nq = n0 + wn * 64 + (lane & 15) * 4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (add_bias) {
        const float* bias = g.bias + gq * g.bias_gstep;
        bv.x = nq < g.N ? bias[nq] : 0.f; bv.y = nq + 1 < g.N ? bias[nq + 1] : 0.f;
        bv.z = nq + 2 < g.N ? bias[nq + 2] : 0.f; bv.w = nq + 3 < g.N ? bias[nq + 3] : 0.f;
    }
    const bool vec = (g.ldc % 4 == 0) && (g.c_gstep % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0) && nq + 3 < g.N;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    patch[(ii * 16 + (lane >> 4) * 4 + r) * 68 + j * 16 + (lane & 15)] = acc[half * 2 + ii][j][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int rr = p * 4 + (lane >> 4);
            const int m = m0 + wm * 64 + half * 32 + rr;
            float4 v = *reinterpret_cast<const float4*>(patch + rr * 68 + (lane & 15) * 4);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (m < m_lim) {
                if constexpr (MODE == 3) {                 // bf16 rows: 8-byte stores, 128-byte row segments
                    __bf16* cb = reinterpret_cast<__bf16*>(g.C) + seg_row(g, m) * g.ldc + nq;
                    if (g.c_f16) {                          // (same rows, IEEE f16 elements: the gi rows cruse_gru_seq_fwd_gi16 reads)
                        _Float16* ch = reinterpret_cast<_Float16*>(cb);
                        if (nq + 3 < g.N && (g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 7) == 0)) {
                            typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_;
                            const f16x4_ o = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                            *reinterpret_cast<f16x4_*>(ch) = o;
                        } else {
                            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (nq + q < g.N) ch[q] = (_Float16)e[q];
                        }
                        continue;
                    }
                    if (nq + 3 < g.N && (g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 7) == 0)) {
                        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
                        const bf16x4_ o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                        *reinterpret_cast<bf16x4_*>(cb) = o;
                    } else {
                        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (nq + q < g.N) cb[q] = (__bf16)e[q];
                    }
                    continue;
                }
                float* c = g.C + (MODE == 4 ? (long long)tz * g.slab : 0ll) + (seg_row(g, m) + c_sh) * g.ldc + gq * g.c_gstep + nq;
                if (MODE == 2) {
                    if (nq < g.N) atomicAdd(c, v.x);
                    if (nq + 1 < g.N) atomicAdd(c + 1, v.y);
                    if (nq + 2 < g.N) atomicAdd(c + 2, v.z);
                    if (nq + 3 < g.N) atomicAdd(c + 3, v.w);
                } else if (vec) {
                    if (MODE == 1) {
                        const float4 o = *reinterpret_cast<const float4*>(c);
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    *reinterpret_cast<float4*>(c) = v;
                } else {
                    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (nq + q < g.N) c[q] = (MODE == 1 ? c[q] : 0.f) + e[q];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// y = bf16(x), same layout; y_lo (optional) = bf16(x - y): the low plane of the split-bf16 x3 form
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* x, __bf16* y, __bf16* y_lo, long long n4) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float f[4] = {v.x, v.y, v.z, v.w};
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = (__bf16)f[e]; l[e] = (__bf16)(f[e] - (float)h[e]); }
        reinterpret_cast<bf16x4*>(y)[i] = h;
        if (y_lo) reinterpret_cast<bf16x4*>(y_lo)[i] = l;
    }
}

// Time-major K-TILED transpose: element (c, r) of x^T lives at yT[(r/64)*cols*64 + c*64 + r%64], i.e. the 64-frame
// k-tile of every line is one 128-byte run and a whole k-tile of the operand is one contiguous block (a row-major
// [cols][ldT] image made every k-tile of a GEMM operand tile touch 128 pages 51 KB apart).  Values:
// bf16(x[r - shift][c]) for r < rows (0 where shift and r is the first frame of its clip), 0 for rows <= r < ldT.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const float* x, long long rows, int cols, long long ld,
                                                             __bf16* yT, long long ldT, int shiftT) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    const long long r0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    // in: 16-byte loads along the columns where the rows allow it (4 per thread), 4-byte ones otherwise
    const bool vec = (ld & 3) == 0 && (cols & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (vec) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int idx = p * 256 + tid, rr = idx >> 4, c4 = (idx & 15) * 4;
            const long long r = r0 + rr;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rows && c0 + c4 < cols) {
                if (shiftT == 0) v = *reinterpret_cast<const float4*>(x + r * ld + c0 + c4);
                else if (r % shiftT != 0) v = *reinterpret_cast<const float4*>(x + (r - 1) * ld + c0 + c4);
            }
            tile[rr][c4] = v.x; tile[rr][c4 + 1] = v.y; tile[rr][c4 + 2] = v.z; tile[rr][c4 + 3] = v.w;
        }
    } else {
        const int cc = tid & 63;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int rr = p * 4 + (tid >> 6);
            const long long r = r0 + rr;
            float v = 0.f;
            if (r < rows && c0 + cc < cols) {
                if (shiftT == 0) v = x[r * ld + c0 + cc];
                else if (r % shiftT != 0) v = x[(r - 1) * ld + c0 + cc];
            }
            tile[rr][cc] = v;
        }
    }
    __syncthreads();
    {
        // out: a lane writes 8 consecutive rows of one column = 16 bytes; 8 lanes cover the 128 bytes of a column's 64 rows
        typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_;
        const int j = tid & 7;                   // row group
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int cc = p * 32 + (tid >> 3);
            if (c0 + cc < cols) {
                bf16x8_ h;
#pragma unroll
                for (int e = 0; e < 8; ++e) h[e] = (__bf16)tile[8 * j + e][cc];
                *reinterpret_cast<bf16x8_*>(yT + (r0 >> 6) * ((long long)cols * 64) + (long long)(c0 + cc) * 64 + 8 * j) = h;
            }
        }
    }
}

// K-tiling WITHOUT transposition: element (n, k) of x [rows n][cols k] goes to y[(k/64)*rows*64 + n*64 + k%64];
// k in [cols, kp) is zero-filled (kp = cols rounded up to 64) -- the K-contiguous operand whose K is not a
// multiple of 64 (W_ih with Hg = 160)
template <bool F16>
__global__ __launch_bounds__(256) void ktile_bf16_kernel(const float* x, int rows, int cols, long long ld, __bf16* y,
                                                         __bf16* y_lo, int kp) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
    const int kq = kp >> 2;
    const long long n4 = (long long)rows * kq;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const int n = (int)(i / kq), k = (int)(i - (long long)n * kq) * 4;
        const long long o = (long long)(k >> 6) * rows * 64 + (long long)n * 64 + (k & 63);
        if constexpr (F16) {                     // (y_lo: the low plane f16(x - f16(x)) of the two-pass form cruse_gemm_f16x2_nt)
            f16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float f = (k + e < cols) ? x[n * ld + k + e] : 0.f;
                h[e] = (_Float16)f; l[e] = (_Float16)(f - (float)h[e]);
            }
            *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(y) + o) = h;
            if (y_lo) *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(y_lo) + o) = l;
        } else {
            bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float f = (k + e < cols) ? x[n * ld + k + e] : 0.f;
                h[e] = (__bf16)f; l[e] = (__bf16)(f - (float)h[e]);
            }
            *reinterpret_cast<bf16x4*>(y + o) = h;
            if (y_lo) *reinterpret_cast<bf16x4*>(y_lo + o) = l;
        }
    }
}

// C[m*ldc + n] += sum_z slabs[z*slab + m*N + n]: the k-slices' partial sums in a fixed order (run-to-run reproducible)
__global__ __launch_bounds__(256) void gemm_slab_reduce_kernel(const float* slabs, int nz, long long slab, int M, int N, float* C,
                                                               long long ldc) {
    const int nq = N >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)M * nq) return;
    const int m = (int)(i / nq), q = (int)(i - (long long)m * nq);
    const float* p = slabs + (long long)m * N + 4 * q;
    float4 s = *reinterpret_cast<const float4*>(p);
    for (int z = 1; z < nz; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(p + (long long)z * slab);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* c = C + (long long)m * ldc + 4 * q;
    c[0] += s.x; c[1] += s.y; c[2] += s.z; c[3] += s.w;
}

}  // namespace

struct GbGroups { int G; long long a_gstep, b_gstep, c_gstep, bias_gstep; };
struct GbCat { int tm_b1, tm_b2; long long a_shift1, a_shift2; const __bf16* B1; const __bf16* B2; int m_end[3], c_rows[3], M_out; };

static int gemm_bf16_impl(int M, int N, int K, const void* A, const void* A_lo, long long lda, long long a_kstride,
                          const void* B, const void* B_lo, long long ldb, long long b_kstride,
                          float* C, long long ldc, const float* bias, int accumulate, int splitk, void* stream,
                          int seg_len = 0, long long seg_stride = 0, long long seg_off = 0, bool c_bf16 = false,
                          float* slabs = nullptr, size_t slab_bytes = 0, bool f16 = false, const GbCat* cat = nullptr,
                          long long atr_mbs = 0, int atr_mb_last = 0, const GbGroups* grp = nullptr, bool c_f16 = false) {
    CRUSE_REQUIRE(!grp || (!c_bf16 && !slabs && splitk == 1 && !cat && atr_mbs == 0 && seg_len == 0 && a_kstride == BK), CRUSE_E_SHAPE,
                  "gemm_bf16_nt_groups: row-major A, f32 result, no split-K");
    const bool atr = atr_mbs != 0;
    CRUSE_REQUIRE(!atr || (!A_lo && !B_lo && !c_bf16 && !slabs && splitk == 1 && !f16 && !cat && seg_len == 0), CRUSE_E_SHAPE,
                  "gemm_bf16_nt_atr: plain bf16, one pass, no split-K");
    CRUSE_REQUIRE(!f16 || (!A_lo && !slabs && splitk == 1 && !accumulate), CRUSE_E_SHAPE,
                  "gemm_f16_nt: f32 result stored (no A low plane, no split-K, no accumulation)");
    CRUSE_REQUIRE(!c_bf16 || (!accumulate && splitk == 1), CRUSE_E_SHAPE, "gemm_bf16_nt: a bf16 result is stored, not accumulated");
    CRUSE_REQUIRE(seg_len >= 0 && (seg_len == 0 || (seg_stride >= seg_len && seg_off >= 0 && M % seg_len == 0 && a_kstride == BK)),
                  CRUSE_E_SHAPE, "gemm_bf16_nt: bad row segments (len %d stride %lld off %lld, M %d; row-major A only)", seg_len,
                  seg_stride, seg_off, M);
    CRUSE_REQUIRE(M > 0 && N > 0 && K > 0, CRUSE_E_SHAPE, "gemm_bf16_nt: empty shape M=%d N=%d K=%d", M, N, K);
    CRUSE_REQUIRE(K % BK == 0, CRUSE_E_SHAPE, "gemm_bf16_nt: K=%d must be a multiple of %d (pad with zeros)", K, BK);
    CRUSE_REQUIRE(a_kstride >= BK && b_kstride >= BK && ldc >= N, CRUSE_E_SHAPE, "gemm_bf16_nt: strides too small");
    CRUSE_REQUIRE((a_kstride == BK ? lda >= K : lda >= BK) && (b_kstride == BK ? ldb >= K : ldb >= BK), CRUSE_E_SHAPE,
                  "gemm_bf16_nt: lda=%lld / ldb=%lld too small for the operand layout", lda, ldb);
    CRUSE_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && a_kstride % 8 == 0 && b_kstride % 8 == 0 &&
                  ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, CRUSE_E_ALIGN,
                  "gemm_bf16_nt: operands need 16-byte aligned rows and k-tiles (strides multiples of 8)");
    const int nkt = K / BK;
    const bool xcdk = splitk < -1;             // negative: |splitk| k-slices pinned to XCDs
    if (xcdk) splitk = -splitk;
    if (splitk < 1) splitk = 1;
    if (splitk > nkt) splitk = nkt;
    const int kt_chunk = cdiv(nkt, splitk);
    splitk = cdiv(nkt, kt_chunk);
    GbArgs g;
    g.A = (const __bf16*)A; g.B = (const __bf16*)B; g.C = C; g.bias = bias;
    g.a_lo = A_lo ? (const __bf16*)A_lo - (const __bf16*)A : 0;
    g.b_lo = B_lo ? (const __bf16*)B_lo - (const __bf16*)B : 0;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.a_ks = a_kstride; g.b_ks = b_kstride;
    g.accumulate = accumulate; g.splitk = splitk; g.kt_chunk = kt_chunk;
    g.seg_len = seg_len; g.seg_stride = seg_stride; g.seg_off = seg_off;
    g.a_mbs = atr_mbs; g.a_mb_last = atr_mb_last;
    g.c_f16 = c_f16 ? 1 : 0;
    g.tn_per_g = 0; g.a_gstep = g.b_gstep = g.c_gstep = g.bias_gstep = 0;
    g.tm_b1 = g.tm_b2 = 0x7fffffff; g.a_shift1 = g.a_shift2 = 0; g.B1 = g.B2 = nullptr;
    g.m_end0 = g.m_end1 = g.m_end2 = M; g.c_rows1 = g.c_rows2 = 0;
    int M_out = M;                                   // rows of the output (cat: the products' rows without the tile padding between them)
    if (cat) {
        g.tm_b1 = cat->tm_b1; g.tm_b2 = cat->tm_b2; g.a_shift1 = cat->a_shift1; g.a_shift2 = cat->a_shift2; g.B1 = cat->B1; g.B2 = cat->B2;
        g.m_end0 = cat->m_end[0]; g.m_end1 = cat->m_end[1]; g.m_end2 = cat->m_end[2]; g.c_rows1 = cat->c_rows[1]; g.c_rows2 = cat->c_rows[2];
        M_out = cat->M_out;
    }
    g.tiles_m = cdiv(M, BM); g.tiles_n = cdiv(N, BN);
    if (grp) {
        g.tn_per_g = g.tiles_n; g.tiles_n *= grp->G;
        g.a_gstep = grp->a_gstep; g.b_gstep = grp->b_gstep; g.c_gstep = grp->c_gstep; g.bias_gstep = grp->bias_gstep;
    }
    g.xcdk = (xcdk && splitk > 1) ? 1 : 0;
    g.slab = 0;
    const bool use_slabs = slabs != nullptr && splitk > 1;
    float* const c_final = C;
    const long long ldc_final = ldc;
    if (use_slabs) {                           // partial sums [slice][M][N] in the caller's scratch, then one ordered sum into C
        CRUSE_REQUIRE(N % 4 == 0 && ((uintptr_t)slabs % 16) == 0 && seg_len == 0, CRUSE_E_ALIGN, "gemm_bf16_nt: slab form needs N %% 4 == 0");
        CRUSE_REQUIRE((size_t)splitk * M_out * N * sizeof(float) <= slab_bytes, CRUSE_E_SHAPE,
                      "gemm_bf16_nt: %d slabs of %d x %d floats do not fit the %zu-byte scratch", splitk, M_out, N, slab_bytes);
        g.C = slabs; g.ldc = N; g.slab = (long long)M_out * N; g.bias = nullptr;
        CRUSE_REQUIRE(bias == nullptr, CRUSE_E_SHAPE, "gemm_bf16_nt: slab form has no bias");
    }
    if (splitk > 1) { g.nunits = splitk * g.tiles_n; g.inner = g.tiles_m; }
    else { g.nunits = g.tiles_m; g.inner = g.tiles_n; }
    const long long nblk = g.xcdk ? (long long)cdiv(splitk, 8) * 8 * g.tiles_m * g.tiles_n
                                  : (long long)cdiv(g.nunits, 8) * 8 * g.inner;
    CRUSE_REQUIRE(nblk < (1ll << 31), CRUSE_E_SHAPE, "gemm_bf16_nt: grid too large");
    CRUSE_REQUIRE(splitk == 1 || accumulate, CRUSE_E_SHAPE, "gemm_bf16_nt: split-K adds into C (accumulate = 1)");
    const dim3 grid((unsigned)nblk);
    hipStream_t st = (hipStream_t)stream;
    const bool deep = kt_chunk >= 64;          // long k-loops: three stages, one block per CU
    const size_t lds = (size_t)(deep ? 3 : 2) * 2 * TILE_BYTES;
#define CRUSE_GB_LAUNCH(MODE, NST)                                                                               \
    do {                                                                                                         \
        int rc0 = cruse_ensure_dyn_lds(reinterpret_cast<const void*>(gemm_bf16_nt_kernel<MODE, NST>), lds,       \
                                       "gemm_bf16_nt");                                                          \
        if (rc0) return rc0;                                                                                     \
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<MODE, NST>), grid, dim3(256), lds, st, g);                       \
    } while (0)
    if (use_slabs) {
        if (deep) CRUSE_GB_LAUNCH(4, 3); else CRUSE_GB_LAUNCH(4, 2);
        CRUSE_LAUNCH_CHECK("gemm_bf16_nt");
        const long long n4 = (long long)M_out * (N / 4);
        hipLaunchKernelGGL(gemm_slab_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, slabs, splitk, g.slab, M_out, N,
                           c_final, ldc_final);
        CRUSE_LAUNCH_CHECK("gemm_slab_reduce");
        return CRUSE_OK;
    }
    if (atr) {
        const size_t ldsa = (size_t)2 * 2 * TILE_BYTES;
        if (accumulate) {
            int rc0 = cruse_ensure_dyn_lds(reinterpret_cast<const void*>(gemm_bf16_nt_kernel<1, 2, 1, false, true>), ldsa, "gemm_bf16_nt_atr");
            if (rc0) return rc0;
            hipLaunchKernelGGL((gemm_bf16_nt_kernel<1, 2, 1, false, true>), grid, dim3(256), ldsa, st, g);
        } else {
            int rc0 = cruse_ensure_dyn_lds(reinterpret_cast<const void*>(gemm_bf16_nt_kernel<0, 2, 1, false, true>), ldsa, "gemm_bf16_nt_atr");
            if (rc0) return rc0;
            hipLaunchKernelGGL((gemm_bf16_nt_kernel<0, 2, 1, false, true>), grid, dim3(256), ldsa, st, g);
        }
        CRUSE_LAUNCH_CHECK("gemm_bf16_nt_atr");
        return CRUSE_OK;
    }
    if (f16 && c_bf16) {
        const size_t lds16 = (size_t)2 * 2 * TILE_BYTES;
        int rc0 = cruse_ensure_dyn_lds(reinterpret_cast<const void*>(gemm_bf16_nt_kernel<3, 2, 1, true>), lds16, "gemm_nt_out16");
        if (rc0) return rc0;
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<3, 2, 1, true>), grid, dim3(256), lds16, st, g);
        CRUSE_LAUNCH_CHECK("gemm_nt_out16");
        return CRUSE_OK;
    }
    if (f16) {
        const size_t lds16 = (size_t)2 * 2 * TILE_BYTES;
        int rc0 = cruse_ensure_dyn_lds(reinterpret_cast<const void*>(gemm_bf16_nt_kernel<0, 2, 1, true>), lds16, "gemm_f16_nt");
        if (rc0) return rc0;
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<0, 2, 1, true>), grid, dim3(256), lds16, st, g);
        CRUSE_LAUNCH_CHECK("gemm_f16_nt");
        return CRUSE_OK;
    }
    if (deep) {
        if (c_bf16) CRUSE_GB_LAUNCH(3, 3);
        else if (splitk > 1) CRUSE_GB_LAUNCH(2, 3); else if (accumulate) CRUSE_GB_LAUNCH(1, 3); else CRUSE_GB_LAUNCH(0, 3);
    } else {
        if (c_bf16) CRUSE_GB_LAUNCH(3, 2);
        else if (splitk > 1) CRUSE_GB_LAUNCH(2, 2); else if (accumulate) CRUSE_GB_LAUNCH(1, 2); else CRUSE_GB_LAUNCH(0, 2);
    }
#undef CRUSE_GB_LAUNCH
    CRUSE_LAUNCH_CHECK("gemm_bf16_nt");
    return CRUSE_OK;
}

extern "C" int cruse_gemm_bf16_nt(int M, int N, int K, const void* A, long long lda, long long a_kstride,
                                  const void* B, long long ldb, long long b_kstride,
                                  float* C, long long ldc, const float* bias, int accumulate, int splitk,
                                  void* stream) {
    return gemm_bf16_impl(M, N, K, A, nullptr, lda, a_kstride, B, nullptr, ldb, b_kstride, C, ldc, bias, accumulate, splitk,
                          stream);
}

// ... with B = B_hi + B_lo (two f16 planes, same layout): C = A . B_hi^T + A . B_lo^T in one launch (the second k-range on the same accumulators) --
// 11 significant bits on the activations and ~20 on the weights: the forward gate projection of GGRU layer 1 (the rounding of W_ih is the same in
// every frame and does not average out of the gradients the way the per-frame rounding of x does)
extern "C" int cruse_gemm_f16x2_nt(int M, int N, int K, const void* A, long long lda, long long a_kstride,
                                   const void* B_hi, const void* B_lo, long long ldb, long long b_kstride,
                                   float* C, long long ldc, const float* bias, void* stream) {
    CRUSE_REQUIRE(B_lo != nullptr, CRUSE_E_SHAPE, "gemm_f16x2_nt: B_lo is required (one plane: cruse_gemm_f16_nt)");
    return gemm_bf16_impl(M, N, K, A, nullptr, lda, a_kstride, B_hi, B_lo, ldb, b_kstride, C, ldc, bias, 0, 1, stream, 0, 0, 0, false,
                          nullptr, 0, true);
}

// The forward gate projections with a 2-BYTE RESULT: C[M,N] = (A_hi + A_lo) . (B_hi + B_lo)^T + bias stored as IEEE f16 (out_dtype = CRUSE_DT_F16) or
// bf16 (CRUSE_DT_BF16) rows -- gi is the largest tensor of the forward pass (197 MB at the bench shape), written once here and read once by the
// recurrence (cruse_gru_seq_fwd_gi16).  operands_f16 = 0: bf16 operand planes (the forms of cruse_gemm_bf16_nt / cruse_gemm_bf16x3_nt: A_lo, B_lo
// nullable, A_lo needs B_lo); 1: IEEE-f16 planes (cruse_gemm_f16_nt / cruse_gemm_f16x2_nt: no A_lo).  The accumulation is the f32 one of
// those entry points; the result is rounded once at the store.  Replaces the output side of nn.GRU's input projection (model/cruse_net.py:23-31).
extern "C" int cruse_gemm_nt_out16(int M, int N, int K, const void* A_hi, const void* A_lo, long long lda, long long a_kstride,
                                   const void* B_hi, const void* B_lo, long long ldb, long long b_kstride,
                                   void* C, long long ldc, const float* bias, int operands_f16, int out_dtype, void* stream) {
    CRUSE_REQUIRE(out_dtype == CRUSE_DT_F16 || out_dtype == CRUSE_DT_BF16, CRUSE_E_DTYPE, "gemm_nt_out16: out_dtype %d (CRUSE_DT_F16, CRUSE_DT_BF16)", out_dtype);
    CRUSE_REQUIRE((A_lo == nullptr || (B_lo != nullptr && !operands_f16)) && ((uintptr_t)A_lo % 16) == 0 && ((uintptr_t)B_lo % 16) == 0, CRUSE_E_ALIGN,
                  "gemm_nt_out16: low planes (A_lo needs B_lo and bf16 operands; 16-byte aligned)");
    return gemm_bf16_impl(M, N, K, A_hi, A_lo, lda, a_kstride, B_hi, B_lo, ldb, b_kstride, reinterpret_cast<float*>(C), ldc, bias, 0, 1, stream, 0, 0, 0,
                          true, nullptr, 0, operands_f16 != 0, nullptr, 0, 0, nullptr, out_dtype == CRUSE_DT_F16);
}

// C[M,N] = A[M,K] . B[N,K]^T + bias with IEEE-f16 operands (layouts as cruse_gemm_bf16_nt): the forward gate projection in one pass
extern "C" int cruse_gemm_f16_nt(int M, int N, int K, const void* A, long long lda, long long a_kstride,
                                 const void* B, long long ldb, long long b_kstride,
                                 float* C, long long ldc, const float* bias, void* stream) {
    return gemm_bf16_impl(M, N, K, A, nullptr, lda, a_kstride, B, nullptr, ldb, b_kstride, C, ldc, bias, 0, 1, stream, 0, 0, 0, false,
                          nullptr, 0, true);
}


// C[M,N] (+)= A[M,K] . B[N,K]^T with A given as its TIME-MAJOR K-tiled image: element (m, k) at A_T[(m / 64) * a_mb_stride + k * 64 + m % 64],
// n_mb 64-row blocks present (rows M <= m < 64 * n_mb hold finite values).  This is the layout of the gate-gradient tensor dgT
// [ceil(rows / 64)][G][4][Hg][64] the weight-gradient GEMMs consume (A_T = dgT + group * 4 * Hg * 64, a_mb_stride = G * 4 * Hg * 64): the input
// gradient dX = dgi . W_ih is formed from it directly and the row-major copy dgi is never written (round 4; back in round 6 for its bytes).  K % 64 == 0.
extern "C" int cruse_gemm_bf16_nt_atr(int M, int N, int K, const void* A_T, long long a_mb_stride, int n_mb,
                                      const void* B, long long ldb, long long b_kstride,
                                      float* C, long long ldc, int accumulate, void* stream) {
    CRUSE_REQUIRE(a_mb_stride >= (long long)K * 64 && a_mb_stride % 8 == 0 && n_mb >= 1 && (long long)n_mb * 64 >= M, CRUSE_E_SHAPE,
                  "gemm_bf16_nt_atr: a_mb_stride=%lld n_mb=%d for M=%d K=%d", a_mb_stride, n_mb, M, K);
    return gemm_bf16_impl(M, N, K, A_T, nullptr, 64, 64 * 64, B, nullptr, ldb, b_kstride, C, ldc, nullptr, accumulate, 1, stream, 0, 0, 0, false,
                          nullptr, 0, false, nullptr, a_mb_stride, n_mb - 1);
}

// G products of the same shape in ONE launch, side by side along N -- the GRU groups of a layer:
//   C[:, q * c_gstep + (0 .. N)] (+)= A[:, q * a_gstep + (0 .. K)] . B_q^T + bias_q      B_q = B + q * b_gstep, bias_q = bias + q * bias_gstep
// A row-major [M, lda] (planes A_hi / A_lo nullable as cruse_gemm_bf16x3_nt), B in the layouts of cruse_gemm_bf16_nt (B_lo nullable, the same
// group stride).  The forward gate projections gi_q = x_q W_ih,q^T + b_ih,q and the input gradients dx_q = dgi_q W_ih,q of a grouped GGRU layer
// (cruse_net.py:14-55 with rnn_groups > 1) were G launches of [rows x 3 Hg or Hg] each -- 2 to 4 column tiles, 400 to 800 blocks.
extern "C" int cruse_gemm_bf16_nt_groups(int M, int N, int K, int G, const void* A_hi, const void* A_lo, long long lda, long long a_gstep,
                                         const void* B_hi, const void* B_lo, long long ldb, long long b_kstride, long long b_gstep,
                                         float* C, long long ldc, long long c_gstep, const float* bias, long long bias_gstep, int accumulate,
                                         void* stream) {
    CRUSE_REQUIRE(G >= 1 && a_gstep % 8 == 0 && b_gstep % 8 == 0 && c_gstep >= N && (long long)(G - 1) * c_gstep + N <= ldc, CRUSE_E_SHAPE,
                  "gemm_bf16_nt_groups: G=%d a_gstep=%lld b_gstep=%lld c_gstep=%lld", G, a_gstep, b_gstep, c_gstep);
    CRUSE_REQUIRE((A_lo == nullptr || B_lo != nullptr) && ((uintptr_t)A_lo % 16) == 0 && ((uintptr_t)B_lo % 16) == 0, CRUSE_E_ALIGN,
                  "gemm_bf16_nt_groups: low planes (A_lo needs B_lo; 16-byte aligned)");
    const GbGroups grp = {G, a_gstep, b_gstep, c_gstep, bias_gstep};
    return gemm_bf16_impl(M, N, K, A_hi, A_lo, lda, BK, B_hi, B_lo, ldb, b_kstride, C, ldc, bias, accumulate, 1, stream, 0, 0, 0, false, nullptr, 0,
                          false, nullptr, 0, 0, &grp);
}

extern "C" size_t cruse_gemm_bf16_slab_bytes(int M, int N, int splitk) {
    const int z = splitk < 0 ? -splitk : splitk;
    return (size_t)(z < 1 ? 1 : z) * M * N * sizeof(float);
}

extern "C" int cruse_gemm_bf16_nt_slabs(int M, int N, int K, const void* A, long long lda, long long a_kstride,
                                        const void* B, long long ldb, long long b_kstride,
                                        float* C, long long ldc, int splitk, void* scratch, size_t scratch_bytes, void* stream) {
    CRUSE_REQUIRE(scratch != nullptr, CRUSE_E_SHAPE, "gemm_bf16_nt_slabs: scratch is NULL");
    return gemm_bf16_impl(M, N, K, A, nullptr, lda, a_kstride, B, nullptr, ldb, b_kstride, C, ldc, nullptr, 1, splitk, stream, 0, 0, 0,
                          false, reinterpret_cast<float*>(scratch), scratch_bytes);
}

// Up to three products that share N, K, the operand strides and the A tensor, concatenated along M into ONE launch and one output:
//   C[sum M_i, N] += cat_i( A[a_rows[i] .. a_rows[i] + M_i) . B_i^T )          (slab form: split-K without atomics, as above)
// The three weight-gradient products of a GRU layer -- (r, z, n_i)^T x, (r, z)^T h_{t-1}, n_h^T h_{t-1} -- are such a set: their A rows are
// slabs of the one time-major gate-gradient tensor, and dW_ih / dW_hh lie back to back in the flat gradient buffer.  One launch walks all 150
// output tiles of a k-slice on its XCD: the gate-gradient k-tiles are fetched once for the three products, and two launches + two reduce
// passes with their ramps and tails go.  A product whose M_i is not a multiple of 128 still starts on a row tile of its own (the padding rows are
// computed and dropped); the output rows stay contiguous.
extern "C" int cruse_gemm_bf16_nt_slabs_cat(int nprob, const int* Ms, int N, int K, const void* A, const long long* a_rows, long long lda,
                                            long long a_kstride, const void* const* Bs, long long ldb, long long b_kstride,
                                            float* C, long long ldc, int splitk, void* scratch, size_t scratch_bytes, void* stream) {
    CRUSE_REQUIRE(nprob >= 1 && nprob <= 3 && Ms && a_rows && Bs && scratch, CRUSE_E_SHAPE, "gemm_bf16_nt_slabs_cat: 1..3 products");
    GbCat cat = {};
    int tiles = 0, rows = 0, first[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) { cat.m_end[i] = 0; cat.c_rows[i] = 0; }
    for (int i = 0; i < nprob; ++i) {
        CRUSE_REQUIRE(Ms[i] > 0 && a_rows[i] >= 0 && Bs[i] != nullptr && ((uintptr_t)Bs[i] % 16) == 0, CRUSE_E_SHAPE,
                      "gemm_bf16_nt_slabs_cat: product %d (M = %d)", i, Ms[i]);
        first[i] = tiles * BM;                       // a product starts on a row tile; the rows that pad its last tile are computed and dropped
        cat.m_end[i] = first[i] + Ms[i];
        cat.c_rows[i] = rows - first[i];
        tiles += cdiv(Ms[i], BM); rows += Ms[i];
    }
    cat.M_out = rows;
    cat.tm_b1 = nprob > 1 ? first[1] / BM : 0x7fffffff; cat.tm_b2 = nprob > 2 ? first[2] / BM : 0x7fffffff;
    // virtual row m of product i is A row a_rows[i] + (m - first[i]); the kernel adds m * lda itself
    cat.a_shift1 = nprob > 1 ? (a_rows[1] - a_rows[0] - first[1]) * lda : 0;
    cat.a_shift2 = nprob > 2 ? (a_rows[2] - a_rows[0] - first[2]) * lda : 0;
    cat.B1 = nprob > 1 ? (const __bf16*)Bs[1] : nullptr; cat.B2 = nprob > 2 ? (const __bf16*)Bs[2] : nullptr;
    const __bf16* A0 = (const __bf16*)A + a_rows[0] * lda;
    CRUSE_REQUIRE(a_kstride > BK, CRUSE_E_SHAPE, "gemm_bf16_nt_slabs_cat: K-tiled A operand (rows lda apart inside a k-tile)");
    const int M_virtual = first[nprob - 1] + Ms[nprob - 1];
    return gemm_bf16_impl(M_virtual, N, K, A0, nullptr, lda, a_kstride, Bs[0], nullptr, ldb, b_kstride, C, ldc, nullptr, 1, splitk, stream, 0, 0, 0,
                          false, reinterpret_cast<float*>(scratch), scratch_bytes, false, &cat);
}


extern "C" int cruse_gemm_bf16_nt_seg(int M, int N, int K, const void* A_hi, const void* A_lo, long long lda,
                                      const void* B_hi, const void* B_lo, long long ldb, long long b_kstride,
                                      float* C, long long ldc, const float* bias, int accumulate,
                                      int seg_len, long long seg_stride, long long seg_off, void* stream) {
    CRUSE_REQUIRE(seg_len > 0, CRUSE_E_SHAPE, "gemm_bf16_nt_seg: seg_len=%d", seg_len);
    CRUSE_REQUIRE((A_lo == nullptr || B_lo != nullptr) && ((uintptr_t)A_lo % 16) == 0 && ((uintptr_t)B_lo % 16) == 0, CRUSE_E_ALIGN,
                  "gemm_bf16_nt_seg: low planes (A_lo needs B_lo; 16-byte aligned)");
    return gemm_bf16_impl(M, N, K, A_hi, A_lo, lda, BK, B_hi, B_lo, ldb, b_kstride, C, ldc, bias, accumulate, 1, stream, seg_len,
                          seg_stride, seg_off);
}

extern "C" int cruse_gemm_bf16x3_nt(int M, int N, int K, const void* A_hi, const void* A_lo, long long lda,
                                    long long a_kstride, const void* B_hi, const void* B_lo, long long ldb,
                                    long long b_kstride, float* C, long long ldc, const float* bias, int accumulate,
                                    void* stream) {
    CRUSE_REQUIRE(B_lo != nullptr && A_lo != A_hi && B_lo != B_hi, CRUSE_E_SHAPE, "gemm_bf16x3_nt: B_lo missing");
    CRUSE_REQUIRE(((uintptr_t)A_lo % 16) == 0 && ((uintptr_t)B_lo % 16) == 0, CRUSE_E_ALIGN, "gemm_bf16x3_nt: unaligned low planes");
    return gemm_bf16_impl(M, N, K, A_hi, A_lo, lda, a_kstride, B_hi, B_lo, ldb, b_kstride, C, ldc, bias, accumulate, 1, stream);
}

extern "C" int cruse_cast_bf16(const float* x, void* y, long long n, void* stream) {
    return cruse_cast_bf16_split(x, y, nullptr, n, stream);
}

extern "C" int cruse_cast_bf16_split(const float* x, void* y, void* y_lo, long long n, void* stream) {
    CRUSE_REQUIRE(n > 0 && n % 4 == 0, CRUSE_E_SHAPE, "cast_bf16: n=%lld must be a positive multiple of 4", n);
    CRUSE_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 8) == 0 && ((uintptr_t)y_lo % 8) == 0, CRUSE_E_ALIGN,
                  "cast_bf16: unaligned buffers");
    long long nb = (n / 4 + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)y, (__bf16*)y_lo, n / 4);
    CRUSE_LAUNCH_CHECK("cast_bf16");
    return CRUSE_OK;
}

extern "C" int cruse_transpose_bf16(const float* x, long long rows, int cols, long long ld, void* yT, long long ldT,
                                    int shift_T, void* stream) {
    CRUSE_REQUIRE(rows > 0 && cols > 0 && ld >= cols, CRUSE_E_SHAPE, "transpose_bf16: bad shape");
    CRUSE_REQUIRE(ldT % 64 == 0 && ldT >= rows, CRUSE_E_SHAPE,
                  "transpose_bf16: ldT=%lld must be a multiple of 64 and >= rows=%lld", ldT, rows);
    CRUSE_REQUIRE(((uintptr_t)yT % 16) == 0, CRUSE_E_ALIGN, "transpose_bf16: yT must be 16-byte aligned");
    CRUSE_REQUIRE(shift_T >= 0 && (shift_T == 0 || rows % shift_T == 0), CRUSE_E_SHAPE,
                  "transpose_bf16: rows must be whole clips of shift_T frames");
    CRUSE_REQUIRE(((uintptr_t)yT % 4) == 0, CRUSE_E_ALIGN, "transpose_bf16: unaligned output");
    dim3 grid((unsigned)(ldT / 64), (unsigned)cdiv(cols, 64));
    hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, (__bf16*)yT,
                       ldT, shift_T);
    CRUSE_LAUNCH_CHECK("transpose_bf16");
    return CRUSE_OK;
}

extern "C" int cruse_ktile_bf16(const float* x, int rows, int cols, long long ld, void* y, void* y_lo, void* stream) {
    CRUSE_REQUIRE(rows > 0 && cols > 0 && ld >= cols, CRUSE_E_SHAPE, "ktile_bf16: bad shape");
    CRUSE_REQUIRE(((uintptr_t)y % 8) == 0 && ((uintptr_t)y_lo % 8) == 0, CRUSE_E_ALIGN, "ktile_bf16: unaligned output");
    const int kp = (cols + 63) / 64 * 64;
    long long nb = ((long long)rows * (kp / 4) + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(ktile_bf16_kernel<false>, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, (__bf16*)y,
                       (__bf16*)y_lo, kp);
    CRUSE_LAUNCH_CHECK("ktile_bf16");
    return CRUSE_OK;
}

// the same K-tiled layout with IEEE-f16 elements (the B operand of cruse_gemm_f16_nt)
static int ktile_f16_impl(const float* x, int rows, int cols, long long ld, void* y, void* y_lo, void* stream) {
    CRUSE_REQUIRE(rows > 0 && cols > 0 && ld >= cols, CRUSE_E_SHAPE, "ktile_f16: bad shape");
    CRUSE_REQUIRE(((uintptr_t)y % 8) == 0 && ((uintptr_t)y_lo % 8) == 0, CRUSE_E_ALIGN, "ktile_f16: unaligned output");
    const int kp = (cols + 63) / 64 * 64;
    long long nb = ((long long)rows * (kp / 4) + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(ktile_bf16_kernel<true>, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, (__bf16*)y,
                       (__bf16*)y_lo, kp);
    CRUSE_LAUNCH_CHECK("ktile_f16");
    return CRUSE_OK;
}
extern "C" int cruse_ktile_f16(const float* x, int rows, int cols, long long ld, void* y, void* stream) {
    return ktile_f16_impl(x, rows, cols, ld, y, nullptr, stream);
}
extern "C" int cruse_ktile_f16_split(const float* x, int rows, int cols, long long ld, void* y, void* y_lo, void* stream) {
    CRUSE_REQUIRE(y_lo != nullptr, CRUSE_E_SHAPE, "ktile_f16_split: y_lo is required");
    return ktile_f16_impl(x, rows, cols, ld, y, y_lo, stream);
}

