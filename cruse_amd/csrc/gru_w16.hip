// Persistent GRU recurrence on WIDE chains: 16 clips per chain -- every column of the 16x16x32 MFMA carries a clip (the chains of 8
// of gru.hip / gru_tf.hip leave columns 8..15 empty: PMC MFMA utilisation 0.09 against 0.043 algorithmic) -- so a batch of 64 at
// Hg = 640 is 4 chains x 20 workgroups = 80 CUs per recurrence and a batch of 128 fits the chip in ONE launch (nn.GRU forward / backward
// at model/cruse_net.py:23-31,44,50).  This is the library's plan for batches whose chains of 8 exceed the CUs (make_plan: B > 96 at
// Hg = 640; forward 1.32 us per step for 16 clips against 1.17 for 8, backward 1.61 against 1.34 -- the round-3 wide kernels took 1.77 /
// 3.08).  Round 5 also ran both GGRU layers co-resident on these chains as a time-chunk wavefront: slower than the serial schedule on
// chains of 8 at B = 64 (profiles/r05_wavefront_chain.txt) -- that schedule is gone, the kernels stay.
//
// Both kernels are the round-4 kernels' steps widened (bf16 mode, Hg % 128 == 0, Hg <= 640):
//   * hand-off: tag-free (the epoch bit in bit 14 of every published bf16, gru_tf.hip) and REGISTER-DIRECT -- the panel is laid out
//     clip-minor, [k chunk of 8 elements][clip 16][8 bf16], so the B fragment of (k-step, lane group q, clip c) is the 16 bytes at
//     ((ks * 4 + q) * 16 + c) * 16: lane (c = lane & 15, q = lane >> 4) loads ITS fragment of each of its k-steps, one load instruction
//     of a wave is 1 KB contiguous, and no LDS image of the panel exists;
//   * forward (gru_fwd_w16_kernel): gru_fwd_lean_kernel<.., TF, RD>'s step -- K split over four waves, 6 x NKW MFMAs per wave, K
//     reduction through LDS in the same order: h, coefficients, a_n and z are BIT-IDENTICAL to the chains of 8 -- with two (clip, unit)
//     pairs of gate math per thread (adjacent units: every LDS access of the gate phase is 8 bytes) and a helper wave that streams
//     16 clips of gi rows / saves;
//   * backward (gru_bwd_w16_kernel): gru_bwd_ag_kernel<.., RD>'s all-gather step on EIGHT compute waves (K = 3 Hg split eight ways:
//     16 weight + 8 fragment vectors per lane instead of 30 + 15, which would not fit the 256 registers a five-wave workgroup
//     leaves) plus the loader wave: 9 waves, <= 168 registers, one (clip, unit) of pointwise work per thread;
//   * a sequence may be run as consecutive time chunks (cruse_gru_seq_*_ex sub-sequences): the first step of a continuation takes its
//     state from the h / dh rows the previous launch wrote (kernel boundary), not from a panel.
#include "gru_common.h"

namespace {

using namespace cruse_gru;

constexpr unsigned OOB = 0xfffffff0u;            // voffset beyond every buffer: raw-buffer loads return 0, stores are dropped

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
template <int NKW, bool TIMED = false>
__global__ __launch_bounds__(320) void gru_fwd_w16_kernel(GruArgs a) {
    constexpr int Hg = NKW * 128, NCHK = Hg / 8;
    // clip strides: 100 floats = 36 (mod 64) -- the 16 clips of a 32-lane group land on 16 different bank quads and the two unit
    // pairs (e2) on the halves of a quad: every ds_read_b64 of the gate phase is conflict-free
    constexpr int GS = 100, SS = 36;
    constexpr unsigned panel_bytes = (unsigned)NCHK * 256u;          // [k chunk][clip 16][8 bf16], the tag inside
    unsigned long long tph[5] = {0, 0, 0, 0, 0}, tq0 = 0, tq1 = 0;
    (void)tph; (void)tq0; (void)tq1;
    __shared__ __attribute__((aligned(16))) float gi_r[4][16][GS];                // gi ring: slot = t & 3, [clip][gate*32 + unit]
    __shared__ __attribute__((aligned(16))) float sv_l[2][6][16][SS];             // saves of step t in parity t & 1
    __shared__ __attribute__((aligned(16))) float red[4 * 6 * RED_TS];            // [4 waves][6 tiles][RED_TS]
    const int H = a.G * Hg;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int chain, part;
    if (!claim_chain(a, a.P, chain, part)) return;
    const int grp = chain % a.G, bgi = a.bg_off + chain / a.G;
    const int b0 = bgi * 16, nb = min(16, a.B - b0);
    const int u0 = part * U;
    const float* W = a.p.w_hh[grp];
    const float* bh = a.p.b_hh[grp];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.xg, 0, a.xg_bytes, 0x00020000);
    const unsigned cbase = (unsigned)chain * 2u * panel_bytes;

    const unsigned frame_bytes = (unsigned)H * 4u, grow_bytes = (unsigned)(a.G * 3 * Hg) * 4u, crow_bytes = grow_bytes >> 1;
    const long long nrow = (long long)(a.B - 1) * a.TS + a.T;           // rows reachable from the (advanced) base pointers
    const unsigned tot_h = (unsigned)min(nrow * H * 4, 0xffffffffll);
    const unsigned tot_g = (unsigned)min(nrow * a.G * 3 * Hg * 4, 0xffffffffll);
    const __amdgpu_buffer_rsrc_t rs_gi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gi), 0, tot_g, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(a.h, 0, tot_h, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_an = __builtin_amdgcn_make_buffer_rsrc(a.an, 0, a.an ? tot_h : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(a.z, 0, a.z ? tot_h : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cf = __builtin_amdgcn_make_buffer_rsrc(a.coef, 0, a.coef ? tot_g >> 1 : 0u, 0x00020000);
    const bool has_h0 = a.h0 != nullptr;

    if (wv == 4) {
        // ---- helper wave: gi rows into the ring four steps ahead, the saves of step t - 1 to HBM during step t.  BRANCH-FREE
        // (lanes without a clip address beyond the buffer; the save buffers of an inference run have zero extent): the in-order
        // vmcnt counts are exact, the only wait is for the set issued two iterations earlier (gru_bwd_ag_kernel's loader).
        // gi: 16 clips x 3 gates x 8 chunks of 4 floats = 384 lane-loads per step
        unsigned gv[6], gdst[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int idx = lane + 64 * i, cl = idx / 24, rem = idx % 24, gate = rem >> 3, chk = rem & 7;
            gv[i] = cl < nb ? (unsigned)((((long long)(b0 + cl) * a.TS * a.G + grp) * 3 * Hg + gate * Hg + u0 + 4 * chk) * 4) : OOB;
            gdst[i] = (unsigned)(cl * GS + gate * 32 + chk * 4);
        }
        // h / a_n / z rows: 16 clips x 8 chunks of 4 floats; coefficient rows: 16 clips x 3 gates x 4 chunks of 8 bf16
        unsigned hv[2], hsrc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = lane + 64 * i, lc = idx >> 3, lq = idx & 7;
            hv[i] = lc < nb ? (unsigned)(((long long)(b0 + lc) * a.TS * H + grp * Hg + u0 + 4 * lq) * 4) : OOB;
            hsrc[i] = (unsigned)(lc * SS + 4 * lq);
        }
        unsigned cv[3], csrc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = lane + 64 * i, cl = idx / 12, rem = idx % 12, gate = rem >> 2, chk = rem & 3;
            cv[i] = cl < nb ? (unsigned)((((long long)(b0 + cl) * a.TS * a.G + grp) * 3 * Hg + gate * Hg + u0 + 8 * chk) * 2) : OOB;
            csrc[i] = (unsigned)(((1 + gate) * 16 + cl) * SS + chk * 8);
        }
        struct GiSet { u32x4 v[6]; };
        auto issue = [&](int t, GiSet& o) {
            const unsigned so = (unsigned)min(t, a.T - 1) * grow_bytes;
#pragma unroll
            for (int i = 0; i < 6; ++i) o.v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_gi, gv[i], so, 0);
        };
        auto put = [&](int t, const GiSet& o) {
            float* d = &gi_r[t & 3][0][0];
#pragma unroll
            for (int i = 0; i < 6; ++i) *reinterpret_cast<u32x4*>(d + gdst[i]) = o.v[i];
        };
        auto flush = [&](int t) {                       // saves of step t from parity t & 1
            const float* sl = &sv_l[t & 1][0][0][0];
            const unsigned so = (unsigned)t * frame_bytes, sc = (unsigned)t * crow_bytes;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(sl + 0 * 16 * SS + hsrc[i]), rs_h, hv[i], so, 0);
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(sl + 4 * 16 * SS + hsrc[i]), rs_an, hv[i], so, 0);
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(sl + 5 * 16 * SS + hsrc[i]), rs_z, hv[i], so, 0);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float4 p0 = *reinterpret_cast<const float4*>(sl + csrc[i]);
                const float4 p1 = *reinterpret_cast<const float4*>(sl + csrc[i] + 4);
                const u32x4 w = {pack2(p0.x, p0.y), pack2(p0.z, p0.w), pack2(p1.x, p1.y), pack2(p1.z, p1.w)};
                __builtin_amdgcn_raw_buffer_store_b128(w, rs_cf, cv[i], sc, 0);
            }
        };
        GiSet s0, s1;
        issue(0, s0); issue(1, s1);
        put(0, s0); put(1, s1);
        issue(2, s0); issue(3, s1);
        (void)team_shares_xcd(a.xid + (size_t)chain * 64, a.P, part, a.status, tid);   // mirrors the compute waves' barriers
        __syncthreads();
        // step t: two barriers (t > 0, or a continuation's first step).  The gi set for step t + 2 goes to its ring slot, the set is
        // re-issued for step t + 4; the saves of step t - 1 are in LDS once the first barrier of step t has passed.
        for (int t = 0; t < a.T; t += 2) {
            if (t > 0 || has_h0) __syncthreads();
            put(t + 2, s0); issue(t + 4, s0);
            if (t > 0) flush(t - 1);
            if (t > 0 || has_h0) __syncthreads();
            if (t + 1 >= a.T) break;
            __syncthreads();
            put(t + 3, s1); issue(t + 5, s1);
            flush(t);
            __syncthreads();
        }
        __syncthreads();                                // the last step's saves are in LDS
        flush(a.T - 1);
        return;
    }

    // resident weight fragments: tile j = gate*2 + half; this wave's k-steps ks = wv + 4*i  (gru_fwd_lean_kernel)
    bf16x8 wf[6][NKW];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int row = (j >> 1) * Hg + u0 + (j & 1) * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < NKW; ++i) {
            const int ks = wv + 4 * i;
#pragma unroll
            for (int e = 0; e < 8; ++e) wf[j][i][e] = (__bf16)W[(long long)row * Hg + ks * 32 + (lane >> 4) * 8 + e];
        }
    }

    // gate math: thread = (clip, unit pair ua, ua + 1).  A wave holds 16 clips x (e2 = bit 4) x (two row quads = bit 5): its 64
    // lanes read 64 consecutive 8-byte slots of a reduction tile and publish one whole 256-byte chunk row of the panel.
    const int clip = tid & 15, e2 = (tid >> 4) & 1, rq = (tid >> 5) & 3, half = tid >> 7;
    const int ua = half * 16 + rq * 4 + e2 * 2;
    const bool act = clip < nb;
    const int rvo = red_vec(rq * 16 + clip) + e2 * 2;                  // float offset inside a reduction tile
    float bias[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g) { bias[g][0] = bh[g * Hg + u0 + ua]; bias[g][1] = bh[g * Hg + u0 + ua + 1]; }
    const unsigned pub_v = (unsigned)(((((u0 + ua) >> 3) * 16 + clip) * 8 + (ua & 7)) * 2);
    // sweep: lane (c16, q) loads the fragment of clip c16, k chunk (wv + 4 i) * 4 + q: + i * 4096 bytes
    const int c16 = lane & 15, q = lane >> 4;
    const unsigned rd_v = (unsigned)(((wv * 4 + q) * 16 + c16) * 16);
    const bool rd_ok = c16 < nb;
    float* const sl0 = &sv_l[0][0][clip][ua];
    const float* const gi0 = &gi_r[0][clip][ua];

    float hp[2] = {0.f, 0.f};
    u32x4 gr[NKW];
    if (has_h0) {
        // continuation: the state is the h row the previous launch wrote (f32), rounded to bf16 exactly as a publish would have
        const int cc = min(c16, nb - 1);
#pragma unroll
        for (int i = 0; i < NKW; ++i) {
            const float* src = a.h0 + (long long)(b0 + cc) * a.h0_bs + grp * Hg + (wv + 4 * i) * 32 + q * 8;
            const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
            gr[i] = (u32x4){pack2(x0.x, x0.y), pack2(x0.z, x0.w), pack2(x1.x, x1.y), pack2(x1.z, x1.w)};
        }
        const float* hs = a.h0 + (long long)(b0 + (act ? clip : 0)) * a.h0_bs + grp * Hg + u0 + ua;
        hp[0] = hs[0]; hp[1] = hs[1];
    }
    bool nowait = false;
    const bool plain = team_shares_xcd(a.xid + (size_t)chain * 64, a.P, part, a.status, tid);
    __syncthreads();                                       // ring slots 0 and 1 are filled
    float gic[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g) { const float2 v = *reinterpret_cast<const float2*>(gi0 + g * 32); gic[g][0] = v.x; gic[g][1] = v.y; }

    for (int t = 0; t < a.T; ++t) {
        const unsigned et = (unsigned)t;                   // epoch of this step's INPUT panel (parity (et - 1) & 1)
        if constexpr (TIMED) tq0 = __builtin_amdgcn_s_memtime();
        if (t > 0) {
            const unsigned soff = cbase + ((et - 1u) & 1u) * panel_bytes;
            const bool expect1 = tag_bit(et) != 0u;
            unsigned spins = 0;
            for (int i = 0; i < a.poll_delay; ++i) __builtin_amdgcn_s_sleep(1);
            for (;;) {
#pragma unroll
                for (int i = 0; i < NKW; ++i) gr[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, rd_v, soff + (unsigned)(i * 4096), 16);
                unsigned bad;
                if (expect1) {
                    unsigned n = 0xffffffffu;
#pragma unroll
                    for (int i = 0; i < NKW; ++i) n = n & (gr[i].x & gr[i].y) & (gr[i].z & gr[i].w);
                    bad = ~n;
                } else {
                    unsigned o = 0u;
#pragma unroll
                    for (int i = 0; i < NKW; ++i) o = o | (gr[i].x | gr[i].y) | (gr[i].z | gr[i].w);
                    bad = o;
                }
                if (__all((bad & TAGM) == 0u || !rd_ok || nowait)) break;
                if (++spins >= SPIN_LIMIT) {
                    if (lane == 0) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    nowait = true;
                }
                if constexpr (TIMED) tph[4] += 1;
                __builtin_amdgcn_s_sleep(1);
            }
            if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[0] += tq1 - tq0; tq0 = tq1; }
        }
        float gh[3][2];
#pragma unroll
        for (int g = 0; g < 3; ++g) { gh[g][0] = bias[g][0]; gh[g][1] = bias[g][1]; }
        if (t > 0 || has_h0) {
            __syncthreads();                               // the helper wave's hand-over point: every wave's sweep has returned
            if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[1] += tq1 - tq0; tq0 = tq1; }
            f32x4 acc[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NKW; ++i) {
                const u32x4 w = {gr[i].x & ~TAGM, gr[i].y & ~TAGM, gr[i].z & ~TAGM, gr[i].w & ~TAGM};
                const bf16x8 fb = __builtin_bit_cast(bf16x8, w);
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][i], fb, acc[j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(red + (wv * 6 + j) * RED_TS + red_vec(lane)) = acc[j];
            __syncthreads();
            if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[2] += tq1 - tq0; tq0 = tq1; }
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float2 v = *reinterpret_cast<const float2*>(red + (w * 6 + g * 2 + half) * RED_TS + rvo);
                    gh[g][0] += v.x; gh[g][1] += v.y;
                }
        }
        float hh[2], sv[6][2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const float r = lean_sigmoid(gic[0][x] + gh[0][x]);
            const float z = lean_sigmoid(gic[1][x] + gh[1][x]);
            const float n = lean_tanh(gic[2][x] + r * gh[2][x]);
            const float h = (1.f - z) * n + z * hp[x];
            const float an = (1.f - z) * (1.f - n * n);
            hh[x] = h;
            sv[0][x] = h;
            sv[1][x] = an * gh[2][x] * r * (1.f - r);
            sv[2][x] = (hp[x] - n) * z * (1.f - z);
            sv[3][x] = an * r;
            sv[4][x] = an;
            sv[5][x] = z;
            hp[x] = h;
        }
        if (act) {
            const unsigned w = with_tag(pack2(hh[0], hh[1]), tag_bit(et + 1u) ? TAGM : 0u);
            const unsigned soff = cbase + (et & 1u) * panel_bytes;
            if (plain) __builtin_amdgcn_raw_buffer_store_b32(w, rs, pub_v, soff, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(w, rs, pub_v, soff, 16);
        }
        if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[3] += tq1 - tq0; tq0 = tq1; }
        {
            // saves into parity t & 1 (the helper reads them after the next barrier); gi of step t + 1 from the ring
            float* sl = sl0 + (t & 1) * (6 * 16 * SS);
#pragma unroll
            for (int s6 = 0; s6 < 6; ++s6) *reinterpret_cast<float2*>(sl + s6 * 16 * SS) = make_float2(sv[s6][0], sv[s6][1]);
            const float* gn = gi0 + ((t + 1) & 3) * (16 * GS);
#pragma unroll
            for (int g = 0; g < 3; ++g) { const float2 v = *reinterpret_cast<const float2*>(gn + g * 32); gic[g][0] = v.x; gic[g][1] = v.y; }
        }
    }
    if constexpr (TIMED) {
        if (tid == 0 && chain == 0 && part == 0) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.status) + 8;     // byte 64 of the status header
            for (int i = 0; i < 5; ++i) dst[i] = tph[i];
            dst[5] = (unsigned long long)a.T;
        }
    }
    __syncthreads();                                       // hands the last step's saves to the helper wave
}

// ---------------------------------------------------------------------------------
// backward, all-gather form: dh_{s-1} = dout_{s-1} + z_s . dh_s + (dh_s . c_s) W_hh.  A workgroup owns 32 OUTPUT units and contracts
// over the whole K = 3 Hg: B operand = the product panel p_s = (dh_s . c_r, dh_s . c_z, dh_s . c_n) of all 16 clips (bf16, scaled by
// 2^-64, epoch bit inside: 60 KB per step at Hg = 640), A = W_hh^T resident in registers.  Eight compute waves: wave w owns k-steps
// w + 8 i (NKW = ceil(3 Hg / 256)); its two accumulator tiles (units u0 .. u0 + 31) go to LDS, every thread sums the eight partials
// of its own (clip, unit).  DGI: 0 = dh only; 3 / 4 = the loader wave also writes the gate-gradient rows (a.dg_slabs slabs).
// ---------------------------------------------------------------------------------
template <int P, int DGI, bool TIMED = false>
__global__ __launch_bounds__(576) void gru_bwd_w16_kernel(GruArgs a) {
    constexpr int Hg = P * 32, K3 = 3 * Hg, NKS = K3 / 32, NW = 8, NKW = (NKS + NW - 1) / NW, NCHK = K3 / 8;
    constexpr float SC = 5.421010862427522e-20f, ISC = 1.8446744073709552e19f;     // 2^-64, 2^64
    constexpr unsigned panel_bytes = (unsigned)NCHK * 256u;            // [k chunk][clip 16][8 bf16]
    constexpr int RS = 40;                                             // clip stride of the operand rings (floats)
    unsigned long long tph[5] = {0, 0, 0, 0, 0}, tq0 = 0, tq1 = 0;
    (void)tph; (void)tq0; (void)tq1;
    __shared__ __attribute__((aligned(16))) float red[NW * 2 * RED_TS];
    __shared__ __attribute__((aligned(16))) float op_d[4][16][RS], op_z[4][16][RS];     // ring slot = iteration & 3
    __shared__ __attribute__((aligned(16))) __bf16 op_c[4][16][96];
    __shared__ __attribute__((aligned(16))) float op_a[4][16][RS];                      // a_n rows (only when dgi is written)
    __shared__ __attribute__((aligned(16))) float dh_l[2][16][RS];                      // dh of iteration k in parity k & 1
    const int H = a.G * Hg;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int chain, part;
    if (!claim_chain(a, P, chain, part)) return;
    const int grp = chain % a.G, bgi = a.bg_off + chain / a.G;
    const int b0 = bgi * 16, nb = min(16, a.B - b0);
    const int u0 = part * U;
    const float* W = a.p.w_hh[grp];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.xg, 0, a.xg_bytes, 0x00020000);
    const unsigned cbase = (unsigned)chain * 2u * panel_bytes;

    const unsigned frame_bytes = (unsigned)H * 4u, crow_bytes = (unsigned)(a.G * K3) * 2u;
    const long long nrow = (long long)(a.B - 1) * a.TS + a.T;           // rows reachable from the (advanced) base pointers
    const unsigned tot_f32 = (unsigned)min(nrow * H * 4, 0xffffffffll);
    const unsigned tot_cf = (unsigned)min(nrow * a.G * K3 * 2, 0xffffffffll);
    const __amdgpu_buffer_rsrc_t rs_dout = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dout), 0, tot_f32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.zs), 0, tot_f32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dh = __builtin_amdgcn_make_buffer_rsrc(a.dh, 0, tot_f32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cf = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.coefs), 0, tot_cf, 0x00020000);

    if (wv == NW) {
        // ---- loader wave (gru_bwd_ag_kernel's, 16 clips).  Iteration j needs dout_{T-1-j}, c_{T-1-j} and z_{T-j}.  BRANCH-FREE.
        unsigned dv[2], ddst[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {                                              // dout / z / dh / a_n: 16 clips x 8 chunks of 4 floats
            const int idx = lane + 64 * i, lc = idx >> 3, lq = idx & 7;
            dv[i] = lc < nb ? (unsigned)(((long long)(b0 + lc) * a.TS * H + grp * Hg + u0 + 4 * lq) * 4) : OOB;
            ddst[i] = (unsigned)(lc * RS + 4 * lq);
        }
        unsigned cv[3], cdst[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {                                              // coef: 16 clips x 3 gates x 4 chunks of 8 bf16
            const int idx = lane + 64 * i, cl = idx / 12, rem = idx % 12, gate = rem >> 2, chk = rem & 3;
            cv[i] = cl < nb ? (unsigned)((((long long)(b0 + cl) * a.TS * a.G + grp) * K3 + gate * Hg + u0 + 8 * chk) * 2) : OOB;
            cdst[i] = (unsigned)(cl * 96 + gate * 32 + chk * 8);
        }
        const __amdgpu_buffer_rsrc_t rs_an = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ans), 0, a.ans ? tot_f32 : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_dgi = __builtin_amdgcn_make_buffer_rsrc(a.dgi, 0, a.dgi ? (DGI == 4 ? (unsigned)min(nrow * a.G * 4 * Hg * 2, 0xffffffffll) : tot_cf) : 0u, 0x00020000);
        struct OpSet { u32x4 d[2], z[2], c[3], an[2]; };
        auto issue = [&](int j, OpSet& o) {                                        // j >= 1
            const unsigned st = (unsigned)max(a.T - 1 - j, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                o.d[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_dout, dv[i], st * frame_bytes, 0);
                o.z[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_z, dv[i], min(st + 1u, (unsigned)(a.T - 1)) * frame_bytes, 0);
                if constexpr (DGI != 0) o.an[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_an, dv[i], st * frame_bytes, 0);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) o.c[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_cf, cv[i], st * crow_bytes, 0);
        };
        auto put = [&](int j, const OpSet& o) {
            const int slot = j & 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *reinterpret_cast<u32x4*>(&op_d[slot][0][0] + ddst[i]) = o.d[i];
                *reinterpret_cast<u32x4*>(&op_z[slot][0][0] + ddst[i]) = o.z[i];
                if constexpr (DGI != 0) *reinterpret_cast<u32x4*>(&op_a[slot][0][0] + ddst[i]) = o.an[i];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(&op_c[slot][0][0] + cdst[i]) = o.c[i];
        };
        constexpr int NSL = DGI == 4 ? 4 : 3;
        const unsigned dgrow_bytes = (unsigned)(a.G * NSL * Hg) * 2u;
        unsigned gi_v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = lane + 64 * i, lc = idx >> 3, lq = idx & 7;
            gi_v[i] = lc < nb ? (unsigned)((((long long)(b0 + lc) * a.TS * a.G + grp) * NSL * Hg + u0 + 4 * lq) * 2) : OOB - 8u * (unsigned)Hg;
        }
        auto flush = [&](int j) {
            const unsigned st = (unsigned)(a.T - 1 - j);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = lane + 64 * i, lc = idx >> 3, lq = idx & 7;
                const float4 d4 = *reinterpret_cast<const float4*>(&dh_l[j & 1][0][0] + ddst[i]);
                const u32x4 dw = {__float_as_uint(d4.x), __float_as_uint(d4.y), __float_as_uint(d4.z), __float_as_uint(d4.w)};
                __builtin_amdgcn_raw_buffer_store_b128(dw, rs_dh, dv[i], st * frame_bytes, 0);
                if constexpr (DGI != 0) {
                    const int slot = j & 3;
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
                    const bf16x4_ cr = *reinterpret_cast<const bf16x4_*>(&op_c[slot][lc][4 * lq]);
                    const bf16x4_ cz = *reinterpret_cast<const bf16x4_*>(&op_c[slot][lc][32 + 4 * lq]);
                    const float4 a4 = *reinterpret_cast<const float4*>(&op_a[slot][0][0] + ddst[i]);
                    const float d[4] = {d4.x, d4.y, d4.z, d4.w}, an_[4] = {a4.x, a4.y, a4.z, a4.w};
                    bf16x4_ o0, o1, o2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o0[e] = (__bf16)(d[e] * (float)cr[e]); o1[e] = (__bf16)(d[e] * (float)cz[e]); o2[e] = (__bf16)(d[e] * an_[e]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2*>(&o0), rs_dgi, gi_v[i], st * dgrow_bytes, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2*>(&o1), rs_dgi, gi_v[i] + (unsigned)Hg * 2u, st * dgrow_bytes, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2*>(&o2), rs_dgi, gi_v[i] + (unsigned)Hg * 4u, st * dgrow_bytes, 0);
                    if constexpr (DGI == 4) {
                        const bf16x4_ cn = *reinterpret_cast<const bf16x4_*>(&op_c[slot][lc][64 + 4 * lq]);
                        bf16x4_ o3;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o3[e] = (__bf16)(d[e] * (float)cn[e]);
                        __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2*>(&o3), rs_dgi, gi_v[i] + (unsigned)Hg * 6u, st * dgrow_bytes, 0);
                    }
                }
            }
        };
        OpSet s0, s1;
        {                                                                          // iteration 0: dh of the last frame may be carried in; no z
            const unsigned st = (unsigned)(a.T - 1);
            const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                s0.d[i] = a.carry ? __builtin_amdgcn_raw_buffer_load_b128(rs_dh, dv[i], st * frame_bytes, 0)
                                  : __builtin_amdgcn_raw_buffer_load_b128(rs_dout, dv[i], st * frame_bytes, 0);
                s0.z[i] = zero;
                s0.an[i] = zero;
                if constexpr (DGI != 0) s0.an[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_an, dv[i], st * frame_bytes, 0);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) s0.c[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_cf, cv[i], st * crow_bytes, 0);
        }
        issue(1, s1);
        put(0, s0); put(1, s1);
        issue(2, s0); issue(3, s1);
        (void)team_shares_xcd(a.xid + (size_t)chain * 64, P, part, a.status, tid);    // mirrors the compute waves' barriers
        __syncthreads();
        // iteration k > 0 has two barriers (sweep returned; partial sums complete); ALL loader work sits between them (gru_bwd_ag_kernel)
        put(2, s0); issue(4, s0);
        for (int k = 1; k < a.T; k += 2) {
            __syncthreads(); put(k + 2, s1); issue(k + 4, s1); flush(k - 1); __syncthreads();
            if (k + 1 >= a.T) break;
            __syncthreads(); put(k + 3, s0); issue(k + 5, s0); flush(k); __syncthreads();
        }
        __syncthreads();                                // the last iteration's dh is in LDS
        flush(a.T - 1);
        return;
    }

    // resident A fragments: tile j (units u0 + 16 j .. + 15), this wave's k-steps ks = wv + 8 i; k = gate * Hg + unit
    bf16x8 wf[2][NKW];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = u0 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < NKW; ++i) {
            const int ks = wv + NW * i;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kk = min(ks, NKS - 1) * 32 + (lane >> 4) * 8 + e;
                wf[j][i][e] = ks < NKS ? (__bf16)W[(long long)kk * Hg + n] : (__bf16)0.f;
            }
        }
    }

    // pointwise: thread = (clip, unit) -- lane bits 0..2 the low unit bits, 3..5 the low clip bits; wave bit 0 the high clip bit,
    // wave bits 1..2 the unit octet: a wave's publish is 8 clips x 16 bytes = 128 contiguous bytes per gate
    const int u = ((wv >> 1) & 3) * 8 + (lane & 7), clip = (wv & 1) * 8 + ((lane >> 3) & 7);
    const bool act = clip < nb;
    const int half = u >> 4, ru = u & 15;
    const int rvo = half * RED_TS + red_vec((ru >> 2) * 16 + clip) + (ru & 3);
    const unsigned pub_v = (unsigned)(((((u0 + u) >> 3) * 16 + clip) * 8 + (u & 7)) * 2);           // + gate * (Hg / 8) * 256
    constexpr unsigned pub_gs = (unsigned)(Hg / 8) * 256u;
    const bool pub_lane = act && !(u & 1);
    // sweep: lane (c16, q) loads the fragment of clip c16, k chunk (wv + 8 i) * 4 + q: + i * 8192 bytes; k-steps beyond NKS
    // (the last slot of the upper waves) re-read k-step NKS - 1 against zero weights
    const int c16 = lane & 15, q = lane >> 4;
    const unsigned rd_v = (unsigned)(((wv * 4 + q) * 16 + c16) * 16);
    const unsigned rd_last = (unsigned)(((min(wv + NW * (NKW - 1), NKS - 1) * 4 + q) * 16 + c16) * 16);
    const bool rd_ok = c16 < nb;
    const int opo = clip * RS + u;

    float dh = 0.f, dd = 0.f, zz = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;    // operands of the current step (time s)
    bool nowait = false;
    const bool plain = team_shares_xcd(a.xid + (size_t)chain * 64, P, part, a.status, tid);
    __syncthreads();                                                      // ring slots 0 and 1 are filled
    dd = (&op_d[0][0][0])[opo];
    c0 = (float)op_c[0][clip][u]; c1 = (float)op_c[0][clip][32 + u]; c2 = (float)op_c[0][clip][64 + u];

    for (int k = 0; k < a.T; ++k) {
        const int s = a.T - 1 - k;
        const unsigned ek = (unsigned)k;
        float m = 0.f;
        if constexpr (TIMED) tq0 = __builtin_amdgcn_s_memtime();
        if (k > 0) {
            const unsigned soff = cbase + ((ek - 1u) & 1u) * panel_bytes;
            const bool expect1 = tag_bit(ek) != 0u;
            u32x4 g[NKW];
            unsigned spins = 0;
            for (int i = 0; i < a.poll_delay; ++i) __builtin_amdgcn_s_sleep(1);
            for (;;) {
#pragma unroll
                for (int i = 0; i < NKW - 1; ++i) g[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, rd_v, soff + (unsigned)(i * 8192), 16);
                g[NKW - 1] = __builtin_amdgcn_raw_buffer_load_b128(rs, rd_last, soff, 16);
                unsigned bad;
                if (expect1) {
                    unsigned n = 0xffffffffu;
#pragma unroll
                    for (int i = 0; i < NKW; ++i) n = n & (g[i].x & g[i].y) & (g[i].z & g[i].w);
                    bad = ~n;
                } else {
                    unsigned o = 0u;
#pragma unroll
                    for (int i = 0; i < NKW; ++i) o = o | (g[i].x | g[i].y) | (g[i].z | g[i].w);
                    bad = o;
                }
                if (__all((bad & TAGM) == 0u || !rd_ok || nowait)) break;
                if (++spins >= SPIN_LIMIT) {
                    if (lane == 0) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    nowait = true;
                }
                if constexpr (TIMED) tph[4] += 1;
                __builtin_amdgcn_s_sleep(1);
            }
            if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[0] += tq1 - tq0; tq0 = tq1; }
            __syncthreads();                                   // (the loader wave's hand-over point: every wave's sweep has returned)
            if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[1] += tq1 - tq0; tq0 = tq1; }
            f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
#pragma unroll
            for (int i = 0; i < NKW; ++i) {
                const u32x4 w = {g[i].x & ~TAGM, g[i].y & ~TAGM, g[i].z & ~TAGM, g[i].w & ~TAGM};
                const bf16x8 fb = __builtin_bit_cast(bf16x8, w);
                if (i & 1) {
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][i], fb, acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][i], fb, acc3, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][i], fb, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][i], fb, acc1, 0, 0, 0);
                }
            }
            acc0 += acc2; acc1 += acc3;
            *reinterpret_cast<f32x4*>(red + (wv * 2 + 0) * RED_TS + red_vec(lane)) = acc0;
            *reinterpret_cast<f32x4*>(red + (wv * 2 + 1) * RED_TS + red_vec(lane)) = acc1;
            __syncthreads();
            if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[2] += tq1 - tq0; tq0 = tq1; }
#pragma unroll
            for (int w = 0; w < NW; ++w) m += red[w * 2 * RED_TS + rvo];
            m *= ISC;
        }
        dh = dd + zz * dh + m;
        (&dh_l[k & 1][0][0])[opo] = dh;                        // the loader wave writes it (and the gate gradients) to HBM
        if (s == 0) break;                                     // nothing consumes the products of time 0
        {
            const float ds = dh * SC;
            const unsigned tagm = tag_bit(ek + 1u) ? TAGM : 0u;
            const unsigned soff = cbase + (ek & 1u) * panel_bytes;
            const float p0 = ds * c0, p1 = ds * c1, p2 = ds * c2;
            const float q0 = __uint_as_float(dpp_xor1(__float_as_uint(p0)));
            const float q1 = __uint_as_float(dpp_xor1(__float_as_uint(p1)));
            const float q2 = __uint_as_float(dpp_xor1(__float_as_uint(p2)));
            if (pub_lane) {
                const unsigned w0 = with_tag(pack2(p0, q0), tagm), w1 = with_tag(pack2(p1, q1), tagm), w2 = with_tag(pack2(p2, q2), tagm);
                if (plain) {
                    __builtin_amdgcn_raw_buffer_store_b32(w0, rs, pub_v, soff, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(w1, rs, pub_v + pub_gs, soff, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(w2, rs, pub_v + 2u * pub_gs, soff, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(w0, rs, pub_v, soff, 16);
                    __builtin_amdgcn_raw_buffer_store_b32(w1, rs, pub_v + pub_gs, soff, 16);
                    __builtin_amdgcn_raw_buffer_store_b32(w2, rs, pub_v + 2u * pub_gs, soff, 16);
                }
            }
        }
        if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[3] += tq1 - tq0; tq0 = tq1; }
        {                                                      // operands of step k+1 from the loader wave's ring (slot = iteration & 3)
            const int slot = (k + 1) & 3;
            dd = (&op_d[slot][0][0])[opo];
            zz = (&op_z[slot][0][0])[opo];
            c0 = (float)op_c[slot][clip][u]; c1 = (float)op_c[slot][clip][32 + u]; c2 = (float)op_c[slot][clip][64 + u];
        }
    }
    if constexpr (TIMED) {
        if (tid == 0 && chain == 0 && part == 0) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.status) + 16;    // byte 128 of the status header
            for (int i = 0; i < 5; ++i) dst[i] = tph[i];
            dst[5] = (unsigned long long)a.T;
        }
    }
    __syncthreads();                                           // hands the last iteration's dh to the loader wave
}

}  // namespace

namespace cruse_gru {

// wide chains: bf16 mode, whole 128-unit k groups; the forward kernel takes f32 gi rows and either no initial state or one with
// |h0| < 1 (the continuation of a sequence that started from zero: the tag-free format needs |h| < 2)
bool w16_eligible(int Hg, int prec) {
    return prec == CRUSE_PREC_BF16 && Hg % 128 == 0 && Hg >= 128 && Hg <= 640;
}
size_t w16_panel_bytes_per_parity(int Hg, bool fwd) { return (size_t)(fwd ? Hg : 3 * Hg) / 8 * 256; }

int dispatch_fwd_w16(const GruArgs& a, int grid, hipStream_t s) {
    if (a.dbg == 32 && a.Hg == 640) return launch_one(gru_fwd_w16_kernel<5, true>, a, grid, 0, s, "gru_seq_fwd", 320);
    switch (a.Hg / 128) {
        case 1: return launch_one(gru_fwd_w16_kernel<1>, a, grid, 0, s, "gru_seq_fwd", 320);
        case 2: return launch_one(gru_fwd_w16_kernel<2>, a, grid, 0, s, "gru_seq_fwd", 320);
        case 3: return launch_one(gru_fwd_w16_kernel<3>, a, grid, 0, s, "gru_seq_fwd", 320);
        case 4: return launch_one(gru_fwd_w16_kernel<4>, a, grid, 0, s, "gru_seq_fwd", 320);
        default: return launch_one(gru_fwd_w16_kernel<5>, a, grid, 0, s, "gru_seq_fwd", 320);
    }
}

template <int P>
static int launch_bwd_w16(const GruArgs& a, int grid, hipStream_t s) {
    if (a.dbg == 32) return launch_one(gru_bwd_w16_kernel<P, 0, true>, a, grid, 0, s, "gru_seq_bwd", 576);
    if (a.dgi == nullptr) return launch_one(gru_bwd_w16_kernel<P, 0>, a, grid, 0, s, "gru_seq_bwd", 576);
    if (a.dg_slabs == 4) return launch_one(gru_bwd_w16_kernel<P, 4>, a, grid, 0, s, "gru_seq_bwd", 576);
    return launch_one(gru_bwd_w16_kernel<P, 3>, a, grid, 0, s, "gru_seq_bwd", 576);
}

int dispatch_bwd_w16(const GruArgs& a, int grid, hipStream_t s) {
    switch (a.Hg / 128) {
        case 1: return launch_bwd_w16<4>(a, grid, s);
        case 2: return launch_bwd_w16<8>(a, grid, s);
        case 3: return launch_bwd_w16<12>(a, grid, s);
        case 4: return launch_bwd_w16<16>(a, grid, s);
        default: return launch_bwd_w16<20>(a, grid, s);
    }
}

}  // namespace cruse_gru
