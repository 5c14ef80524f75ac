// Persistent grouped-GRU recurrence, round-4 kernels for the bf16 mode (nn.GRU forward / backward at
// model/cruse_net.py:23-31,44,50): chains of 8 clips, teams of Hg/32 workgroups, register-resident W_hh -- the design of
// gru.hip -- with two changes to the per-step chain:
//
//   1. TAG-FREE HAND-OFF.  The granules of gru.hip carry {epoch u32, 2 x bf16}: half of every published and swept byte is
//      a tag.  Here the tag rides INSIDE the payload: bit 14 of a bf16 (the top exponent bit) is 0 for every |v| < 2, so it
//      is free to carry one bit of epoch -- forward the hidden state is a convex combination of tanh values and the previous
//      state, |h| < 1 always (h0 = 0; a caller-supplied h0 keeps the tagged kernels); backward the partial sums are exchanged
//      scaled by 2^-64 (exact), so anything below 2^65 qualifies and anything above 2^-62 survives.  A slot of parity p is
//      written at epochs p+1, p+3, ...: one alternating bit tells "this epoch" from "two epochs ago", bit(e) = ((e+1)>>1)&1,
//      and bit(first epoch) = 1 differs from the zeroed scratch.  EVERY 2-byte value carries its own tag, so no access wider
//      than a bf16 has to be single-copy atomic (gru.hip relies on 8-byte halves).  Sweep and publish bytes halve: forward
//      10 KB per workgroup and step instead of 20 (3 x 16-byte loads per thread instead of 5), backward 10 KB of publishes
//      in 3 store instructions per wave instead of 20 KB in 5.
//   2. NO K SPLIT in the forward step.  gru_fwd_lean splits K over the four waves and reduces through LDS (six tile
//      writes per wave, a barrier, twelve reads per thread).  Here wave w owns units [8w, 8w+8) of the workgroup's 32 and
//      walks the whole K itself: two 16-row tiles -- rows (r, r, z, z) and (n, n, 0, 0) per unit pair, so that a lane's
//      accumulators hold all three gates of its (clip, unit pair) -- NK = Hg/32 k-steps, 2 NK MFMAs per wave and step (40 at
//      Hg = 640 against 30), 8 NK weight registers per lane (160).  The accumulators ARE the gate pre-activations (bias
//      preloaded); one DPP row rotate hands the second unit of a pair to the idle column lane (the MFMA leaves clips in 8
//      of its 16 columns) and every lane runs one gate evaluation.  One barrier per step (the LDS image of the panel is
//      double-buffered by step parity).
//
// The helper wave (gi ring, saves) is the one of gru_fwd_lean; the loader wave that of gru_bwd_rs.
#include "gru_common.h"

namespace {

using namespace cruse_gru;

// lanes 8..15 of every 16-lane row take `src` of lane - 8, lanes 0..7 keep `old` (v_mov_b32_dpp row_ror:8, bank_mask 0xC)
__device__ __forceinline__ float take_hi8(float old, float src) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(src), 0x128, 0xf, 0xC, false));
}

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
// NK = Hg / 32 k-steps; NS = ceil(Hg / 256) sweep slots per compute thread; WLO: W_hh as hi + lo bf16 planes (Hg <= 320)
// RD: register-direct sweep (see gru_fwd_lean_kernel / gru_bwd_ag_kernel): clip-minor panel, every wave loads ITS B fragments -- here the
// whole K: ceil(NK / 2) 16-byte loads per lane, the MFMA's idle columns 8..15 fetching the odd k-step of each pair -- no LDS image.
template <int NK, int NS, bool WLO, bool TIMED = false, bool RD = false>
__global__ __launch_bounds__(320) void gru_fwd_tf_kernel(GruArgs a) {
    static_assert(!WLO || NK <= 10, "two weight planes fit 256 registers up to Hg = 320");
    // Row stride of the LDS image: a ds_read_b128 is served in groups of 16 lanes over 64 banks, and a group holds the fragments
    // of all 8 clips for two neighbouring 16-byte k-chunks (lane >> 4 = q, q + 1): bank quad = (clip * R + q + 4 ks) mod 16 with R the
    // row stride in 16-byte units -- R = 2 (mod 16) makes the 16 lanes of a group cover all 64 banks (with R = 1 (mod 16), Hg + 8,
    // clip c chunk q + 1 collided with clip c + 1 chunk q: 2-way conflicts on every fragment read, 1435 instead of ~800 cycles)
    constexpr int Hg = NK * 32, LD = 8 * (((NK * 4 - 2 + 15) / 16) * 16 + 2);
    static_assert(LD >= Hg && (LD / 8) % 16 == 2, "LDS row stride");
    unsigned long long tph[5] = {0, 0, 0, 0, 0}, tq0 = 0, tq1 = 0;
    (void)tph; (void)tq0; (void)tq1;
    __shared__ __attribute__((aligned(16))) __bf16 hB[2][8 * LD];                 // B operand (h_{t-1}) by step parity
    // clip strides of 8 (mod 32) floats: a compute wave's 32-lane half touches 8 clips x 4 units per access -- with strides of
    // 96 / 32 floats all 8 clips of a unit fell on ONE bank (8-way conflicts on the 3 gi reads and the 6 saves of every step,
    // ~450 cycles per step that no phase stamp showed)
    constexpr int GS = 104, SS = 40;
    __shared__ __attribute__((aligned(16))) float gi_r[4][8][GS];                 // gi ring: slot = t & 3, [clip][gate*32 + unit]
    __shared__ __attribute__((aligned(16))) float sv_l[2][6][8][SS];              // saves of step t in parity t & 1
    const int H = a.G * Hg;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int chain, part;
    if (!claim_chain(a, a.P, chain, part)) return;
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1); else if (a.prio == 2) __builtin_amdgcn_s_setprio(2); else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
    const int grp = chain % a.G, bgi = a.bg_off + chain / a.G;
    const int b0 = bgi * 8, nb = min(8, a.B - b0);
    const int u0 = part * U;
    const float* W = a.p.w_hh[grp];
    const float* bh = a.p.b_hh[grp];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.xg, 0, a.xg_bytes, 0x00020000);
    constexpr unsigned panel_bytes = (unsigned)(8 * Hg) * 2u;          // [clip][Hg] bf16, the tag inside
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
    const bool save = a.coef != nullptr;

    if (wv == 4) {
        // ---- helper wave: gi rows into the ring four steps ahead, the saves of step t - 1 to HBM during step t ------------
        unsigned gv[3], gdst[3];
#pragma unroll
        for (int i3 = 0; i3 < 3; ++i3) {
            const int idx = lane + 64 * i3, cl = idx / 24, rem = idx % 24, gate = rem >> 3, chk = rem & 7;
            gv[i3] = (unsigned)((((long long)(b0 + (cl < nb ? cl : 0)) * a.TS * a.G + grp) * 3 * Hg + gate * Hg + u0 + 4 * chk) * 4);
            gdst[i3] = (unsigned)(cl * GS + gate * 32 + chk * 4);
        }
        const int lc = lane >> 3, lq = lane & 7;
        const bool rok = lc < nb;
        const unsigned hv = (unsigned)(((long long)(b0 + (rok ? lc : 0)) * a.TS * H + grp * Hg + u0 + 4 * lq) * 4);
        unsigned cv[2], csrc[2];
        bool cok[2];
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            const int idx = min(lane + 64 * i2, 95), cl = idx / 12, rem = idx % 12, gate = rem >> 2, chk = rem & 3;
            cok[i2] = lane + 64 * i2 < 96 && cl < nb;
            cv[i2] = (unsigned)((((long long)(b0 + (cl < nb ? cl : 0)) * a.TS * a.G + grp) * 3 * Hg + gate * Hg + u0 + 8 * chk) * 2);
            csrc[i2] = (unsigned)(((1 + gate) * 8 + cl) * SS + chk * 8);
        }
        struct GiSet { u32x4 v[3]; };
        auto issue = [&](int t, GiSet& o) {
            const unsigned so = (unsigned)min(t, a.T - 1) * grow_bytes;
#pragma unroll
            for (int i3 = 0; i3 < 3; ++i3) o.v[i3] = __builtin_amdgcn_raw_buffer_load_b128(rs_gi, gv[i3], so, 0);
        };
        auto put = [&](int t, const GiSet& o) {
            float* d = &gi_r[t & 3][0][0];
#pragma unroll
            for (int i3 = 0; i3 < 3; ++i3) *reinterpret_cast<u32x4*>(d + gdst[i3]) = o.v[i3];
        };
        auto flush = [&](int t) {                       // saves of step t from parity t & 1
            const float* sl = &sv_l[t & 1][0][0][0];
            const unsigned so = (unsigned)t * frame_bytes, sc = (unsigned)t * crow_bytes;
            if (rok) {
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(sl + (0 * 8 + lc) * SS + 4 * lq), rs_h, hv, so, 0);
                if (save) {
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(sl + (4 * 8 + lc) * SS + 4 * lq), rs_an, hv, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(sl + (5 * 8 + lc) * SS + 4 * lq), rs_z, hv, so, 0);
                }
            }
            if (save) {
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    if (cok[i2]) {
                        const float4 p0 = *reinterpret_cast<const float4*>(sl + csrc[i2]);
                        const float4 p1 = *reinterpret_cast<const float4*>(sl + csrc[i2] + 4);
                        const u32x4 w = {pack2(p0.x, p0.y), pack2(p0.z, p0.w), pack2(p1.x, p1.y), pack2(p1.z, p1.w)};
                        __builtin_amdgcn_raw_buffer_store_b128(w, rs_cf, cv[i2], sc, 0);
                    }
                }
            }
        };
        GiSet s0, s1;
        issue(0, s0); issue(1, s1);
        put(0, s0); put(1, s1);
        issue(2, s0); issue(3, s1);
        if (a.dbg != 9) (void)team_shares_xcd(a.xid + (size_t)chain * 64, a.P, part, a.status, tid);   // mirrors the compute waves' barriers
        __syncthreads();
        // ONE barrier per step.  After the barrier of step t: the set for step t + 2 goes to its ring slot (the compute waves read
        // it after the barrier of step t + 1), the set is re-issued for step t + 4, the saves of step t - 1 are in LDS.
        for (int t = 0; t < a.T; t += 2) {
            __syncthreads();
            put(t + 2, s0); issue(t + 4, s0);
            if (t > 0) flush(t - 1);
            if (t + 1 >= a.T) break;
            __syncthreads();
            put(t + 3, s1); issue(t + 5, s1);
            flush(t);
        }
        __syncthreads();                                // the last step's saves are in LDS
        flush(a.T - 1);
        return;
    }

    // lane = (column c = lane & 15, row group q = lane >> 4): clip c & 7, unit 8 wv + 2q + (c >> 3) of the workgroup's 32
    const int c16 = lane & 15, q = lane >> 4;
    const int clip = c16 & 7, uw = 8 * wv + 2 * q + (c16 >> 3);
    const bool act = clip < nb;
    const int clipc = act ? clip : 0;

    // resident weight fragments.  A-operand row i = lane & 15 of tile A is gate (i >> 1) & 1 (r, z) of unit 8 wv + 2 (i >> 2) + (i & 1);
    // of tile B gate n of the same unit for (i & 3) < 2, zero otherwise.  Element e of k-step ks: column ks*32 + q*8 + e.
    bf16x8 wfA[NK], wfB[NK], wlA[WLO ? NK : 1], wlB[WLO ? NK : 1];
    {
        const int i = c16, un = 8 * wv + 2 * (i >> 2) + (i & 1);
        const int rowA = ((i >> 1) & 1) * Hg + u0 + un, rowB = 2 * Hg + u0 + un;
        const bool okB = (i & 3) < 2;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float wa = W[(long long)rowA * Hg + ks * 32 + q * 8 + e];
                const float wb = okB ? W[(long long)rowB * Hg + ks * 32 + q * 8 + e] : 0.f;
                wfA[ks][e] = (__bf16)wa; wfB[ks][e] = (__bf16)wb;
                if constexpr (WLO) { wlA[ks][e] = (__bf16)(wa - (float)wfA[ks][e]); wlB[ks][e] = (__bf16)(wb - (float)wfB[ks][e]); }
            }
        }
    }
    // accumulator start values = b_hh of the rows this lane's accumulators hold: A (r, r', z, z'), B (n, n', -, -) of units 2q, 2q + 1
    f32x4 biasA, biasB;
    {
        const int un = u0 + 8 * wv + 2 * q;
        biasA = (f32x4){bh[un], bh[un + 1], bh[Hg + un], bh[Hg + un + 1]};
        biasB = (f32x4){bh[2 * Hg + un], bh[2 * Hg + un + 1], 0.f, 0.f};
    }

    // sweep slots: 16-byte load e = tid + 256 j covers clip e / (Hg/8), units 8 (e % (Hg/8)) .. +7 (clamped for short chains and
    // for the slots beyond the panel: the last valid load is then fetched and stored twice)
    constexpr int per = Hg >> 3;
    const int nload = nb * per;
    unsigned sw_v[NS];
    int sw_l[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int e = min(tid + 256 * j, nload - 1);
        const int bl = e / per, v = 8 * (e - bl * per);
        sw_v[j] = (unsigned)e * 16u;
        sw_l[j] = bl * LD + v;
    }
    const unsigned pub_v = RD ? (unsigned)(((((u0 >> 3) + wv) * 8 + clipc) * 8 + 2 * q) * 2) : (unsigned)(clipc * Hg + u0 + 8 * wv + 2 * q) * 2u;
    const bool pub_lane = act && c16 < 8;
    constexpr int NLR = (NK + 1) / 2;
    const unsigned rd_v = (unsigned)((q * 8 + clip) * 16), rd_v2 = rd_v + ((c16 >> 3) ? 512u : 0u);     // + j * 1024: k-steps 2 j, 2 j + 1
    const int fb_off = clip * LD + q * 8;                     // B fragment of k-step ks: + ks * 32 (columns 8..15 re-read clips 0..7)
    float* const sl0 = &sv_l[0][0][clipc][uw];
    const float* const gi0 = &gi_r[0][clipc][uw];

    float hp = 0.f;
    bool nowait = a.dbg >= 1 && a.dbg < 6;
    const bool plain = a.dbg != 9 && team_shares_xcd(a.xid + (size_t)chain * 64, a.P, part, a.status, tid);
    __syncthreads();                                           // ring slots 0 and 1 are filled

    for (int t = 0; t < a.T; ++t) {
        if constexpr (TIMED) tq0 = __builtin_amdgcn_s_memtime();
        // gi of this step (ring slot t & 3, filled two steps ago)
        float gic[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) gic[g] = gi0[(t & 3) * (8 * GS) + g * 32];
        __bf16* const hb = hB[t & 1];
        u32x4 gr[RD ? NLR : 1];
        if (RD && t > 0) {
            const unsigned soff = cbase + (unsigned)((t - 1) & 1) * panel_bytes;
            const bool expect1 = tag_bit((unsigned)t) != 0u;
            unsigned spins = 0;
            for (int i = 0; i < a.poll_delay; ++i) __builtin_amdgcn_s_sleep(1);     // (see poll_delay)
            for (;;) {
#pragma unroll
                for (int j = 0; j < (RD ? NLR : 1); ++j)
                    gr[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (2 * j + 1 < NK || j + 1 < NLR) ? rd_v2 : rd_v, soff + (unsigned)(j * 1024), 16);
                unsigned bad;
                if (expect1) {
                    unsigned n = 0xffffffffu;
#pragma unroll
                    for (int j = 0; j < (RD ? NLR : 1); ++j) n = n & (gr[j].x & gr[j].y) & (gr[j].z & gr[j].w);
                    bad = ~n;
                } else {
                    unsigned o = 0u;
#pragma unroll
                    for (int j = 0; j < (RD ? NLR : 1); ++j) o = o | (gr[j].x | gr[j].y) | (gr[j].z | gr[j].w);
                    bad = o;
                }
                if (__all((bad & TAGM) == 0u || !act || nowait)) break;
                if (++spins >= SPIN_LIMIT) {
                    if (lane == 0) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    nowait = true;
                }
                if constexpr (TIMED) tph[4] += 1;
                __builtin_amdgcn_s_sleep(1);
            }
            if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[0] += tq1 - tq0; tq0 = tq1; }
        }
        if (!RD && t > 0) {
            const unsigned soff = cbase + (unsigned)((t - 1) & 1) * panel_bytes;
            const unsigned flip = tag_bit((unsigned)t) ? 0xffffffffu : 0u;
            u32x4 g[NS];
            unsigned spins = 0;
            for (int i = 0; i < a.poll_delay; ++i) __builtin_amdgcn_s_sleep(1);     // (see poll_delay)
            for (;;) {
#pragma unroll
                for (int j = 0; j < NS; ++j) g[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, sw_v[j], soff, 16);
                unsigned bad = 0u;
#pragma unroll
                for (int j = 0; j < NS; ++j) bad |= (g[j].x ^ flip) | (g[j].y ^ flip) | (g[j].z ^ flip) | (g[j].w ^ flip);
                if (__all((bad & TAGM) == 0u || nowait)) break;
                if (++spins >= SPIN_LIMIT) {
                    if (lane == 0) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    nowait = true;
                }
                if constexpr (TIMED) tph[4] += 1;
                __builtin_amdgcn_s_sleep(1);
            }
            if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[0] += tq1 - tq0; tq0 = tq1; }
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const u32x4 w = {g[j].x & ~TAGM, g[j].y & ~TAGM, g[j].z & ~TAGM, g[j].w & ~TAGM};
                *reinterpret_cast<u32x4*>(hb + sw_l[j]) = w;
            }
        }
        __syncthreads();                                       // panel of step t complete (and the helper's ring / saves hand-over)
        if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[1] += tq1 - tq0; tq0 = tq1; }
        f32x4 aA = biasA, aB = biasB;
        if (t > 0) {
            f32x4 aA1 = (f32x4){0.f, 0.f, 0.f, 0.f}, aB1 = aA1;
            // Fragment reads run PF k-steps ahead of their MFMAs; sched_barrier(0) after every k-step pins that order.  (Left to
            // itself the scheduler keeps two fragment registers and issues each pair of reads right in front of its MFMAs: every
            // second k-step exposed a full LDS round trip, 1370 cycles for the 40 MFMAs of Hg = 640 instead of ~700 -- s_memtime
            // stamps, gru_dbg = 32; sched_group_barrier pipelines were followed for six k-steps and then abandoned.)
            constexpr int PF = NK < 6 ? NK : 6;
            bf16x8 fr[NK];
            if constexpr (RD) {
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    const u32x4 gg = gr[ks >> 1];
                    u32x4 w = {gg.x & ~TAGM, gg.y & ~TAGM, gg.z & ~TAGM, gg.w & ~TAGM};
                    if (ks & 1) { w.x = dpp_ror8(w.x); w.y = dpp_ror8(w.y); w.z = dpp_ror8(w.z); w.w = dpp_ror8(w.w); }     // columns 8..15 -> 0..7
                    fr[ks] = __builtin_bit_cast(bf16x8, w);
                }
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    if ((ks & 1) == 0) {
                        aA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfA[ks], fr[ks], aA, 0, 0, 0);
                        aB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfB[ks], fr[ks], aB, 0, 0, 0);
                        if constexpr (WLO) {
                            aA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlA[ks], fr[ks], aA, 0, 0, 0);
                            aB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlB[ks], fr[ks], aB, 0, 0, 0);
                        }
                    } else {
                        aA1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfA[ks], fr[ks], aA1, 0, 0, 0);
                        aB1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfB[ks], fr[ks], aB1, 0, 0, 0);
                        if constexpr (WLO) {
                            aA1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlA[ks], fr[ks], aA1, 0, 0, 0);
                            aB1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlB[ks], fr[ks], aB1, 0, 0, 0);
                        }
                    }
                }
            } else {
    #pragma unroll
                for (int ks = 0; ks < PF; ++ks) fr[ks] = *reinterpret_cast<const bf16x8*>(hb + fb_off + ks * 32);
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    if ((ks & 1) == 0) {
                        aA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfA[ks], fr[ks], aA, 0, 0, 0);
                        aB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfB[ks], fr[ks], aB, 0, 0, 0);
                        if constexpr (WLO) {
                            aA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlA[ks], fr[ks], aA, 0, 0, 0);
                            aB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlB[ks], fr[ks], aB, 0, 0, 0);
                        }
                    } else {
                        aA1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfA[ks], fr[ks], aA1, 0, 0, 0);
                        aB1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfB[ks], fr[ks], aB1, 0, 0, 0);
                        if constexpr (WLO) {
                            aA1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlA[ks], fr[ks], aA1, 0, 0, 0);
                            aB1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlB[ks], fr[ks], aB1, 0, 0, 0);
                        }
                    }
                    if (ks + PF < NK) fr[ks + PF] = *reinterpret_cast<const bf16x8*>(hb + fb_off + (ks + PF) * 32);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            aA += aA1; aB += aB1;
        }
        if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[2] += tq1 - tq0; tq0 = tq1; }
        // gate pre-activations of this lane's (clip, unit): columns 0..7 unit 2q, columns 8..15 unit 2q + 1 of the same clip
        const float ghr = take_hi8(aA[0], aA[1]);
        const float ghz = take_hi8(aA[2], aA[3]);
        const float ghn = take_hi8(aB[0], aB[1]);
        const float r = lean_sigmoid(gic[0] + ghr);
        const float z = lean_sigmoid(gic[1] + ghz);
        const float n = lean_tanh(gic[2] + r * ghn);
        const float h = (1.f - z) * n + z * hp;
        {
            const float hn = __uint_as_float(dpp_ror8(__float_as_uint(h)));       // unit 2q + 1 of the clip, from lane + 8
            if (pub_lane) {
                const unsigned w = with_tag(pack2(h, hn), tag_bit((unsigned)(t + 1)) ? TAGM : 0u);
                const unsigned soff = cbase + (unsigned)(t & 1) * panel_bytes;
                if (plain) __builtin_amdgcn_raw_buffer_store_b32(w, rs, pub_v, soff, 0);
                else __builtin_amdgcn_raw_buffer_store_b32(w, rs, pub_v, soff, 16);
            }
        }
        if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[3] += tq1 - tq0; tq0 = tq1; }
        const float an = (1.f - z) * (1.f - n * n);
        if (act) {                                             // saves into parity t & 1 (the helper reads them after the next barrier)
            float* sl = sl0 + (t & 1) * (6 * 8 * SS);
            sl[0 * 8 * SS] = h;
            sl[1 * 8 * SS] = an * ghn * r * (1.f - r);
            sl[2 * 8 * SS] = (hp - n) * z * (1.f - z);
            sl[3 * 8 * SS] = an * r;
            sl[4 * 8 * SS] = an;
            sl[5 * 8 * SS] = z;
        }
        hp = h;
    }
    if constexpr (TIMED) {
        if (tid == 0 && chain == 0 && part == 0) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.status) + 8;     // byte 64 of the status header
            for (int i = 0; i < 5; ++i) dst[i] = tph[i];
            dst[5] = (unsigned long long)a.T;
        }
    }
    __syncthreads();                                           // hands the last step's saves to the helper wave
}


typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------
// backward, ALL-GATHER form on the tag-free hand-off (round 4).  dh_{s-1} = dout_{s-1} + z_s . dh_s + (dh_s . c_s) W_hh:
// a workgroup owns 32 OUTPUT units and contracts over the whole K = 3 Hg itself -- the forward lean kernel's step with the
// product panel p_s = (dh_s . c_r, dh_s . c_z, dh_s . c_n) as its B operand and W_hh^T as the resident A operand:
//   * publish: the 3 x 32 products of the own units per clip, bf16 with the epoch bit in bit 14, scaled by 2^-64 (exact; the
//     reduced sum is multiplied by 2^64): 1.5 KB per workgroup and step instead of the reduce-scatter form's 10 KB;
//   * sweep: the whole panel [clip][gate][Hg] (30 KB at Hg = 640, 8 x 16-byte loads per thread) straight into the LDS image --
//     measured on the forward kernel (gru_xsweep: 30 KB swept + 3 publishes per step): sweep 590 -> 890 cycles, against the
//     1520 cycles the reduce-scatter kernel waits for its 20 producers' partial sums;
//   * 3 Hg / 32 k-steps split over the four waves (15 each at Hg = 640), two 16-unit tiles: 30 MFMAs per wave and step as
//     before, the K reduction through LDS is two tiles per wave instead of the forward's six;
//   * the partial sums are never rounded: f32 accumulation over the whole K (the reduce-scatter form exchanged 20 bf16 partials).
// The loader wave (operand ring, dh / gate-gradient rows to HBM) is that of gru_bwd_tf_kernel with the forward helper's two
// barriers per step.
// ---------------------------------------------------------------------------------
// DGI: 0 = dh only; 3 / 4 = the gate-gradient rows too (a.dg_slabs slabs) -- compile-time, so that the loader wave's loop is
// straight-line code and the compiler's in-order vmcnt counts are EXACT (see the loader)
// RD: no LDS image -- a wave sweeps exactly the chunks that ARE its MFMA B fragments (k-steps wv + 4 i of every clip: NKW 16-byte loads per
// lane, one base address + immediate offsets) straight into registers and checks its own tags; the tag bits are cleared beside the MFMAs.
template <int P, bool TIMED = false, int DGI = 0, bool RD = false>
__global__ __launch_bounds__(320) void gru_bwd_ag_kernel(GruArgs a) {
    constexpr int Hg = P * 32, K3 = 3 * Hg, NKS = K3 / 32, NKW = (NKS + 3) / 4, LD = K3 + 8;
    constexpr int NCH = K3 / 8, NS = (8 * NCH + 255) / 256;         // 16-byte chunks per clip row; sweep slots per thread
    constexpr float SC = 5.421010862427522e-20f, ISC = 1.8446744073709552e19f;     // 2^-64, 2^64
    unsigned long long tph[5] = {0, 0, 0, 0, 0}, tq0 = 0, tq1 = 0;
    (void)tph; (void)tq0; (void)tq1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* const pB = reinterpret_cast<__bf16*>(smem_raw);                             // [8 clips][LD]: B operand (p_s)
    float* const red = reinterpret_cast<float*>(pB + 8 * LD);                           // [4 waves][2 tiles][RED_TS]
    constexpr int RS = 36;
    __shared__ __attribute__((aligned(16))) float op_d[4][8][RS], op_z[4][8][RS];       // ring slot = iteration & 3
    __shared__ __attribute__((aligned(16))) __bf16 op_c[4][8][96];
    __shared__ __attribute__((aligned(16))) float op_a[4][8][RS];                       // a_n rows (only when dgi is written)
    __shared__ __attribute__((aligned(16))) float dh_l[2][8][RS];                       // dh of iteration k in parity k & 1
    const int H = a.G * Hg;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int chain, part;
    if (!claim_chain(a, P, chain, part)) return;
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1); else if (a.prio == 2) __builtin_amdgcn_s_setprio(2); else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
    const int grp = chain % a.G, bgi = a.bg_off + chain / a.G;
    const int b0 = bgi * 8, nb = min(8, a.B - b0);
    const int u0 = part * U;
    const float* W = a.p.w_hh[grp];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.xg, 0, a.xg_bytes, 0x00020000);
    constexpr unsigned panel_bytes = (unsigned)(8 * K3) * 2u;          // [clip][gate][Hg] bf16, the tag inside
    const unsigned cbase = (unsigned)chain * 2u * panel_bytes;

    for (int i = tid; i < 8 * LD / 8; i += 320) reinterpret_cast<u32x4*>(pB)[i] = (u32x4){0u, 0u, 0u, 0u};

    const unsigned frame_bytes = (unsigned)H * 4u, crow_bytes = (unsigned)(a.G * K3) * 2u;
    const long long nrow = (long long)(a.B - 1) * a.TS + a.T;           // rows reachable from the (advanced) base pointers
    const unsigned tot_f32 = (unsigned)min(nrow * H * 4, 0xffffffffll);
    const unsigned tot_cf = (unsigned)min(nrow * a.G * K3 * 2, 0xffffffffll);
    const __amdgpu_buffer_rsrc_t rs_dout = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dout), 0, tot_f32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.zs), 0, tot_f32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dh = __builtin_amdgcn_make_buffer_rsrc(a.dh, 0, tot_f32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cf = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.coefs), 0, tot_cf, 0x00020000);

    if (wv == 4) {
        // ---- loader wave.  Iteration j needs dout_{T-1-j}, c_{T-1-j} and z_{T-j}; lane = (clip, 16-byte chunk).
        // BRANCH-FREE: every iteration issues exactly NLD loads and NST stores -- rows beyond the sequence are clamped (never used),
        // lanes without a clip / chunk address beyond the buffer (raw-buffer loads return 0 there, stores are dropped) -- so the
        // compiler counts the in-order vmcnt queue exactly and waits only for the set it is about to use (issued two iterations
        // earlier).  With data-dependent branches around the loads / stores it assumed none of the younger ones had been issued and
        // waited for ALL of them, the stores of the running iteration included: 0.4 us per step (gru_dbg = 7: 1.36 against 1.76).
        const int lc = lane >> 3, lq = lane & 7;                                   // dout / z: 8 clips x 8 chunks of 4 floats
        constexpr unsigned OOB = 0xfffffff0u;
        const unsigned dv = lc < nb ? (unsigned)(((long long)(b0 + lc) * a.TS * H + grp * Hg + u0 + 4 * lq) * 4) : OOB;
        unsigned cv[2], cdst[2];
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {                                           // coef: 8 clips x 3 gates x 4 chunks of 8 bf16
            const int idx = lane + 64 * i2;
            const int cl = min(idx, 95) / 12, rem = min(idx, 95) % 12, gate = rem >> 2, chk = rem & 3;
            cv[i2] = (idx < 96 && cl < nb) ? (unsigned)((((long long)(b0 + cl) * a.TS * a.G + grp) * K3 + gate * Hg + u0 + 8 * chk) * 2) : OOB;
            cdst[i2] = (unsigned)(cl * 96 + gate * 32 + chk * 8);
        }
        const bool c1ok = lane < 32;                                               // (slots 64..95 of the coefficient set)
        const __amdgpu_buffer_rsrc_t rs_an = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ans), 0, a.ans ? tot_f32 : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_dgi = __builtin_amdgcn_make_buffer_rsrc(a.dgi, 0, a.dgi ? (DGI == 4 ? (unsigned)min(nrow * a.G * 4 * Hg * 2, 0xffffffffll) : tot_cf) : 0u, 0x00020000);
        struct OpSet { u32x4 d, z, c0, c1, an; };
        auto issue = [&](int j, OpSet& o) {                                        // j >= 1
            const unsigned st = (unsigned)max(a.T - 1 - j, 0);
            o.d = __builtin_amdgcn_raw_buffer_load_b128(rs_dout, dv, st * frame_bytes, 0);
            o.z = __builtin_amdgcn_raw_buffer_load_b128(rs_z, dv, min(st + 1u, (unsigned)(a.T - 1)) * frame_bytes, 0);
            o.c0 = __builtin_amdgcn_raw_buffer_load_b128(rs_cf, cv[0], st * crow_bytes, 0);
            o.c1 = __builtin_amdgcn_raw_buffer_load_b128(rs_cf, cv[1], st * crow_bytes, 0);
            if constexpr (DGI != 0) o.an = __builtin_amdgcn_raw_buffer_load_b128(rs_an, dv, st * frame_bytes, 0);
        };
        auto put = [&](int j, const OpSet& o) {
            const int slot = j & 3;
            *reinterpret_cast<u32x4*>(&op_d[slot][lc][4 * lq]) = o.d;
            *reinterpret_cast<u32x4*>(&op_z[slot][lc][4 * lq]) = o.z;
            *reinterpret_cast<u32x4*>(&op_c[slot][0][0] + cdst[0]) = o.c0;
            if (c1ok) *reinterpret_cast<u32x4*>(&op_c[slot][0][0] + cdst[1]) = o.c1;
            if constexpr (DGI != 0) *reinterpret_cast<u32x4*>(&op_a[slot][lc][4 * lq]) = o.an;
        };
        constexpr int NSL = DGI == 4 ? 4 : 3;
        const unsigned dgrow_bytes = (unsigned)(a.G * NSL * Hg) * 2u;
        const unsigned gi_v = lc < nb ? (unsigned)((((long long)(b0 + lc) * a.TS * a.G + grp) * NSL * Hg + u0 + 4 * lq) * 2) : OOB - 8u * (unsigned)Hg;
        auto flush = [&](int j) {
            const unsigned st = (unsigned)(a.T - 1 - j);
            const float4 d4 = *reinterpret_cast<const float4*>(&dh_l[j & 1][lc][4 * lq]);
            const u32x4 dw = {__float_as_uint(d4.x), __float_as_uint(d4.y), __float_as_uint(d4.z), __float_as_uint(d4.w)};
            __builtin_amdgcn_raw_buffer_store_b128(dw, rs_dh, dv, st * frame_bytes, 0);
            if constexpr (DGI != 0) {
                const int slot = j & 3;
                typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
                const bf16x4_ cr = *reinterpret_cast<const bf16x4_*>(&op_c[slot][lc][4 * lq]);
                const bf16x4_ cz = *reinterpret_cast<const bf16x4_*>(&op_c[slot][lc][32 + 4 * lq]);
                const float4 a4 = *reinterpret_cast<const float4*>(&op_a[slot][lc][4 * lq]);
                const float d[4] = {d4.x, d4.y, d4.z, d4.w}, an_[4] = {a4.x, a4.y, a4.z, a4.w};
                bf16x4_ o0, o1, o2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o0[e] = (__bf16)(d[e] * (float)cr[e]); o1[e] = (__bf16)(d[e] * (float)cz[e]); o2[e] = (__bf16)(d[e] * an_[e]);
                }
                __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2*>(&o0), rs_dgi, gi_v, st * dgrow_bytes, 0);
                __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2*>(&o1), rs_dgi, gi_v + (unsigned)Hg * 2u, st * dgrow_bytes, 0);
                __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2*>(&o2), rs_dgi, gi_v + (unsigned)Hg * 4u, st * dgrow_bytes, 0);
                if constexpr (DGI == 4) {
                    const bf16x4_ cn = *reinterpret_cast<const bf16x4_*>(&op_c[slot][lc][64 + 4 * lq]);
                    bf16x4_ o3;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o3[e] = (__bf16)(d[e] * (float)cn[e]);
                    __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2*>(&o3), rs_dgi, gi_v + (unsigned)Hg * 6u, st * dgrow_bytes, 0);
                }
            }
        };
        OpSet s0, s1;
        {                                                                          // iteration 0: dh of the last frame may be carried in; no z
            const unsigned st = (unsigned)(a.T - 1);
            const u32x4 zero = {0u, 0u, 0u, 0u};
            s0.d = a.carry ? __builtin_amdgcn_raw_buffer_load_b128(rs_dh, dv, st * frame_bytes, 0)
                           : __builtin_amdgcn_raw_buffer_load_b128(rs_dout, dv, st * frame_bytes, 0);
            s0.z = zero;
            s0.c0 = __builtin_amdgcn_raw_buffer_load_b128(rs_cf, cv[0], st * crow_bytes, 0);
            s0.c1 = __builtin_amdgcn_raw_buffer_load_b128(rs_cf, cv[1], st * crow_bytes, 0);
            s0.an = zero;
            if constexpr (DGI != 0) s0.an = __builtin_amdgcn_raw_buffer_load_b128(rs_an, dv, st * frame_bytes, 0);
        }
        issue(1, s1);
        put(0, s0); put(1, s1);
        issue(2, s0); issue(3, s1);
        if (a.dbg != 9) (void)team_shares_xcd(a.xid + (size_t)chain * 64, P, part, a.status, tid);    // mirrors the compute waves' barriers
        __syncthreads();
        // iteration k > 0 has two barriers (image complete; partial sums complete).  ALL loader work sits between them, right after
        // the sweep has returned: its loads then have the whole MFMA + pointwise + publish time to come back before the next sweep
        // is issued (loads of the loader in flight DURING a sweep delay it -- the CU's vector-memory return path is shared: with
        // the ring work after the second barrier a step took 1.78 us instead of 1.36 without streams), and the only wait is for
        // the set issued two iterations earlier (exact counts: the loop is branch-free).
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

    // resident A fragments: tile j (units u0 + 16 j .. + 15), this wave's k-steps ks = wv + 4 i; k = gate * Hg + unit
    bf16x8 wf[2][NKW];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = u0 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < NKW; ++i) {
            const int ks = wv + 4 * i;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kk = ks * 32 + (lane >> 4) * 8 + e;
                wf[j][i][e] = ks < NKS ? (__bf16)W[(long long)kk * Hg + n] : (__bf16)0.f;
            }
        }
    }

    // sweep slots: 16-byte chunk e = tid + 256 j of the panel = clip e / NCH, elements 8 (e % NCH) .. + 7 of its row (clamped for
    // short chains and beyond the panel: the last valid chunk is then fetched and stored twice)
    const int nload = nb * NCH;
    unsigned sw_v[NS];
    int sw_l[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int e = min(tid + 256 * j, nload - 1);
        const int bl_ = e / NCH, v = 8 * (e - bl_ * NCH);
        sw_v[j] = (unsigned)e * 16u;
        sw_l[j] = bl_ * LD + v;
    }

    // pointwise: thread = (clip bl, unit u)
    const int u = tid & 31, bl = tid >> 5;
    const bool act = bl < nb;
    const int blc = act ? bl : 0;
    const int half = u >> 4, ru = u & 15;
    const int lp = (ru >> 2) * 16 + bl;
    const unsigned pub_v = RD ? (unsigned)(((((u0 + u) >> 3) * 8 + blc) * 8 + (u & 7)) * 2)           // + gate * (Hg / 8) * 128
                              : (unsigned)((blc * 3) * Hg + u0 + u) * 2u;                            // + gate * Hg * 2
    constexpr unsigned pub_gs = RD ? (unsigned)(Hg / 8) * 128u : (unsigned)Hg * 2u;
    const bool pub_lane = act && !(u & 1);
    const int fb_off = (lane & 7) * LD + (lane >> 4) * 8;                 // (columns 8..15 re-read clips 0..7)
    // (NKS % 4 != 0 -- Hg = 160, 320: the last waves' missing k-steps re-read k-step NKS - 1 (valid tags, finite data) against zero weights)
    // RD panel layout: CLIP-MINOR -- [k chunk of 8 elements][clip][8] -- so that the 8 clips of one (k-step, lane group) are one 128-byte
    // line and a wave's load instruction is 512 contiguous bytes (clip-major rows made every 16-lane pass touch 8 lines: 4500-cycle sweeps)
    const unsigned rd_v = (unsigned)(((wv * 4 + (lane >> 4)) * 8 + (lane & 7)) * 16);         // + i * 2048 bytes: k-step wv + 4 i
    // (columns 8..15: the odd k-step of the pair, + 2048; in the last, half-empty pair they re-read the even one)
    const unsigned rd_v2 = rd_v + (((lane >> 3) & 1) ? 2048u : 0u), rd_v2l = rd_v;
    constexpr int NL_ = (NKW + 1) / 2;
    unsigned rd_o[(RD && NKS % 4 != 0) ? NL_ : 1];          // per-load byte offsets where k-steps have to be clamped
    if constexpr (RD && NKS % 4 != 0) {
#pragma unroll
        for (int j = 0; j < NL_; ++j) {
            const int i = min(2 * j + ((lane >> 3) & 1), NKW - 1);
            rd_o[j] = (unsigned)(((min(wv + 4 * i, NKS - 1) * 4 + (lane >> 4)) * 8 + (lane & 7)) * 16);
        }
    }
    const bool rd_ok = (lane & 7) < nb;                                  // (columns 8..15 re-read clips 0..7; masking them off made every load a branch: slower)

    float dh = 0.f, dd = 0.f, zz = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;    // operands of the current step (time s)
    bool nowait = a.dbg >= 1 && a.dbg < 7;
    const bool plain = a.dbg != 9 && team_shares_xcd(a.xid + (size_t)chain * 64, P, part, a.status, tid);
    __syncthreads();                                                      // ring slots 0 and 1 are filled (and the image is zeroed)
    dd = op_d[0][blc][u];
    c0 = (float)op_c[0][blc][u]; c1 = (float)op_c[0][blc][32 + u]; c2 = (float)op_c[0][blc][64 + u];

    for (int k = 0; k < a.T; ++k) {
        const int s = a.T - 1 - k;
        float m = 0.f;
        if constexpr (TIMED) tq0 = __builtin_amdgcn_s_memtime();
        if (k > 0 && RD) {
            // NL = ceil(NKW / 2) loads per lane: columns 0..7 of the MFMA (lanes with (lane & 15) < 8) fetch the fragment of k-step 2 j, the
            // otherwise idle columns 8..15 that of k-step 2 j + 1 of the same clip; a DPP row rotate brings it over when its MFMAs are due.
            // (With every lane loading its own column's fragment -- columns 8..15 duplicates -- the 60 load instructions of a workgroup
            //  kept the CU's address unit busy for ~1000 cycles: sweep 1700 cycles.)
            constexpr int NL = (NKW + 1) / 2;
            const unsigned soff = cbase + (unsigned)((k - 1) & 1) * panel_bytes;
            const bool expect1 = tag_bit((unsigned)k) != 0u;
            u32x4 g[NL];
            unsigned spins = 0;
            for (int i = 0; i < a.poll_delay; ++i) __builtin_amdgcn_s_sleep(1);     // (see poll_delay)
            for (;;) {
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    if constexpr (NKS % 4 != 0) g[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, rd_o[j], soff, 16);
                    else g[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (2 * j + 1 < NKW || j + 1 < NL) ? rd_v2 : rd_v2l, soff + (unsigned)(j * 4096), 16);
                }
                unsigned bad;
                if (expect1) {                                 // every tag bit set <=> the AND of all dwords has both
                    unsigned n = 0xffffffffu;
#pragma unroll
                    for (int j = 0; j < NL; ++j) n = n & (g[j].x & g[j].y) & (g[j].z & g[j].w);
                    bad = ~n;
                } else {
                    unsigned o = 0u;
#pragma unroll
                    for (int j = 0; j < NL; ++j) o = o | (g[j].x | g[j].y) | (g[j].z | g[j].w);
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
                const u32x4 gg = g[i >> 1];
                u32x4 w = {gg.x & ~TAGM, gg.y & ~TAGM, gg.z & ~TAGM, gg.w & ~TAGM};
                if (i & 1) { w.x = dpp_ror8(w.x); w.y = dpp_ror8(w.y); w.z = dpp_ror8(w.z); w.w = dpp_ror8(w.w); }     // columns 8..15 -> 0..7
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
            for (int w = 0; w < 4; ++w) m += red[(w * 2 + half) * RED_TS + red_vec(lp) + (ru & 3)];
            m *= ISC;
        }
        if (k > 0 && !RD) {
            const unsigned soff = cbase + (unsigned)((k - 1) & 1) * panel_bytes;
            const unsigned flip = tag_bit((unsigned)k) ? 0xffffffffu : 0u;
            u32x4 g[NS];
            unsigned spins = 0;
            for (int i = 0; i < a.poll_delay; ++i) __builtin_amdgcn_s_sleep(1);     // (see poll_delay)
            for (;;) {
#pragma unroll
                for (int j = 0; j < NS; ++j) g[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, sw_v[j], soff, 16);
                unsigned bad = 0u;
#pragma unroll
                for (int j = 0; j < NS; ++j) bad |= (g[j].x ^ flip) | (g[j].y ^ flip) | (g[j].z ^ flip) | (g[j].w ^ flip);
                if (__all((bad & TAGM) == 0u || nowait)) break;
                if (++spins >= SPIN_LIMIT) {
                    if (lane == 0) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    nowait = true;
                }
                if constexpr (TIMED) tph[4] += 1;
                __builtin_amdgcn_s_sleep(1);
            }
            if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[0] += tq1 - tq0; tq0 = tq1; }
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const u32x4 w = {g[j].x & ~TAGM, g[j].y & ~TAGM, g[j].z & ~TAGM, g[j].w & ~TAGM};
                *reinterpret_cast<u32x4*>(pB + sw_l[j]) = w;
            }
            __syncthreads();                                   // image complete
            if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[1] += tq1 - tq0; tq0 = tq1; }
            // four accumulator chains (two per tile): with two, every MFMA waited for the one issued just before its predecessor
            // (1445 cycles for the 30 MFMAs against the forward kernel's 1040 with six chains, s_memtime stamps)
            f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
            // fragment reads run PF k-steps ahead of their MFMAs, pinned by sched_barrier (left alone the scheduler keeps two fragment
            // registers and exposes an LDS round trip every second k-step -- as in gru_fwd_tf_kernel)
            constexpr int PF = NKW < 6 ? NKW : 6;
            bf16x8 fr[NKW];
#pragma unroll
            for (int i = 0; i < PF; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(pB + fb_off + min(wv + 4 * i, NKS - 1) * 32);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NKW; ++i) {
                if (NKS % 4 == 0 || wv + 4 * i < NKS) {        // wave-uniform (weights beyond NKS are zero anyway)
                    if (i & 1) {
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][i], fr[i], acc2, 0, 0, 0);
                        acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][i], fr[i], acc3, 0, 0, 0);
                    } else {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][i], fr[i], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][i], fr[i], acc1, 0, 0, 0);
                    }
                }
                if (i + PF < NKW) fr[i + PF] = *reinterpret_cast<const bf16x8*>(pB + fb_off + min(wv + 4 * (i + PF), NKS - 1) * 32);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc0 += acc2; acc1 += acc3;
            *reinterpret_cast<f32x4*>(red + (wv * 2 + 0) * RED_TS + red_vec(lane)) = acc0;
            *reinterpret_cast<f32x4*>(red + (wv * 2 + 1) * RED_TS + red_vec(lane)) = acc1;
            __syncthreads();
            if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[2] += tq1 - tq0; tq0 = tq1; }
#pragma unroll
            for (int w = 0; w < 4; ++w) m += red[(w * 2 + half) * RED_TS + red_vec(lp) + (ru & 3)];
            m *= ISC;
        }
        dh = dd + zz * dh + m;
        if (act) dh_l[k & 1][bl][u] = dh;                      // the loader wave writes it (and the gate gradients) to HBM
        if (s == 0) break;                                     // nothing consumes the products of time 0
        {
            const float ds = dh * SC;
            const unsigned tagm = tag_bit((unsigned)(k + 1)) ? TAGM : 0u;
            const unsigned soff = cbase + (unsigned)(k & 1) * panel_bytes;
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
            dd = op_d[slot][blc][u];
            zz = op_z[slot][blc][u];
            c0 = (float)op_c[slot][blc][u]; c1 = (float)op_c[slot][blc][32 + u]; c2 = (float)op_c[slot][blc][64 + u];
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

bool fwd_tf_eligible(int Bg, int Hg, int prec, bool has_h0, bool gi_bf16) {
    if (cruse_opt("gru_tf", 1) == 0) return false;             // A/B switch (tests, probes): 0 = the tagged kernels of gru.hip
    // Hg = 640: the K-split-free step measured slower than the lean kernel's (1.36 against 1.27 us per step; 40 instead of 30 MFMAs
    // per wave on the serial chain) -- there the lean kernel runs on the tag-free hand-off instead (gru.hip, TF)
    return prec == CRUSE_PREC_BF16 && Bg == 8 && !has_h0 && !gi_bf16 && (Hg == 160 || Hg == 320);
}

int dispatch_fwd_tf(const GruArgs& a, int grid, bool wlo, hipStream_t s) {
    if (a.Hg == 160) return wlo ? launch_one(gru_fwd_tf_kernel<5, 1, true, false, true>, a, grid, 0, s, "gru_seq_fwd", 320)
                                : launch_one(gru_fwd_tf_kernel<5, 1, false, false, true>, a, grid, 0, s, "gru_seq_fwd", 320);
    return wlo ? launch_one(gru_fwd_tf_kernel<10, 2, true, false, true>, a, grid, 0, s, "gru_seq_fwd", 320)
               : launch_one(gru_fwd_tf_kernel<10, 2, false, false, true>, a, grid, 0, s, "gru_seq_fwd", 320);
}

bool bwd_tf_eligible(int Bg, int Hg, int prec) {
    if (cruse_opt("gru_tf", 1) == 0) return false;
    return prec == CRUSE_PREC_BF16 && Bg == 8 && (Hg == 160 || Hg == 320 || Hg == 640);
}

size_t bwd_ag_lds(int Hg) { return (size_t)8 * (3 * Hg + 8) * 2 + (size_t)4 * 2 * RED_TS * 4; }

// the all-gather kernel with the register-direct sweep (the LDS-image form and the tag-free reduce-scatter kernel measured slower: r4)
int dispatch_bwd_tf(const GruArgs& a, int grid, hipStream_t s) {
#define CRUSE_AG_LAUNCH(PV)                                                                                                             \
    do {                                                                                                                                \
        if (a.dbg == 32) return launch_one(gru_bwd_ag_kernel<PV, true, 0, true>, a, grid, bwd_ag_lds(32 * PV), s, "gru_seq_bwd", 320);   \
        if (a.dgi == nullptr) return launch_one(gru_bwd_ag_kernel<PV, false, 0, true>, a, grid, bwd_ag_lds(32 * PV), s, "gru_seq_bwd", 320); \
        if (a.dg_slabs == 4) return launch_one(gru_bwd_ag_kernel<PV, false, 4, true>, a, grid, bwd_ag_lds(32 * PV), s, "gru_seq_bwd", 320); \
        return launch_one(gru_bwd_ag_kernel<PV, false, 3, true>, a, grid, bwd_ag_lds(32 * PV), s, "gru_seq_bwd", 320);                   \
    } while (0)
    if (a.Hg == 640) CRUSE_AG_LAUNCH(20);
    if (a.Hg == 320) CRUSE_AG_LAUNCH(10);
    CRUSE_AG_LAUNCH(5);
#undef CRUSE_AG_LAUNCH
}

}  // namespace cruse_gru
