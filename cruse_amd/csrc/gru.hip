// Persistent grouped-GRU recurrence for gfx950 (nn.GRU forward / backward at
// model/cruse_net.py:23-31,44,50; gate order r,z,n; n = tanh(gi_n + r*(W_hn h + b_hn));
// h' = (1-z)*n + z*h; h0 = 0).
//
// The T = 401 steps are strictly sequential, so step latency -- not FLOPs or bytes --
// bounds this kernel.  Design:
//   * a CHAIN = (batch group of Bg <= 16 clips) x (GRU group); chains are independent.
//   * a chain is served by a TEAM of P = Hg/32 workgroups; workgroup p owns hidden units
//     [32p, 32p+32) and keeps its slice of W_hh (96 x Hg forward, its transpose 32 x 3Hg
//     backward) resident in REGISTERS as MFMA A-operand fragments for the whole sequence.
//   * hand-off between the workgroups of a team uses DATA-TAGGED GRANULES (the guide's R2
//     form): each exchanged value travels as an 8-byte {epoch tag, f32 value} pair written
//     with a write-through (sc1) 16-byte buffer store (two granules); consumers sweep the
//     team's panel with 16-byte sc1 buffer loads until every tag equals the step's epoch.
//     There is no separate flag, no release fence and no store drain on the critical path;
//     8-byte halves are single-copy atomic.  The panel is double-buffered by step parity: a
//     workgroup can only write epoch e+2 after every team mate published e+1, which implies
//     all of them finished reading e.  Nothing depends on dispatch order or workgroup->XCD
//     placement; block ids are arranged so that, with the observed id%8 placement, a chain
//     sits on one XCD (its panel then stays in that XCD's L2 / the Infinity Cache).
//   * forward step: sweep h_{t-1} [Bg,Hg] into LDS -> 16x16x32 MFMA tiles (K split over the
//     4 wavefronts, reduced through LDS) -> gate math for the 32 own units -> publish h_t.
//   * backward: with c = d(gh)/d(dh) coefficients saved by the forward pass
//     (dgh_t = dh_t * c_t elementwise), only dh_t [Bg,Hg] is exchanged per step -- the same
//     volume as forward -- and dh_{t-1} = dout_{t-1} + z_t*dh_t + (dh_t*c_t) W_hh is the same
//     MFMA shape.  dgi / dgh for the weight-gradient GEMMs are formed afterwards by an
//     elementwise kernel from dh (cruse_gru_gate_grads).
//   * all workgroups of a launch must be co-resident: grid <= number of CUs
//     (one 256-thread workgroup per CU); larger batches are split into launches.
#include "common.h"
#include "gru_common.h"

namespace {

using namespace cruse_gru;


template <int PREC>
__device__ __forceinline__ void store_coef(void* base, long long off, float2 v) {
    if constexpr (PREC == CRUSE_PREC_BF16) reinterpret_cast<unsigned*>(base)[off >> 1] = pack2(v.x, v.y);   // off is even
    else *reinterpret_cast<float2*>(reinterpret_cast<float*>(base) + off) = v;
}

template <int PREC, int NKW>
__global__ __launch_bounds__(256) void gru_fwd_kernel(GruArgs a) {
    typedef typename Panel<PREC>::elem elem;
    constexpr int NPL = Panel<PREC>::NPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int Hg = a.Hg, KS = Hg >> 5, LD = Hg + Panel<PREC>::PAD, H = a.G * Hg;
    const int PLANE = 16 * LD;
    elem* hB = reinterpret_cast<elem*>(smem_raw);                 // [NPL][16][LD]  B operand (h_{t-1})
    float* red = reinterpret_cast<float*>(hB + NPL * PLANE);      // [4 waves][6 tiles][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int chain, part;
    if (!claim_chain(a, a.P, chain, part)) return;
    const int grp = chain % a.G, bgi = a.bg_off + chain / a.G;
    const int b0 = bgi * a.Bg, nb = min(a.Bg, a.B - b0);
    const int u0 = part * U;
    const float* W = a.p.w_hh[grp];
    const float* bh = a.p.b_hh[grp];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.xg, 0, a.xg_bytes, 0x00020000);
    const unsigned panel_bytes = (unsigned)(a.Bg * Hg) * (16u / Gran<PREC>::VPL);
    const unsigned cbase = (unsigned)chain * 2u * panel_bytes;
    const int nload = nb * Hg / Gran<PREC>::VPL;
    SweepIdx<PREC> si0;
    make_idx<PREC>(si0, nload, Hg, LD, 0, 0, tid);

    for (int i = tid; i < NPL * PLANE; i += 256) hB[i] = (elem)0.f;
    const bool has_h0 = a.h0 != nullptr;
    if (has_h0) {                                                 // the panel of step 0 is the initial state
        __syncthreads();
        for (int e = tid; e < nb * (Hg >> 1); e += 256) {
            const int bl_ = e / (Hg >> 1), v = 2 * (e - bl_ * (Hg >> 1));
            const float2 hv = *reinterpret_cast<const float2*>(a.h0 + (long long)(b0 + bl_) * a.h0_bs + grp * Hg + v);
            panel_put2<PREC>(hB, PLANE, bl_ * LD + v, hv.x, hv.y);
        }
    }

    // resident weight fragments: tile j = gate*2 + half; this wave's k-steps ks = wv + 4*i
    Frag<PREC> wf[6][NKW];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int row = (j >> 1) * Hg + u0 + (j & 1) * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < NKW; ++i) {
            const int ks = wv + 4 * i;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[e] = ks < KS ? W[(long long)row * Hg + ks * 32 + (lane >> 4) * 8 + e] : 0.f;
            wf[j][i].set(v);
        }
    }

    // items of this thread: unit u of local clips bl = (tid >> 5) + 8*q  (NIT = Bg/8 items): the gate math
    // (3 transcendentals per item) is spread over all 256 threads
    const int u = tid & 31;
    const int half = u >> 4, ru = u & 15;
    const int NIT = a.Bg >> 3;
    float bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bias[g] = bh[g * Hg + u0 + u];
    float hp[2] = {0.f, 0.f};
    const long long gi_row = (long long)a.G * 3 * Hg;             // gi floats per frame
    float gic[2][3], sv[2][6];                                    // gi of the CURRENT step; deferred saves
    long long gp[2];                                              // element index of (clip, unit) in the first frame
    const bool gib = a.gi_bf16 != 0;
    auto ldgi = [&](long long e) -> float { return gib ? (float)reinterpret_cast<const __bf16*>(a.gi)[e] : a.gi[e]; };
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int bl = (tid >> 5) + 8 * q;
        gp[q] = ((long long)(b0 + bl) * a.TS * a.G + grp) * 3 * Hg + u0 + u;
#pragma unroll
        for (int g = 0; g < 3; ++g) gic[q][g] = (q < NIT && bl < nb) ? ldgi(gp[q] + g * Hg) : 0.f;
#pragma unroll
        for (int e = 0; e < 6; ++e) sv[q][e] = 0.f;
        if (has_h0 && q < NIT && bl < nb) hp[q] = a.h0[(long long)(b0 + bl) * a.h0_bs + grp * Hg + u0 + u];
    }
    bool aborted = a.dbg >= 1 && a.dbg < 8;
    const bool plain = a.dbg != 9 && team_shares_xcd(a.xid + (size_t)chain * 64, a.P, part, a.status, tid);
    __syncthreads();

    auto save_step = [&](int q, int t) {
        const int bl = (tid >> 5) + 8 * q;
        const long long o = ((long long)(b0 + bl) * a.TS + t) * H + grp * Hg + u0 + u;
        a.h[o] = sv[q][0];
        if (a.coef) {
            const long long o3 = (((long long)(b0 + bl) * a.TS + t) * a.G + grp) * 3 * Hg + u0 + u;
            if constexpr (PREC == CRUSE_PREC_BF16) {
                __bf16* cp = reinterpret_cast<__bf16*>(a.coef);
                cp[o3] = (__bf16)sv[q][1]; cp[o3 + Hg] = (__bf16)sv[q][2]; cp[o3 + 2 * Hg] = (__bf16)sv[q][3];
            } else {
                float* cp = reinterpret_cast<float*>(a.coef);
                cp[o3] = sv[q][1]; cp[o3 + Hg] = sv[q][2]; cp[o3 + 2 * Hg] = sv[q][3];
            }
            a.an[o] = sv[q][4];
            a.z[o] = sv[q][5];
        }
    };

    for (int t = 0; t < a.T; ++t) {
        if (t > 0)
            aborted |= sweep_panel<PREC, false>(hB, PLANE, LD, rs, cbase + (unsigned)((t - 1) & 1) * panel_bytes, nload, Hg,
                                                (unsigned)t, si0, nullptr, nullptr, 0, a.status, tid, aborted);
        // (1) saves of step t-1, (2) gi rows of step t+1: both are old by the time of the next sweep
        float gin_[2][3];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int bl = (tid >> 5) + 8 * q;
            const bool act = q < NIT && bl < nb;
            if (act && t > 0) save_step(q, t - 1);
#pragma unroll
            for (int g = 0; g < 3; ++g)
                gin_[q][g] = (act && t + 1 < a.T) ? ldgi(gp[q] + (long long)(t + 1) * gi_row + g * Hg) : 0.f;
        }
        if (t > 0 || has_h0) {
            f32x4 acc[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NKW; ++i) {
                const int ks = wv + 4 * i;
                if (ks < KS && (a.dbg < 2 || a.dbg >= 8)) {
                    const Frag<PREC> fb = panel_get<PREC>(hB, PLANE, (lane & 15) * LD + ks * 32 + (lane >> 4) * 8);
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[j] = mma(wf[j][i], fb, acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 6; ++j)
                *reinterpret_cast<f32x4*>(red + (wv * 6 + j) * RED_TS + red_vec(lane)) = acc[j];
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (q >= NIT) continue;                      // uniform: Bg == 8 has one item per thread
            const int bl = (tid >> 5) + 8 * q;
            const bool act = bl < nb;
            float gh[3] = {bias[0], bias[1], bias[2]};
            if (t > 0 || has_h0) {
                const int lp = (ru >> 2) * 16 + bl;
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int w = 0; w < 4; ++w) gh[g] += red[(w * 6 + g * 2 + half) * RED_TS + red_vec(lp) + (ru & 3)];
            }
            const float r = fast_sigmoid(gic[q][0] + gh[0]);
            const float z = fast_sigmoid(gic[q][1] + gh[1]);
            const float n = fast_tanh(gic[q][2] + r * gh[2]);
            const float h = (1.f - z) * n + z * hp[q];
            {
                // hand-off: f32 granule per value, or one bf16x2 granule per unit pair (even lane publishes)
                const unsigned pbase = cbase + (unsigned)(t & 1) * panel_bytes;
                const unsigned vidx = (unsigned)(bl * Hg + u0 + u);
                if constexpr (PREC == CRUSE_PREC_BF16) {
                    const float hn = __shfl_xor(h, 1, 64);
                    if (act && !(u & 1)) publish_pair<PREC>(rs, pbase, vidx >> 1, (unsigned)(t + 1), h, hn, plain);
                } else {
                    if (act) {
                        const u32x2 w = {(unsigned)(t + 1), __float_as_uint(h)};
                        if (plain) __builtin_amdgcn_raw_buffer_store_b64(w, rs, pbase + vidx * 8u, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b64(w, rs, pbase + vidx * 8u, 0, 16);
                    }
                }
            }
            // dgh = dh * (c_r, c_z, c_n); dgi_n = dh * a_n   (see header)
            const float an = (1.f - z) * (1.f - n * n);
            sv[q][0] = h;
            sv[q][1] = an * gh[2] * r * (1.f - r);
            sv[q][2] = (hp[q] - n) * z * (1.f - z);
            sv[q][3] = an * r;
            sv[q][4] = an;
            sv[q][5] = z;
            hp[q] = h;
#pragma unroll
            for (int g = 0; g < 3; ++g) gic[q][g] = gin_[q][g];
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int bl = (tid >> 5) + 8 * q;
        if (q < NIT && bl < nb) save_step(q, a.T - 1);
    }
}

// ---------------------------------------------------------------------------------
// forward, LEAN form of the kernel above for CRUSE_PREC_BF16, Bg = 8, Hg % 128 == 0, Hg <= 640 (the bench shape).
// Same algorithm, hand-off format and queue discipline; the step loop is rewritten for one wave per SIMD, where
// every instruction's issue slot is exposed (the generic loop spent ~1100 instructions per step, mostly 64-bit
// address arithmetic, exec-mask branches around each granule and the three barriers + LDS atomic of
// __syncthreads_and per sweep attempt):
//   * every access is a buffer instruction: per-thread constant voffset + scalar per-step soffset;
//   * a wave waits for ITS granules with a wave-level vote and reloads all of them per attempt (no per-granule
//     masks); block-wide agreement comes from the panel barrier that is needed anyway;
//   * the k-loop is fully unrolled (NKW = Hg/128 k-steps per wave, all valid).
// ---------------------------------------------------------------------------------
// NKW = ceil(Hg/32 / 4) k-steps per wave, NS = ceil(Hg/128) sweep slots per thread; FULL: Hg % 128 == 0, every k-step
// of every wave exists (the guards below then vanish at compile time -- as run-time tests they cost the bench shape
// 0.9 us per step: the compiler no longer overlaps the fragment reads of one k-step with the MFMAs of the previous)
// WLO: W_hh is carried as TWO bf16 planes (hi + lo, ~2^-17 relative) and every k-step issues a second MFMA with the low
// plane on the same accumulators.  The rounding of the recurrent WEIGHTS is the systematic part of the bf16 recurrence
// error (the same perturbation at every one of the T steps; the rounding of h is fresh noise per step): see DESIGN.md
// section 2 for the measured effect.  Costs 6*NKW more MFMAs per wave and step and 24*NKW more registers per lane.
// HELPER WAVE (HW; every variant but the Hg > 384 two-plane one, whose weights leave no room for a fifth wave): the
// compute waves' gi loads and their six saves per step (h, three coefficient rows, a_n, z) sat in the same in-order vmcnt
// queue as the next step's granule sweep (CRUSE_GRU_DBG=6 drops both: 675 vs 797 us per launch, tools/gru_hog_probe.py).
// A fifth wavefront streams the gi rows into a 4-slot LDS ring four steps ahead and writes the saves -- which the compute
// threads leave in LDS -- to HBM as 16-byte stores one step behind.
// (A WAVE-LOCAL panel -- a producer-major hand-off panel, every wave sweeping exactly the 1 KB blocks of the producers whose
// k-steps it multiplies and reading its fragments back without the "panel complete" barrier -- was measured at 1.66 us per step
// against 1.28, r03: the workgroup then reaches its next sweep ~250 cycles earlier, BEFORE its team mates' granules are in L2,
// and a missed poll costs a whole L2 round trip (~800 cycles): 1.0 re-polls per step instead of 0.1.  With the barrier the
// team sits where the first poll just hits: a step is the workgroup's own ~1900 cycles plus one ~700-cycle load round trip,
// the publish latency hidden under the load's way out.)
// TIMED (profiling, library option gru_dbg = 32 at Hg = 640): s_memtime stamps at the four phase boundaries of a step, summed by
// workgroup (chain 0, part 0) into the status header (tools/gru_probe.py prints them).
// GIB: the gi rows are bf16 (a compile-time variant: the f32 instances stay instruction for instruction what they were -- this
// kernel sits at the 256-register edge, and a run-time switch in the helper wave moved spills into the compute waves' step loop:
// 2.2 us per step instead of 1.39).
// TF: the TAG-FREE hand-off of gru_tf.hip (the epoch bit inside every published bf16, half the sweep and publish bytes: 3 instead of
// 5 sweep loads per thread at Hg = 640) under this kernel's K-split step -- at Hg = 640 the K-split-free step of gru_fwd_tf_kernel
// measured slower (1.36 against 1.27 us per step), the hand-off alone pays.  Needs h0 == NULL (|h| < 1).
// RD (with TF): REGISTER-DIRECT sweep -- no LDS image of the panel.  A wave sweeps exactly the 16-byte chunks that ARE its MFMA B fragments
// (its k-steps wv + 4 i of every clip: NKW loads per lane) and checks its own tags; the panel is laid out CLIP-MINOR ([k chunk of 8][clip][8])
// so that the 8 clips of one (k-step, lane group) are one 128-byte line and a load instruction is 512 contiguous bytes.  The panel barrier
// stays (the helper wave's hand-over point), the image write, the wait for it and the fragment reads go.
// GIH (with GIB, round 6): the 2-byte gi rows are IEEE f16, not bf16 (written by cruse_gemm_nt_out16; 11 significant bits: the enhanced spectrum
// at T = 401 moves 4.10e-4 -> 4.25e-4 / 2.81e-4 -> 2.81e-4, where bf16 rows leave the G6 fixture's bar) -- the helper wave widens with v_cvt_f32_f16.
template <int NKW, int NS, bool FULL, bool WLO, bool TIMED = false, bool GIB = false, bool TF = false, bool RD = false, bool GIH = false>
__global__ __launch_bounds__((WLO && NKW > 3) ? 256 : 320) void gru_fwd_lean_kernel(GruArgs a) {
    static_assert(!GIH || GIB, "f16 gi rows use the 2-byte row geometry of the bf16 form");
    static_assert(!TF || FULL, "tag-free sweeps are built for Hg % 128 == 0");
    static_assert(!RD || (TF && !WLO), "register-direct sweep: tag-free hand-off");
    constexpr int NSW = TF ? (NKW * 128 + 255) / 256 : NS;      // sweep slots per thread
    constexpr bool HW = !(WLO && NKW > 3);
    unsigned long long tph[5] = {0, 0, 0, 0, 0}, tq0 = 0, tq1 = 0;
    (void)tph; (void)tq0; (void)tq1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ __attribute__((aligned(16))) float gi_r[HW ? 4 : 1][8][96];        // gi ring: slot = t & 3, [clip][gate*32 + unit]
    __shared__ __attribute__((aligned(16))) float sv_l[HW ? 2 : 1][6][8][32];     // saves of step t in parity t & 1
    const int Hg = a.Hg, H = a.G * Hg, LD = Hg + 8, KS = Hg >> 5;
    __bf16* hB = reinterpret_cast<__bf16*>(smem_raw);                    // [16][LD]  B operand (h_{t-1})
    float* red = reinterpret_cast<float*>(hB + 16 * LD);                  // [4 waves][6 tiles][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int chain, part;
    if (!claim_chain(a, a.P, chain, part)) return;
    const int grp = chain % a.G, bgi = a.bg_off + chain / a.G;
    const int b0 = bgi * 8, nb = min(8, a.B - b0);
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1); else if (a.prio == 2) __builtin_amdgcn_s_setprio(2); else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
    const int u0 = part * U;
    const float* W = a.p.w_hh[grp];
    const float* bh = a.p.b_hh[grp];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.xg, 0, a.xg_bytes, 0x00020000);
    const unsigned panel_bytes = (unsigned)(8 * Hg) * (TF ? 2u : 4u);  // bf16-pair granules: 4 B per value (TF: 2, the tag inside)
    const unsigned cbase = (unsigned)chain * 2u * panel_bytes;

    for (int i = tid; i < 16 * LD; i += (HW ? 320 : 256)) hB[i] = (__bf16)0.f;

    const unsigned frame_bytes = (unsigned)H * 4u, grow_bytes = (unsigned)(a.G * 3 * Hg) * 4u, crow_bytes = grow_bytes >> 1;
    const long long nrow = (long long)(a.B - 1) * a.TS + a.T;           // rows reachable from the (advanced) base pointers
    const unsigned tot_h = (unsigned)min(nrow * H * 4, 0xffffffffll);
    const unsigned tot_g = (unsigned)min(nrow * a.G * 3 * Hg * 4, 0xffffffffll);
    constexpr bool gib = GIB;                                           // bf16 gi rows: the coefficient rows' geometry
    const __amdgpu_buffer_rsrc_t rs_gi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gi), 0, gib ? tot_g >> 1 : tot_g, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(a.h, 0, tot_h, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_an = __builtin_amdgcn_make_buffer_rsrc(a.an, 0, a.an ? tot_h : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(a.z, 0, a.z ? tot_h : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cf = __builtin_amdgcn_make_buffer_rsrc(a.coef, 0, a.coef ? tot_g >> 1 : 0u, 0x00020000);
    const bool save = a.coef != nullptr;

    if (HW && wv == 4) {
        // ---- helper wave ---------------------------------------------------------------------------------------------
        // gi: 8 clips x 3 gates x 8 chunks of 4 floats = 192 lane-loads per step (3 instructions)
        unsigned gv[3], gdst[3];
#pragma unroll
        for (int i3 = 0; i3 < 3; ++i3) {
            const int idx = lane + 64 * i3, cl = idx / 24, rem = idx % 24, gate = rem >> 3, chk = rem & 7;
            gv[i3] = (unsigned)((((long long)(b0 + (cl < nb ? cl : 0)) * a.TS * a.G + grp) * 3 * Hg + gate * Hg + u0 + 4 * chk) * 4);
            gdst[i3] = (unsigned)(cl * 96 + gate * 32 + chk * 4);
        }
        // saves: h / a_n / z rows: lane = (clip, chunk of 4 floats); coefficient rows: 8 clips x 3 gates x 4 chunks of 8 bf16
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
            csrc[i2] = (unsigned)(((1 + gate) * 8 + cl) * 32 + chk * 8);
        }
        // bf16 gi: 8 clips x 3 gates x 4 chunks of 8 bf16 = 96 lane-loads per step (the coefficient rows' lane map, cv[] below),
        // widened to the same f32 ring
        unsigned gdb[2] = {0u, 0u};
        if constexpr (gib) {
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                const int idx = min(lane + 64 * i2, 95), cl = idx / 12, rem = idx % 12, gate = rem >> 2, chk = rem & 3;
                gdb[i2] = (unsigned)(cl * 96 + gate * 32 + chk * 8);
            }
        }
        struct GiSet { u32x4 v[3]; };
        auto issue = [&](int t, GiSet& o) {
            if constexpr (gib) {
                const unsigned so = (unsigned)min(t, a.T - 1) * crow_bytes;
                o.v[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_gi, cv[0], so, 0);
                if (lane < 32) o.v[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_gi, cv[1], so, 0);
                return;
            }
            const unsigned so = (unsigned)min(t, a.T - 1) * grow_bytes;
#pragma unroll
            for (int i3 = 0; i3 < 3; ++i3) o.v[i3] = __builtin_amdgcn_raw_buffer_load_b128(rs_gi, gv[i3], so, 0);
        };
        auto put = [&](int t, const GiSet& o) {
            float* d = &gi_r[t & 3][0][0];
            if constexpr (gib) {
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    if (i2 == 1 && lane >= 32) break;
                    const u32x4 w = o.v[i2];
                    if constexpr (GIH) {
                        auto lo16 = [](unsigned x) -> float { return (float)__builtin_bit_cast(_Float16, (unsigned short)(x & 0xffffu)); };
                        auto hi16 = [](unsigned x) -> float { return (float)__builtin_bit_cast(_Float16, (unsigned short)(x >> 16)); };
                        *reinterpret_cast<float4*>(d + gdb[i2]) = make_float4(lo16(w.x), hi16(w.x), lo16(w.y), hi16(w.y));
                        *reinterpret_cast<float4*>(d + gdb[i2] + 4) = make_float4(lo16(w.z), hi16(w.z), lo16(w.w), hi16(w.w));
                        continue;
                    }
                    const u32x4 lo = {w.x << 16, w.x & 0xffff0000u, w.y << 16, w.y & 0xffff0000u};
                    const u32x4 hi = {w.z << 16, w.z & 0xffff0000u, w.w << 16, w.w & 0xffff0000u};
                    *reinterpret_cast<u32x4*>(d + gdb[i2]) = lo;
                    *reinterpret_cast<u32x4*>(d + gdb[i2] + 4) = hi;
                }
                return;
            }
#pragma unroll
            for (int i3 = 0; i3 < 3; ++i3) *reinterpret_cast<u32x4*>(d + gdst[i3]) = o.v[i3];
        };
        auto flush = [&](int t) {                       // saves of step t from parity t & 1
            const float* sl = &sv_l[t & 1][0][0][0];
            const unsigned so = (unsigned)t * frame_bytes, sc = (unsigned)t * crow_bytes;
            if (rok) {
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(sl + (0 * 8 + lc) * 32 + 4 * lq), rs_h, hv, so, 0);
                if (save) {
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(sl + (4 * 8 + lc) * 32 + 4 * lq), rs_an, hv, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(sl + (5 * 8 + lc) * 32 + 4 * lq), rs_z, hv, so, 0);
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
        if (a.h0 != nullptr) __syncthreads();           // mirrors the compute waves' barrier before the h0 panel fill
        if (a.dbg != 9) (void)team_shares_xcd(a.xid + (size_t)chain * 64, a.P, part, a.status, tid);   // mirrors the compute waves' barriers
        __syncthreads();
        // step t: two barriers (t > 0).  The gi set for step t + 2 goes to its ring slot, the set is re-issued for step
        // t + 4; the saves of step t - 1 are in LDS once the first barrier of step t has passed.
        const bool h0_step = a.h0 != nullptr;              // step 0 then runs its MFMA phase too (two barriers like any other)
        // (RD: the panel barrier stays although nothing is shared through it any more -- it is the helper wave's hand-over point.  With
        //  per-wave "sweep done" flags in LDS instead, the helper started only after the LAST wave's sweep and became the long pole of the
        //  second barrier: 1.25 against 1.18 us per step.)
        for (int t = 0; t < a.T; t += 2) {
            if (t > 0 || h0_step) __syncthreads();
            put(t + 2, s0); issue(t + 4, s0);
            if (t > 0) flush(t - 1);
            if (t > 0 || h0_step) __syncthreads();
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

    // resident weight fragments: tile j = gate*2 + half; this wave's k-steps ks = wv + 4*i
    bf16x8 wf[6][NKW], wl[WLO ? 6 : 1][WLO ? NKW : 1];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int row = (j >> 1) * Hg + u0 + (j & 1) * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < NKW; ++i) {
            const int ks = wv + 4 * i;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float w = (FULL || ks < KS) ? W[(long long)row * Hg + ks * 32 + (lane >> 4) * 8 + e] : 0.f;
                wf[j][i][e] = (__bf16)w;
                if constexpr (WLO) wl[j][i][e] = (__bf16)(w - (float)wf[j][i][e]);
            }
        }
    }

    // sweep slots: load e = tid + 256*j covers clip e / (Hg/4), units 4*(e % (Hg/4)) ..+3 (clamped for short chains:
    // the last valid granule is then fetched and stored twice)
    // (TF: a 16-byte load carries 8 values)
    const int per = TF ? Hg >> 3 : Hg >> 2, nload = nb * per;
    unsigned sw_v[NSW];
    int sw_l[NSW];
#pragma unroll
    for (int j = 0; j < NSW; ++j) {
        const int e = min(tid + 256 * j, nload - 1);
        const int bl = e / per, v = (TF ? 8 : 4) * (e - bl * per);
        sw_v[j] = (unsigned)e * 16u;
        sw_l[j] = bl * LD + v;
    }

    // gate math: thread = (clip bl, unit u)
    const int u = tid & 31, bl = tid >> 5;
    const bool act = bl < nb;
    const int blc = act ? bl : 0;
    const int half = u >> 4, ru = u & 15;
    const int lp = (ru >> 2) * 16 + bl;
    float bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bias[g] = bh[g * Hg + u0 + u];
    const unsigned own_v = (unsigned)(((long long)(b0 + blc) * a.TS * H + grp * Hg + u0 + u) * 4);                    // + t*frame_bytes
    const unsigned g_v = (unsigned)((((long long)(b0 + blc) * a.TS * a.G + grp) * 3 * Hg + u0 + u) * 4);               // + t*grow_bytes
    const unsigned hg4 = (unsigned)Hg * 4u;
    const unsigned pub_v = RD ? (unsigned)(((((u0 + u) >> 3) * 8 + bl) * 8 + (u & 7)) * 2) : (unsigned)((bl * Hg + u0 + u) >> 1) * (TF ? 4u : 8u);
    const unsigned rd_v = (unsigned)(((wv * 4 + (lane >> 4)) * 8 + (lane & 7)) * 16);         // RD: + i * 2048 bytes = k-step wv + 4 i
    const unsigned rd_v2 = rd_v + (((lane >> 3) & 1) ? 2048u : 0u);                           // (columns 8..15: the odd k-step of a pair)
    const bool rd_ok = (lane & 7) < nb;
    const bool pub_lane = act && !(u & 1);

    float hp = 0.f, gic[3], sv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    unsigned xsink = 0u;
    const bool has_h0 = a.h0 != nullptr;
    if (!TF && has_h0) {                                   // the panel of step 0 is the initial state (bf16, like any h_{t-1})
        __syncthreads();                                   // (the zero fill above)
#pragma unroll
        for (int j = 0; j < NS && j < NSW; ++j) {
            const int e = min(tid + 256 * j, nload - 1);
            const int bl_ = e / per, v = 4 * (e - bl_ * per);
            const float4 hv = *reinterpret_cast<const float4*>(a.h0 + (long long)(b0 + bl_) * a.h0_bs + grp * Hg + v);
            const u32x2 w = {pack2(hv.x, hv.y), pack2(hv.z, hv.w)};
            *reinterpret_cast<u32x2*>(hB + sw_l[j]) = w;
        }
        if (act) hp = a.h0[(long long)(b0 + bl) * a.h0_bs + grp * Hg + u0 + u];
    }
    auto ldgi = [&](int g, unsigned so) -> float {        // one gi value of this thread's (clip, unit): f32, or bf16 widened
        if constexpr (GIH) return (float)__builtin_bit_cast(_Float16, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs_gi, (g_v + g * hg4) >> 1, so >> 1, 0));
        if constexpr (gib) return __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs_gi, (g_v + g * hg4) >> 1, so >> 1, 0) << 16);
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_gi, g_v + g * hg4, so, 0));
    };
    if constexpr (!HW) {
#pragma unroll
        for (int g = 0; g < 3; ++g) gic[g] = ldgi(g, 0u);
    }
    bool nowait = a.dbg >= 1 && a.dbg < 6;
    const bool plain = a.dbg != 9 && team_shares_xcd(a.xid + (size_t)chain * 64, a.P, part, a.status, tid);
    __syncthreads();
    if constexpr (HW) {                                    // ring slots 0 and 1 are filled
#pragma unroll
        for (int g = 0; g < 3; ++g) gic[g] = gi_r[0][blc][g * 32 + u];
    }

    for (int t = 0; t < a.T; ++t) {
        if constexpr (TIMED) tq0 = __builtin_amdgcn_s_memtime();
        constexpr int NLR = RD ? (NKW + 1) / 2 : 1;       // RD: two k-steps per load -- columns 8..15 of the MFMA fetch the odd one (gru_bwd_ag_kernel)
        u32x4 gr[NLR];
        if (RD && t > 0) {
            const unsigned soff = cbase + (unsigned)((t - 1) & 1) * panel_bytes;
            const bool expect1 = tag_bit((unsigned)t) != 0u;
            unsigned spins = 0;
            for (int i = 0; i < a.poll_delay; ++i) __builtin_amdgcn_s_sleep(1);
            for (;;) {
#pragma unroll
                for (int j = 0; j < NLR; ++j)
                    gr[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (2 * j + 1 < NKW || j + 1 < NLR) ? rd_v2 : rd_v, soff + (unsigned)(j * 4096), 16);
                unsigned bad;
                if (expect1) {
                    unsigned n = 0xffffffffu;
#pragma unroll
                    for (int i = 0; i < NLR; ++i) n = n & (gr[i].x & gr[i].y) & (gr[i].z & gr[i].w);
                    bad = ~n;
                } else {
                    unsigned o = 0u;
#pragma unroll
                    for (int i = 0; i < NLR; ++i) o = o | (gr[i].x | gr[i].y) | (gr[i].z | gr[i].w);
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
        if (!RD && t > 0) {
            const unsigned soff = cbase + (unsigned)((t - 1) & 1) * panel_bytes;
            u32x4 g[NSW];
            unsigned spins = 0;
            const unsigned flip = tag_bit((unsigned)t) ? 0xffffffffu : 0u;
            for (;;) {
#pragma unroll
                for (int j = 0; j < NSW; ++j) g[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, sw_v[j], soff, 16);
                for (int rep = 0; rep < a.xsweep; ++rep) {      // profiling (gru_xsweep): the sweep volume of an all-gather backward
#pragma unroll
                    for (int j = 0; j < NSW; ++j) { const u32x4 gx = __builtin_amdgcn_raw_buffer_load_b128(rs, sw_v[j], soff, 16); xsink ^= gx.x ^ gx.w; }
                }
                bool ok = true;
                if constexpr (TF) {
                    unsigned bad = 0u;
#pragma unroll
                    for (int j = 0; j < NSW; ++j) bad |= (g[j].x ^ flip) | (g[j].y ^ flip) | (g[j].z ^ flip) | (g[j].w ^ flip);
                    ok = (bad & TAGM) == 0u;
                } else {
#pragma unroll
                    for (int j = 0; j < NSW; ++j) ok = ok & (g[j].x == (unsigned)t) & (g[j].z == (unsigned)t);
                }
                if (__all(ok || nowait)) break;
                if (++spins >= SPIN_LIMIT) {
                    if (lane == 0) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    nowait = true;
                }
                if constexpr (TIMED) tph[4] += 1;
                __builtin_amdgcn_s_sleep(1);
            }
            if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[0] += tq1 - tq0; tq0 = tq1; }
#pragma unroll
            for (int j = 0; j < NSW; ++j) {
                if constexpr (TF) {
                    const u32x4 w = {g[j].x & ~TAGM, g[j].y & ~TAGM, g[j].z & ~TAGM, g[j].w & ~TAGM};
                    *reinterpret_cast<u32x4*>(hB + sw_l[j]) = w;
                } else {
                    const u32x2 w = {g[j].y, g[j].w};               // already bf16 pairs: the LDS image as is
                    *reinterpret_cast<u32x2*>(hB + sw_l[j]) = w;
                }
            }
            // saves of step t-1: issued after the sweep has returned, old by the time of the next one
            // (helper-wave variants: the helper writes them from LDS)
            if (!HW && act && a.dbg != 6) {                // dbg 6 (profiling): no saves either
                const unsigned so = (unsigned)(t - 1) * frame_bytes;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[0]), rs_h, own_v, so, 0);
                if (save) {
                    const unsigned sc = (unsigned)(t - 1) * crow_bytes;
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pack2(sv[1], 0.f) & 0xffffu), rs_cf, g_v >> 1, sc, 0);
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pack2(sv[2], 0.f) & 0xffffu), rs_cf, (g_v >> 1) + (hg4 >> 1), sc, 0);
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pack2(sv[3], 0.f) & 0xffffu), rs_cf, (g_v >> 1) + hg4, sc, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[4]), rs_an, own_v, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[5]), rs_z, own_v, so, 0);
                }
            }
        }
        // gi rows of step t+1 (clamped at the end: the extra row is never used)
        float gin_[3] = {0.1f, -0.2f, 0.3f};
        if (!HW && a.dbg != 7 && a.dbg != 6) {             // dbg 6 / 7 (profiling): no gi stream
            const unsigned so = (unsigned)min(t + 1, a.T - 1) * grow_bytes;
#pragma unroll
            for (int g = 0; g < 3; ++g) gin_[g] = ldgi(g, so);
        }
        float gh[3] = {bias[0], bias[1], bias[2]};
        if (t > 0 || has_h0) {
            __syncthreads();                               // panel complete (RD: the helper wave's hand-over point)
            if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[1] += tq1 - tq0; tq0 = tq1; }
            f32x4 acc[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NKW; ++i) {
                if (FULL || wv + 4 * i < KS) {              // wave-uniform
                    bf16x8 fb;
                    if constexpr (RD) {
                        const u32x4 gg = gr[RD ? (i >> 1) : 0];
                        u32x4 w = {gg.x & ~TAGM, gg.y & ~TAGM, gg.z & ~TAGM, gg.w & ~TAGM};
                        if (i & 1) { w.x = dpp_ror8(w.x); w.y = dpp_ror8(w.y); w.z = dpp_ror8(w.z); w.w = dpp_ror8(w.w); }     // columns 8..15 -> 0..7
                        fb = __builtin_bit_cast(bf16x8, w);
                    } else {
                        fb = *reinterpret_cast<const bf16x8*>(hB + (lane & 15) * LD + (wv + 4 * i) * 32 + (lane >> 4) * 8);
                    }
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][i], fb, acc[j], 0, 0, 0);
                    if constexpr (WLO) {
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j][i], fb, acc[j], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(red + (wv * 6 + j) * RED_TS + red_vec(lane)) = acc[j];
            __syncthreads();
            if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[2] += tq1 - tq0; tq0 = tq1; }
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int w = 0; w < 4; ++w) gh[g] += red[(w * 6 + g * 2 + half) * RED_TS + red_vec(lp) + (ru & 3)];
        }
        const float r = lean_sigmoid(gic[0] + gh[0]);
        const float z = lean_sigmoid(gic[1] + gh[1]);
        const float n = lean_tanh(gic[2] + r * gh[2]);
        const float h = (1.f - z) * n + z * hp;
        {
            const float hn = __uint_as_float(dpp_xor1(__float_as_uint(h)));
            if (pub_lane) {
                const unsigned soff = cbase + (unsigned)(t & 1) * panel_bytes;
                if constexpr (TF) {
                    const unsigned w = with_tag(pack2(h, hn), tag_bit((unsigned)(t + 1)) ? TAGM : 0u);
                    if (plain) __builtin_amdgcn_raw_buffer_store_b32(w, rs, pub_v, soff, 0);
                    else __builtin_amdgcn_raw_buffer_store_b32(w, rs, pub_v, soff, 16);
                    for (int rep = 0; rep < a.xsweep; ++rep) __builtin_amdgcn_raw_buffer_store_b32(w, rs, pub_v, soff, 0);
                } else {
                    const u32x2 w = {(unsigned)(t + 1), pack2(h, hn)};
                    if (plain) __builtin_amdgcn_raw_buffer_store_b64(w, rs, pub_v, soff, 0);
                    else __builtin_amdgcn_raw_buffer_store_b64(w, rs, pub_v, soff, 16);
                }
            }
        }
        if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[3] += tq1 - tq0; tq0 = tq1; }
        const float an = (1.f - z) * (1.f - n * n);
        sv[0] = h;
        sv[1] = an * gh[2] * r * (1.f - r);
        sv[2] = (hp - n) * z * (1.f - z);
        sv[3] = an * r;
        sv[4] = an;
        sv[5] = z;
        hp = h;
        if constexpr (HW) {
            // saves into parity t & 1 (the helper reads them after the next barrier); gi of step t + 1 from the ring
            float* sl = &sv_l[t & 1][0][blc][u];
            if (act) {
#pragma unroll
                for (int q = 0; q < 6; ++q) sl[q * 256] = sv[q];
            }
            const int slot = (t + 1) & 3;
#pragma unroll
            for (int g = 0; g < 3; ++g) gic[g] = gi_r[slot][blc][g * 32 + u];
        } else {
#pragma unroll
            for (int g = 0; g < 3; ++g) gic[g] = gin_[g];
        }
    }
    if (xsink == 0x9E3779B9u && a.xsweep > 1000) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (keeps the profiling loads alive)
    if constexpr (TIMED) {
        if (tid == 0 && chain == 0 && part == 0) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.status) + 8;     // byte 64 of the status header
            for (int i = 0; i < 5; ++i) dst[i] = tph[i];
            dst[5] = (unsigned long long)a.T;
        }
    }
    if constexpr (HW) {
        __syncthreads();                                   // hands the last step's saves to the helper wave
        return;
    }
    if (act) {
        const unsigned so = (unsigned)(a.T - 1) * frame_bytes, sc = (unsigned)(a.T - 1) * crow_bytes;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[0]), rs_h, own_v, so, 0);
        if (save) {
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pack2(sv[1], 0.f) & 0xffffu), rs_cf, g_v >> 1, sc, 0);
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pack2(sv[2], 0.f) & 0xffffu), rs_cf, (g_v >> 1) + (hg4 >> 1), sc, 0);
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pack2(sv[3], 0.f) & 0xffffu), rs_cf, (g_v >> 1) + hg4, sc, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[4]), rs_an, own_v, so, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[5]), rs_z, own_v, so, 0);
        }
    }
}

// ---------------------------------------------------------------------------------
// backward: dh_s = dout_s + z_{s+1} * dh_{s+1} + (dh_{s+1} * c_{s+1}) W_hh
// Same queue discipline: dout/z rows and the coefficient panel of the NEXT step are requested right
// after this step's sweep has returned; the dh save is deferred by one step.
// ---------------------------------------------------------------------------------
template <int PREC, int NKW>
__global__ __launch_bounds__(256) void gru_bwd_kernel(GruArgs a) {
    typedef typename Panel<PREC>::elem elem;
    constexpr int NPL = Panel<PREC>::NPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int Hg = a.Hg, K = 3 * Hg, KS = K >> 5, LD = K + Panel<PREC>::PAD, H = a.G * Hg;
    const int PLANE = 16 * LD;
    elem* dB = reinterpret_cast<elem*>(smem_raw);                 // [NPL][16][LD]  B operand (dh_{s+1} * c_{s+1})
    float* red = reinterpret_cast<float*>(dB + NPL * PLANE);      // [4 waves][2 tiles][64][4]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int chain, part;
    if (!claim_chain(a, a.P, chain, part)) return;
    const int grp = chain % a.G, bgi = a.bg_off + chain / a.G;
    const int b0 = bgi * a.Bg, nb = min(a.Bg, a.B - b0);
    const int u0 = part * U;
    const float* W = a.p.w_hh[grp];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.xg, 0, a.xg_bytes, 0x00020000);
    typedef typename Gran<PREC>::coef_t coef_t;
    const unsigned panel_bytes = (unsigned)(a.Bg * Hg) * (16u / Gran<PREC>::VPL);
    const unsigned cbase = (unsigned)chain * 2u * panel_bytes;
    const int nload = nb * Hg / Gran<PREC>::VPL;

    for (int i = tid; i < NPL * PLANE; i += 256) dB[i] = (elem)0.f;

    // A operand = W_hh^T slice: A[row = unit][k = gate row j] = W_hh[j][unit]
    Frag<PREC> wf[2][NKW];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = u0 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < NKW; ++i) {
            const int ks = wv + 4 * i;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[e] = ks < KS ? W[(long long)(ks * 32 + (lane >> 4) * 8 + e) * Hg + col] : 0.f;
            wf[j][i].set(v);
        }
    }

    const int u = 2 * (tid & 15);
    const int bl = tid >> 4;
    const bool active = bl < nb;
    const int half = u >> 4, ru = u & 15;
    const int lp = (ru >> 2) * 16 + bl;
    float dh0 = 0.f, dh1 = 0.f;          // dh_{s+1} of the own units
    const long long own = (long long)(b0 + bl) * a.TS * H + grp * Hg + u0 + u;     // + s*H
    const long long cf_row = (long long)a.TS * a.G * K;
    const coef_t* cf_base = reinterpret_cast<const coef_t*>(a.coefs) + ((long long)b0 * a.TS * a.G + grp) * K;   // + s*G*K
    SweepIdx<PREC> si0;
    make_idx<PREC>(si0, nload, Hg, LD, cf_row, 0, tid);
    // operands of the CURRENT step k (time s = T-1-k): dout_s, z_{s+1}, coefficient panel c_{s+1}
    float2 dd = make_float2(0.f, 0.f), zz = dd;
    CoefRegs<PREC> cr;
    if (active) dd = *reinterpret_cast<const float2*>((a.carry ? a.dh : a.dout) + own + (long long)(a.T - 1) * H);
    float2 sv_dh = make_float2(0.f, 0.f);
    bool aborted = a.dbg >= 1 && a.dbg < 8;
    const bool plain = a.dbg != 9 && team_shares_xcd(a.xid + (size_t)chain * 64, a.P, part, a.status, tid);
    __syncthreads();

    for (int k = 0; k < a.T; ++k) {
        const int s = a.T - 1 - k;
        if (k > 0)
            aborted |= sweep_panel<PREC, true>(dB, PLANE, LD, rs, cbase + (unsigned)((k - 1) & 1) * panel_bytes, nload, Hg,
                                               (unsigned)k, si0, &cr, cf_base + (long long)(s + 1) * a.G * K, cf_row,
                                               a.status, tid, aborted);
        // deferred save of dh_{s+1}; operands of step k+1 (time s-1): dout_{s-1}, z_s, c_s
        float2 ndd = make_float2(0.f, 0.f), nzz = ndd;
        if (active && k > 0) *reinterpret_cast<float2*>(a.dh + own + (long long)(s + 1) * H) = sv_dh;
        if (s > 0) {
            if (active) {
                ndd = *reinterpret_cast<const float2*>(a.dout + own + (long long)(s - 1) * H);
                nzz = *reinterpret_cast<const float2*>(a.zs + own + (long long)s * H);
            }
            load_coefs<PREC>(cr, cf_base + (long long)s * a.G * K, Hg, si0);
        }
        float mm0 = 0.f, mm1 = 0.f;
        if (k > 0) {
            f32x4 acc[2];
            acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NKW; ++i) {
                const int ks = wv + 4 * i;
                if (ks < KS && (a.dbg < 2 || a.dbg == 8)) {
                    const Frag<PREC> fb = panel_get<PREC>(dB, PLANE, (lane & 15) * LD + ks * 32 + (lane >> 4) * 8);
                    acc[0] = mma(wf[0][i], fb, acc[0]);
                    acc[1] = mma(wf[1][i], fb, acc[1]);
                }
            }
            *reinterpret_cast<f32x4*>(red + ((wv * 2 + 0) * 64 + lane) * 4) = acc[0];
            *reinterpret_cast<f32x4*>(red + ((wv * 2 + 1) * 64 + lane) * 4) = acc[1];
            __syncthreads();
            if (active) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float2 pr = *reinterpret_cast<const float2*>(red + ((w * 2 + half) * 64 + lp) * 4 + (ru & 3));
                    mm0 += pr.x; mm1 += pr.y;
                }
            }
        }
        if (active) {
            dh0 = dd.x + zz.x * dh0 + mm0;
            dh1 = dd.y + zz.y * dh1 + mm1;
            publish_pair<PREC>(rs, cbase + (unsigned)(k & 1) * panel_bytes, (unsigned)(bl * Hg + u0 + u) >> 1,
                               (unsigned)(k + 1), dh0, dh1, plain);
            sv_dh = make_float2(dh0, dh1);
            dd = ndd; zz = nzz;
        }
    }
    if (active) *reinterpret_cast<float2*>(a.dh + own) = sv_dh;          // s = 0
}

// ---------------------------------------------------------------------------------
// backward, REDUCE-SCATTER form (CRUSE_PREC_BF16, Bg = 8, Hg % 32 == 0, Hg <= 640).
// The all-gather form above makes every workgroup of a team rebuild the full [Bg x 3Hg] panel dh (.) c from the
// swept dh and 30 KB of coefficient rows per step.  Here a workgroup contracts over the gate rows it OWNS:
//   partial_p[b, n] = sum_{k in own 96 rows} (dh_s (.) c_s)[b, k] W_hh[k, n]      for ALL n in [0, Hg)
// (W_hh[own rows, :] lives in registers as A fragments: 3 k-steps x Hg/16 tiles, the same 120 KB as forward), and
// publishes the partials as {epoch, 2 x bf16} granules laid out [consumer q][producer p][clip][unit pair]: a lane's four
// units of a tile leave as ONE 16-byte store and a wave store covers 64 contiguous bytes per clip (a first layout with
// the producer index innermost made every lane of every store hit its own cache line: 3.8 us/step).  The consumer
// sums the P partials of its 32 units with P/4 16-byte loads per thread -- no coefficient panel, no
// LDS image of the exchange, a 3 KB operand panel instead of 62 KB.  Sweep bytes per step are unchanged (8 x Hg
// bf16-pair granules); publishes grow from 1 to Hg/32 stores per lane.  Same epoch / parity discipline.
// ---------------------------------------------------------------------------------
// The step loop is written for ONE WAVE PER SIMD (512-register kernel): nothing hides an instruction's issue slot,
// so the loop carries no address arithmetic (every access is a buffer instruction with a per-thread constant
// voffset and a scalar per-step soffset), no divergent control flow, and one barrier (the waves wait for their own
// granules with a wave-level vote; the operand panel is double-buffered by step parity instead of fenced).
// NP: tile PAIRS per wavefront -- a pair is the 32 units of one consumer; wave w owns pairs [w*NP, w*NP + NP), NP = ceil(P/4).
// FULL: P % 4 == 0 (Hg % 128 == 0): every pair and every producer slot exists, the validity masks fold away at
// compile time.
// LOADER WAVE: the per-step operands (dout_s, z_{s+1}, the three coefficient rows c_s of the 8 clips x 32 own units:
// 3.5 KB per step) used to be loaded by the compute threads one step ahead -- but vmcnt returns in order, so those HBM
// loads sat in front of the NEXT step's granule sweep and every step waited for them: 0.8 us of a 2.7 us step
// (CRUSE_GRU_DBG=7 drops them: 769 vs 1087 us per launch alone, tools/gru_hog_probe.py).  A fifth wavefront now streams
// them into a 4-slot LDS ring four steps ahead on its own vmcnt counter; the compute waves read them with ds_reads.
// Step anatomy after the lane-linear sweep (s_memtime stamps of workgroup 0, gru_dbg = 32; cycles of a ~3400-cycle step): sweep
// until the tags match ~1250 | partial sums, dh, panel ~500 | barrier ~240 | fragment reads, 30 MFMAs, 5 publishes ~1550, of
// which the MFMAs are ~430 and the stores ~420 (gru_dbg 33 / 37).  Measured and NOT kept, all neutral on the step although they
// shorten the last phase by 100-230 cycles (the team's pace is set by the hand-off -- 20 KB of publishes per workgroup and step
// take ~490 cycles to drain after the last one is issued, gru_dbg = 35 -- not by one workgroup's own work): issuing pair
// np + 1's MFMAs before pair np is converted and stored; four accumulators in rotation; compiling the step loop twice instead
// of choosing plain / write-through stores by two scalar branches per store; delaying the first poll by 64-384 cycles.
template <int NP, bool FULL, bool TIMED = false>
__global__ __launch_bounds__(320) void gru_bwd_rs_kernel(GruArgs a) {
    unsigned long long tph[5] = {0, 0, 0, 0, 0}, tq0 = 0, tq1 = 0;
    (void)tph; (void)tq0; (void)tq1;
    constexpr int KP = 96 + 8;                   // panel row stride (bf16): 208 B, de-phases the 16 rows of a b128 read
    constexpr int NT = 2 * NP;                   // 16-unit output tiles per wavefront
    constexpr int NL = NP;                       // 16-byte loads per thread and sweep: producers quarter*NL + j, j < NL = ceil(P/4)
    __shared__ __attribute__((aligned(16))) __bf16 panel[2][16 * KP];
    __shared__ __attribute__((aligned(16))) float op_d[4][8][32], op_z[4][8][32];       // ring slot = iteration & 3
    __shared__ __attribute__((aligned(16))) __bf16 op_c[4][8][96];
    __shared__ __attribute__((aligned(16))) float op_a[4][8][32];                       // a_n rows (only when dgi is written)
    __shared__ __attribute__((aligned(16))) float dh_l[2][8][32];                       // dh of iteration k in parity k & 1
    const int Hg = a.Hg, H = a.G * Hg, K3 = 3 * Hg, P = a.P, NTt = Hg >> 4;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int chain, part;
    if (!claim_chain(a, P, chain, part)) return;
    const int grp = chain % a.G, bgi = a.bg_off + chain / a.G;
    const int b0 = bgi * 8, nb = min(8, a.B - b0);
    const int u0 = part * U;
    const float* W = a.p.w_hh[grp];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.xg, 0, a.xg_bytes, 0x00020000);
    const unsigned cons_bytes = (unsigned)P * 8u * 16u * 8u;        // [producer P][clip 8][pair 16] granules
    const unsigned panel_bytes = (unsigned)P * cons_bytes;          // one parity of one chain
    const unsigned cbase = (unsigned)chain * 2u * panel_bytes;

    for (int i = tid; i < 2 * 16 * KP; i += 320) panel[0][i] = (__bf16)0.f;

    const unsigned frame_bytes = (unsigned)H * 4u, crow_bytes = (unsigned)(a.G * K3) * 2u;
    const long long nrow = (long long)(a.B - 1) * a.TS + a.T;           // rows reachable from the (advanced) base pointers
    const unsigned tot_f32 = (unsigned)min(nrow * H * 4, 0xffffffffll);
    const unsigned tot_cf = (unsigned)min(nrow * a.G * K3 * 2, 0xffffffffll);
    const __amdgpu_buffer_rsrc_t rs_dout = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dout), 0, tot_f32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.zs), 0, tot_f32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dh = __builtin_amdgcn_make_buffer_rsrc(a.dh, 0, tot_f32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cf = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.coefs), 0, tot_cf, 0x00020000);
    const bool nodata = a.dbg == 7 || a.dbg == 34;       // profiling: no operand streams

    if (wv == 4) {
        // ---- loader wave.  Iteration j needs dout_{T-1-j}, c_{T-1-j} and z_{T-j}; lane = (clip, 16-byte chunk).
        const int lc = lane >> 3, lq = lane & 7;                                   // dout / z: 8 clips x 8 chunks of 4 floats
        const unsigned dv = (unsigned)(((long long)(b0 + (lc < nb ? lc : 0)) * a.TS * H + grp * Hg + u0 + 4 * lq) * 4);
        unsigned cv[2], cdst[2];
        bool cok[2];
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {                                           // coef: 8 clips x 3 gates x 4 chunks of 8 bf16
            const int idx = lane + 64 * i2;
            cok[i2] = idx < 96;
            const int cl = min(idx, 95) / 12, rem = min(idx, 95) % 12, gate = rem >> 2, chk = rem & 3;
            cv[i2] = (unsigned)((((long long)(b0 + (cl < nb ? cl : 0)) * a.TS * a.G + grp) * K3 + gate * Hg + u0 + 8 * chk) * 2);
            cdst[i2] = (unsigned)(cl * 96 + gate * 32 + chk * 8);
        }
        const bool want_dgi = a.dgi != nullptr;
        const __amdgpu_buffer_rsrc_t rs_an = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ans), 0, a.ans ? tot_f32 : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_dgi = __builtin_amdgcn_make_buffer_rsrc(a.dgi, 0, a.dgi ? (a.dg_slabs == 4 ? (unsigned)min(nrow * a.G * 4 * Hg * 2, 0xffffffffll) : tot_cf) : 0u, 0x00020000);
        struct OpSet { u32x4 d, z, c0, c1, an; };
        auto issue = [&](int j, OpSet& o) {
            const u32x4 zero = {0u, 0u, 0u, 0u};
            o.d = zero; o.z = zero; o.c0 = zero; o.c1 = zero; o.an = zero;
            if (j >= a.T || nodata) return;
            const unsigned st = (unsigned)(a.T - 1 - j);
            // (a carried run: the dh of its last frame was written by the run that followed it in time)
            o.d = (j == 0 && a.carry) ? __builtin_amdgcn_raw_buffer_load_b128(rs_dh, dv, st * frame_bytes, 0)
                                      : __builtin_amdgcn_raw_buffer_load_b128(rs_dout, dv, st * frame_bytes, 0);
            if (j > 0) o.z = __builtin_amdgcn_raw_buffer_load_b128(rs_z, dv, (st + 1u) * frame_bytes, 0);
            o.c0 = __builtin_amdgcn_raw_buffer_load_b128(rs_cf, cv[0], st * crow_bytes, 0);
            if (cok[1]) o.c1 = __builtin_amdgcn_raw_buffer_load_b128(rs_cf, cv[1], st * crow_bytes, 0);
            if (want_dgi) o.an = __builtin_amdgcn_raw_buffer_load_b128(rs_an, dv, st * frame_bytes, 0);
        };
        auto put = [&](int j, const OpSet& o) {
            const int slot = j & 3;
            *reinterpret_cast<u32x4*>(&op_d[slot][lc][4 * lq]) = o.d;
            *reinterpret_cast<u32x4*>(&op_z[slot][lc][4 * lq]) = o.z;
            *reinterpret_cast<u32x4*>(&op_c[slot][0][0] + cdst[0]) = o.c0;
            if (cok[1]) *reinterpret_cast<u32x4*>(&op_c[slot][0][0] + cdst[1]) = o.c1;
            if (want_dgi) *reinterpret_cast<u32x4*>(&op_a[slot][lc][4 * lq]) = o.an;
        };
        // Results of iteration j (time T-1-j), one iteration behind the compute waves: dh_s as 16-byte stores, and -- when
        // asked for -- the gate gradients dgi_s = dh_s * (c_r, c_z, a_n) in bf16, the A operand of dX = dgi W_ih: the
        // separate gate-gradient pass then only has to make the time-major copies for the weight-gradient GEMMs, off the
        // main stream.  lane = (clip lc, unit quad lq).
        const int NSL = a.dg_slabs == 4 ? 4 : 3;
        const unsigned dgrow_bytes = (unsigned)(a.G * NSL * Hg) * 2u;
        const unsigned gi_v = (unsigned)((((long long)(b0 + (lc < nb ? lc : 0)) * a.TS * a.G + grp) * NSL * Hg + u0 + 4 * lq) * 2);
        auto flush = [&](int j) {
            if (lc >= nb) return;
            const unsigned st = (unsigned)(a.T - 1 - j);
            const float4 d4 = *reinterpret_cast<const float4*>(&dh_l[j & 1][lc][4 * lq]);
            const u32x4 dw = {__float_as_uint(d4.x), __float_as_uint(d4.y), __float_as_uint(d4.z), __float_as_uint(d4.w)};
            __builtin_amdgcn_raw_buffer_store_b128(dw, rs_dh, dv, st * frame_bytes, 0);
            if (want_dgi) {
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
                if (NSL == 4) {
                    const bf16x4_ cn = *reinterpret_cast<const bf16x4_*>(&op_c[slot][lc][64 + 4 * lq]);
                    bf16x4_ o3;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o3[e] = (__bf16)(d[e] * (float)cn[e]);
                    __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2*>(&o3), rs_dgi, gi_v + (unsigned)Hg * 6u, st * dgrow_bytes, 0);
                }
            }
        };
        OpSet s0, s1;
        issue(0, s0); issue(1, s1);
        put(0, s0); put(1, s1);
        issue(2, s0); issue(3, s1);
        if (a.dbg != 9) (void)team_shares_xcd(a.xid + (size_t)chain * 64, P, part, a.status, tid);    // mirrors the compute waves' barriers
        __syncthreads();
        // Iteration k: the set issued two iterations ago (for iteration k + 2) goes to its slot before this iteration's
        // barrier -- the compute waves read it after the NEXT barrier -- and the set is re-issued for iteration k + 4.
        for (int k = 0; k < a.T; k += 2) {
            put(k + 2, s0);
            issue(k + 4, s0);
            if (k > 0) flush(k - 1);
            if (a.T - 1 - k == 0) break;
            __syncthreads();
            put(k + 3, s1);
            issue(k + 5, s1);
            flush(k);
            if (a.T - 2 - k == 0) break;
            __syncthreads();
        }
        __syncthreads();                                // the last iteration's dh is in LDS
        flush(a.T - 1);
        return;
    }

    // A operand = W_hh[own gate rows, :]^T: A[row = output unit][k = own gate row]; k = gate*32 + unit, so k-step == gate
    bf16x8 wf[NT][3];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int gt = 2 * (wv * NP + (nt >> 1)) + (nt & 1);  // pairs beyond P (P % 4 != 0) carry zero weights
        const int n = min(gt, NTt - 1) * 16 + (lane & 15);
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                wf[nt][kk][e] = (FULL || gt < NTt) ? (__bf16)W[(long long)(kk * Hg + u0 + (lane >> 4) * 8 + e) * Hg + n] : (__bf16)0.f;
        }
    }

    // thread = (clip bl, quarter, unit quad pp): owns unit u0 + 4*pp + quarter and sums, for all four units of its
    // quad, the producers [quarter*NL, quarter*NL + NL) -- 16-byte loads; the quarters meet through two lane exchanges.
    // pp is the FASTEST lane index: eight consecutive lanes read the 128 contiguous bytes of one (producer, clip) row.  (With
    // the quarter fastest, neighbouring lanes read rows 5 KB apart -- four 64-byte requests with 16 useful bytes each where
    // one would do: the first sweep of a step took ~1700 cycles to return against ~700 in the forward kernel, whose sweep
    // is lane-linear; s_memtime stamps, gru_dbg = 32 / 35.)
    const int pp = tid & 7, quarter = (tid >> 3) & 3, bl = tid >> 5;
    const bool active = bl < nb;
    const int blc = active ? bl : 0;                                  // inactive threads shadow clip 0 (loads only)
    unsigned sweep_v[NL];
    bool sweep_ok[NL];                                                // the producer exists (P % 4 != 0: the last ones may not)
#pragma unroll
    for (int jj = 0; jj < NL; ++jj) {
        const int pr = quarter * NL + jj;
        sweep_ok[jj] = FULL || pr < P;
        sweep_v[jj] = (unsigned)part * cons_bytes + (unsigned)(((min(pr, P - 1) * 8 + blc) * 16 + 2 * pp) * 8);
    }
    // publish: the MFMA leaves clips in columns (lane & 15) < 8 only, so tiles are published in PAIRS -- the lanes of
    // the idle columns take the second tile of the pair from lane ^ 8 (a 16-lane-row rotate) and every store
    // instruction carries 64 x 16 bytes.  Offset of this lane's 16-byte piece per tile pair (clip = lane & 7):
    const int hi8 = (lane >> 3) & 1;
    unsigned pub_v[NP];
    bool pub_ok[NP];
#pragma unroll
    for (int np = 0; np < NP; ++np) {
        const int gt = 2 * (wv * NP + np) + hi8;             // this lane's tile of the pair
        pub_ok[np] = (FULL || gt < NTt) && (lane & 7) < nb;
        pub_v[np] = (unsigned)(gt >> 1) * cons_bytes +
                    (((((unsigned)part * 8u + (unsigned)(lane & 7)) * 16u) + (unsigned)(gt & 1) * 8u + (unsigned)(lane >> 4) * 2u) << 3);
    }
    const int pw = blc * KP + 4 * pp + quarter;                      // panel element of the own unit (+ gate*32)

    float dh = 0.f, dd = 0.f, zz = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;  // operands of the current step (time s)
    const int ou = 4 * pp + quarter;                                  // own unit inside the workgroup's 32
    bool nowait = a.dbg >= 1 && a.dbg < 7;
    const bool nopub = TIMED && a.dbg == 33;                          // profiling: no publishes (and no tag waits): garbage results
    nowait = nowait || nopub;
    const bool plain = a.dbg != 9 && team_shares_xcd(a.xid + (size_t)chain * 64, P, part, a.status, tid);
    __syncthreads();                                                  // ring slots 0 and 1 are filled
    dd = op_d[0][blc][ou];
    c0 = (float)op_c[0][blc][ou]; c1 = (float)op_c[0][blc][32 + ou]; c2 = (float)op_c[0][blc][64 + ou];

    for (int k = 0; k < a.T; ++k) {
        const int s = a.T - 1 - k;
        float m = 0.f;
        if constexpr (TIMED) tq0 = __builtin_amdgcn_s_memtime();
        if constexpr (TIMED) {
            if (a.dbg == 35) {                             // profiling: drain the publish stores first, stamp, then sweep
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                tq1 = __builtin_amdgcn_s_memtime(); tph[4] += tq1 - tq0; tq0 = tq1;
            }
        }
        if (k > 0) {
            const unsigned soff = cbase + (unsigned)((k - 1) & 1) * panel_bytes;
            u32x4 g[NL];
            unsigned spins = 0;
            for (;;) {
#pragma unroll
                for (int j = 0; j < NL; ++j) g[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, sweep_v[j], soff, 16);
                bool ok = true;
#pragma unroll
                for (int j = 0; j < NL; ++j) ok = ok & (((g[j].x == (unsigned)k) & (g[j].z == (unsigned)k)) | (!FULL && !sweep_ok[j]));
                if (__all(ok || !active || nowait)) break;            // wave-level: every lane's granules carry this epoch
                if (++spins >= SPIN_LIMIT) {
                    if (lane == 0) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    nowait = true;
                }
                if constexpr (TIMED) { if (a.dbg != 35) tph[4] += 1; }
                __builtin_amdgcn_s_sleep(1);
            }
            if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[0] += tq1 - tq0; tq0 = tq1; }
            float sm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const unsigned y = (FULL || sweep_ok[j]) ? g[j].y : 0u, w = (FULL || sweep_ok[j]) ? g[j].w : 0u;
                sm[0] += bf16lo(y); sm[1] += bf16hi(y);
                sm[2] += bf16lo(w); sm[3] += bf16hi(w);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sm[e] += __uint_as_float(dpp_ror8(__float_as_uint(sm[e])));          // lane ^ 8:  quarter ^ 1
                sm[e] += __uint_as_float(swz_xor16(__float_as_uint(sm[e])));         // lane ^ 16: quarter ^ 2
            }
            m = quarter == 0 ? sm[0] : quarter == 1 ? sm[1] : quarter == 2 ? sm[2] : sm[3];
        }
        dh = dd + zz * dh + m;
        __bf16* pn = panel[k & 1];
        if (active) {
            dh_l[k & 1][bl][ou] = dh;                      // the loader wave writes it (and the gate gradients) to HBM
            pn[pw] = (__bf16)(dh * c0); pn[pw + 32] = (__bf16)(dh * c1); pn[pw + 64] = (__bf16)(dh * c2);
        }
        if (s == 0) break;                                 // nothing consumes the partials of time 0
        // operands of step k+1 (time s-1): dout_{s-1}, z_s, c_{s-1}
        {                                                  // from the loader wave's ring (slot = iteration & 3)
            const int slot = (k + 1) & 3;
            dd = op_d[slot][blc][ou];
            zz = op_z[slot][blc][ou];
            c0 = (float)op_c[slot][blc][ou]; c1 = (float)op_c[slot][blc][32 + ou]; c2 = (float)op_c[slot][blc][64 + ou];
        }
        if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[1] += tq1 - tq0; tq0 = tq1; }
        __syncthreads();                                   // panel[k & 1] complete; panel[(k+1) & 1] is free again
        if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[2] += tq1 - tq0; tq0 = tq1; }
        // tile PAIRS, each finished (3 k-steps, the two tiles interleaved so no MFMA waits on its own accumulator) and
        // published before the next pair starts: the stores of the first pairs travel while the MFMAs still run.
        // acc[j]: output unit tile*16 + (lane>>4)*4 + j of clip (lane & 15); see pub_v
        bf16x8 fb[3];
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
            fb[kk] = *reinterpret_cast<const bf16x8*>(pn + (lane & 15) * KP + kk * 32 + (lane >> 4) * 8);
        const unsigned soff = cbase + (unsigned)(k & 1) * panel_bytes;
        const unsigned ep = (unsigned)(k + 1);
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            f32x4 c0_ = (f32x4){0.f, 0.f, 0.f, 0.f}, c1_ = c0_;
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                c0_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * np][kk], fb[kk], c0_, 0, 0, 0);
                c1_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * np + 1][kk], fb[kk], c1_, 0, 0, 0);
            }
            const unsigned a0 = pack2(c0_[0], c0_[1]), a1 = pack2(c0_[2], c0_[3]);
            const unsigned b0_ = pack2(c1_[0], c1_[1]), b1_ = pack2(c1_[2], c1_[3]);
            const unsigned x0 = dpp_ror8(b0_), x1 = dpp_ror8(b1_);
            const u32x4 w = {ep, hi8 ? x0 : a0, ep, hi8 ? x1 : a1};
            if (pub_ok[np] && !nopub) {
                if (plain) __builtin_amdgcn_raw_buffer_store_b128(w, rs, pub_v[np], soff, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(w, rs, pub_v[np], soff, 16);
            }
        }
        if constexpr (TIMED) { tq1 = __builtin_amdgcn_s_memtime(); tph[3] += tq1 - tq0; tq0 = tq1; }
    }
    if constexpr (TIMED) {
        if (tid == 0 && chain == 0 && part == 0) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.status) + 16;    // byte 128 of the status header
            for (int i = 0; i < 5; ++i) dst[i] = tph[i];
            dst[5] = (unsigned long long)a.T;
        }
    }
    __syncthreads();                                       // hands the last iteration's dh to the loader wave
}

// dgi = dh * (c_r, c_z, a_n), dgh = dh * (c_r, c_z, c_n); layouts [rows][G][3][Hg]
template <typename CT>
__global__ __launch_bounds__(256) void gru_gate_grads_kernel(const float* dh, const CT* coef, const float* an,
                                                             float* dgi, float* dgh, long long rows, int G, int Hg) {
    const int H = G * Hg;
    const long long n = rows * H;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long row = i / H;
        const int c = (int)(i - row * H);
        const int g = c / Hg, j = c - g * Hg;
        const float d = dh[i];
        const long long o3 = (row * G + g) * 3 * Hg + j;
        const float cr = (float)coef[o3], cz = (float)coef[o3 + Hg], cn = (float)coef[o3 + 2 * Hg];
        dgi[o3] = d * cr; dgi[o3 + Hg] = d * cz; dgi[o3 + 2 * Hg] = d * an[i];
        dgh[o3] = d * cr; dgh[o3 + Hg] = d * cz; dgh[o3 + 2 * Hg] = d * cn;
    }
}

// CRUSE_PREC_BF16 form of the gate gradients, laid out for gemm_bf16.hip.  dgi and dgh share their r and z gates
// (dgi = dh*(c_r,c_z,a_n), dgh = dh*(c_r,c_z,c_n)), so four slabs (r, z, n_i, n_h) carry both:
//   dgi [rows][G][3][Hg] bf16  (r, z, n_i)  row-major, the A operand of dX = dgi W_ih
//   dgT [ldT/64][G][4][Hg][64] bf16         time-major k-tiled (see transpose_bf16_kernel), the A operand of dW_ih
//                                           (slabs 0-2) and dW_hh (0, 1, 3)
//   db_ih[g][3*Hg] += column sums (r, z, n_i),  db_hh[g][3*Hg] += (r, z, n_h), summed from the f32 values
// One block = 64 frames x CW hidden units of one group; the time-major copy goes through an LDS image of
// [slab][unit][64 frames] bf16 whose 16-byte chunks are XOR-swizzled by (line >> 2) & 7.
struct GateBiasPtrs { float* ih[MAXG]; float* hh[MAXG]; };
constexpr int GG_RT = 4;               // 64-frame tiles per block

template <int CW>
__global__ __launch_bounds__(256) void gru_gate_grads_bf16_kernel(const float* dh, const __bf16* coef, const float* an,
                                                                  __bf16* dgi, __bf16* dgT, long long ldT,
                                                                  GateBiasPtrs bp, long long rows, int G, int Hg) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    __shared__ __attribute__((aligned(16))) unsigned img[4 * CW * 32];
    __shared__ float bsum[4][CW];
    constexpr int NCQ = CW / 4;
    const int tid = threadIdx.x;
    const int H = G * Hg, ntj = Hg / CW;
    const int g = blockIdx.y / ntj, j0 = (blockIdx.y % ntj) * CW;
    float bs[4][4];                            // bias partial sums of this thread's unit quad (fixed across items)
#pragma unroll
    for (int sl = 0; sl < 4; ++sl)
#pragma unroll
        for (int e = 0; e < 4; ++e) bs[sl][e] = 0.f;
    for (int i = tid; i < 4 * CW; i += 256) bsum[i / CW][i % CW] = 0.f;
    // RT consecutive 64-frame tiles per block: the block keeps writing the same 4*CW time-major lines
    for (int sub = 0; sub < GG_RT; ++sub) {
        const long long r0 = ((long long)blockIdx.x * GG_RT + sub) * 64;
        if (r0 >= ldT) break;
        __syncthreads();                       // previous tile's image has been drained
        for (int item = tid; item < 32 * NCQ; item += 256) {
            const int cq = item % NCQ, q = item / NCQ;      // 256 % NCQ == 0: cq is the same for every item of a thread
            const int j = j0 + cq * 4;
            float v[2][4][4];                  // [row of the pair][slab][unit]
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const long long r = r0 + 2 * q + rr;
                if (r < rows) {
                    const float4 d4 = *reinterpret_cast<const float4*>(dh + r * H + g * Hg + j);
                    const float4 a4 = *reinterpret_cast<const float4*>(an + r * H + g * Hg + j);
                    const long long o3 = (r * G + g) * 3 * Hg + j;
                    const bf16x4 cr = *reinterpret_cast<const bf16x4*>(coef + o3);
                    const bf16x4 cz = *reinterpret_cast<const bf16x4*>(coef + o3 + Hg);
                    const bf16x4 cn = *reinterpret_cast<const bf16x4*>(coef + o3 + 2 * Hg);
                    const float d[4] = {d4.x, d4.y, d4.z, d4.w}, a[4] = {a4.x, a4.y, a4.z, a4.w};
                    bf16x4 o0, o1, o2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[rr][0][e] = d[e] * (float)cr[e]; v[rr][1][e] = d[e] * (float)cz[e];
                        v[rr][2][e] = d[e] * a[e];         v[rr][3][e] = d[e] * (float)cn[e];
                        o0[e] = (__bf16)v[rr][0][e]; o1[e] = (__bf16)v[rr][1][e]; o2[e] = (__bf16)v[rr][2][e];
                    }
                    if (dgi) {                 // null: the backward recurrence has written dgi itself (cruse_gru_seq_bwd_on)
                        *reinterpret_cast<bf16x4*>(dgi + o3) = o0;
                        *reinterpret_cast<bf16x4*>(dgi + o3 + Hg) = o1;
                        *reinterpret_cast<bf16x4*>(dgi + o3 + 2 * Hg) = o2;
                    }
                } else {
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[rr][sl][e] = 0.f;
                }
            }
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int line = sl * CW + cq * 4 + e;
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                    bf16x2 pr;
                    pr[0] = (__bf16)v[0][sl][e]; pr[1] = (__bf16)v[1][sl][e];
                    img[line * 32 + (((q >> 2) ^ ((line >> 2) & 7)) << 2) + (q & 3)] = *reinterpret_cast<unsigned*>(&pr);
                    bs[sl][e] += v[0][sl][e] + v[1][sl][e];
                }
        }
        __syncthreads();
        for (int item = tid; dgT && item < 4 * CW * 8; item += 256) {
            const int k = item & 7, line = item >> 3;
            const int sl = line / CW, col = line % CW;
            const uint4 w = *reinterpret_cast<const uint4*>(&img[line * 32 + ((k ^ ((line >> 2) & 7)) << 2)]);
            *reinterpret_cast<uint4*>(dgT + (r0 >> 6) * ((long long)4 * G * Hg * 64) + ((long long)(g * 4 + sl) * Hg + j0 + col) * 64 + k * 8) = w;
        }
    }
    {
        const int cq = tid % NCQ;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(&bsum[sl][cq * 4 + e], bs[sl][e]);
    }
    __syncthreads();
    for (int i = tid; i < 4 * CW; i += 256) {
        const int sl = i / CW, j = j0 + i % CW;
        const float val = bsum[sl][i % CW];
        if (sl < 2) {
            if (bp.ih[g]) atomicAdd(bp.ih[g] + sl * Hg + j, val);
            if (bp.hh[g]) atomicAdd(bp.hh[g] + sl * Hg + j, val);
        } else if (sl == 2) {
            if (bp.ih[g]) atomicAdd(bp.ih[g] + 2 * Hg + j, val);
        } else {
            if (bp.hh[g]) atomicAdd(bp.hh[g] + 2 * Hg + j, val);
        }
    }
}

int num_cus() {
    static int cached = 0;
    if (cached == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached = n;
    }
    return cached;
}

struct Plan { int Bg, P, nbg, bg_per_launch, nlaunch; bool wide; };

}  // namespace
namespace cruse_gru {       // gru_tf.hip: the tag-free hand-off kernels (bf16, chains of 8, Hg in {160, 320, 640})
bool fwd_tf_eligible(int Bg, int Hg, int prec, bool has_h0, bool gi_bf16);
bool bwd_tf_eligible(int Bg, int Hg, int prec);
int dispatch_fwd_tf(const GruArgs& a, int grid, bool wlo, hipStream_t s);
int dispatch_bwd_tf(const GruArgs& a, int grid, hipStream_t s);
// gru_w16.hip: wide chains (16 clips), tag-free register-direct hand-off, epochs continuing across time chunks
bool w16_eligible(int Hg, int prec);
size_t w16_panel_bytes_per_parity(int Hg, bool fwd);
int dispatch_fwd_w16(const GruArgs& a, int grid, hipStream_t s);
int dispatch_bwd_w16(const GruArgs& a, int grid, hipStream_t s);
}
namespace {

bool bwd_rs_eligible(int Bg, int Hg, int prec);
bool fwd_lean_eligible(int Bg, int Hg, int prec);

// Chains of 8 clips while the batch's chains fit the CUs.  A larger batch: WIDE chains (16 clips per chain, one launch of
// half the workgroups, gru_w16.hip) where that kernel applies (wide_ok) and needs fewer launches; otherwise several launches on
// chains of 8 (the generic kernels: chains of 16).  chain_clips = 16 asks for wide chains outright (the GGRU wavefront: both
// layers' recurrences co-resident).
int make_plan(int B, int G, int Hg, int prec, bool fwd, int chain_clips, bool wide_ok, Plan& pl) {
    pl.P = Hg / U;
    const int maxblk = num_cus();
    if (G * pl.P > maxblk) return -1;
    wide_ok = wide_ok && w16_eligible(Hg, prec);
    pl.wide = chain_clips == 16 && wide_ok;
    pl.Bg = pl.wide ? 16 : 8;
    pl.nbg = cdiv(B, pl.Bg);
    int per_launch = ((maxblk / pl.P) / 8 * 8) / G;        // chains per launch padded to a multiple of 8
    if (per_launch < 1) per_launch = 1;
    if (pl.nbg * G * pl.P > maxblk && chain_clips != 8 && pl.Bg == 8) {
        const bool fast8 = fwd ? fwd_lean_eligible(8, Hg, prec) : bwd_rs_eligible(8, Hg, prec);
        const int n8 = cdiv(pl.nbg, per_launch), n16 = cdiv(cdiv(B, 16), per_launch);
        if (wide_ok && n16 < n8) { pl.wide = true; pl.Bg = 16; pl.nbg = cdiv(B, 16); }
        else if (!fast8) { pl.Bg = 16; pl.nbg = cdiv(B, 16); }          // the generic kernels serve chains of 16
    }
    pl.bg_per_launch = per_launch;
    if (pl.bg_per_launch > pl.nbg) pl.bg_per_launch = pl.nbg;
    pl.nlaunch = cdiv(pl.nbg, pl.bg_per_launch);
    return 0;
}

constexpr int MAX_LAUNCH_TICKETS = 64;      // launches of one call that get their own ticket counters
// (sized for chains of 8 or of 16 clips: a 16-clip chain takes the room of two chains of 8)
int chains8(int B) { return 2 * cdiv(B, 16); }
size_t xid_bytes_total(int B, int G) { return (size_t)chains8(B) * G * 64 * 8; }

// granules (8 bytes) of one parity of one chain: all-gather forms keep up to 16 rows of Hg values; the
// reduce-scatter backward keeps [consumer P][clip 8][pair 16][producer P]
bool bwd_rs_eligible(int Bg, int Hg, int prec) {
    if (cruse_opt("gru_bwd_rs", 1) == 0) return false;    // A/B switch (tests, probes)
    return prec == CRUSE_PREC_BF16 && Bg == 8 && Hg % 32 == 0 && Hg <= 640;
}
size_t rs_gran_per_parity(int Hg) { const size_t P = Hg / 32; return P * 8 * 16 * P; }
size_t xg_bytes_total(int B, int G, int Hg) {
    size_t per = (size_t)16 * Hg;
    if (Hg % 32 == 0 && Hg <= 640 && rs_gran_per_parity(Hg) > per) per = rs_gran_per_parity(Hg);
    return (size_t)chains8(B) * G * 2 * per * 8;
}


bool fwd_lean_eligible(int Bg, int Hg, int prec) {
    if (cruse_opt("gru_fwd_lean", 1) == 0) return false;   // A/B switch (tests, probes)
    return prec == CRUSE_PREC_BF16 && Bg == 8 && Hg % 32 == 0 && Hg <= 640;
}
// W_hh low plane in the lean forward recurrence: CRUSE_GRU_WLO = 1 always, 0 never; default: when the second plane is
// nearly free (Hg <= 320: at most 48 extra registers and 12 extra MFMAs per wave and step)
bool fwd_wlo(int Hg) {
    const int e = cruse_opt("gru_wlo", -1);
    if (e >= 0) return e != 0;
    return Hg <= 320;
}
template <bool WLO, bool GIB>
int dispatch_fwd_lean_w(const GruArgs& a, int grid, size_t lds, hipStream_t s) {
    const int n = (a.Hg + 127) / 128;            // = k-steps per wave = sweep slots per thread
    if (a.Hg % 128 == 0) {
        switch (n) {
            case 1: return launch_one(gru_fwd_lean_kernel<1, 1, true, WLO, false, GIB>, a, grid, lds, s, "gru_seq_fwd", (WLO && 1 > 3) ? 256 : 320);
            case 2: return launch_one(gru_fwd_lean_kernel<2, 2, true, WLO, false, GIB>, a, grid, lds, s, "gru_seq_fwd", (WLO && 2 > 3) ? 256 : 320);
            case 3: return launch_one(gru_fwd_lean_kernel<3, 3, true, WLO, false, GIB>, a, grid, lds, s, "gru_seq_fwd", (WLO && 3 > 3) ? 256 : 320);
            case 4: return launch_one(gru_fwd_lean_kernel<4, 4, true, WLO, false, GIB>, a, grid, lds, s, "gru_seq_fwd", (WLO && 4 > 3) ? 256 : 320);
            default: return launch_one(gru_fwd_lean_kernel<5, 5, true, WLO, false, GIB>, a, grid, lds, s, "gru_seq_fwd", (WLO && 5 > 3) ? 256 : 320);
        }
    }
    switch (n) {
        case 1: return launch_one(gru_fwd_lean_kernel<1, 1, false, WLO, false, GIB>, a, grid, lds, s, "gru_seq_fwd", (WLO && 1 > 3) ? 256 : 320);
        case 2: return launch_one(gru_fwd_lean_kernel<2, 2, false, WLO, false, GIB>, a, grid, lds, s, "gru_seq_fwd", (WLO && 2 > 3) ? 256 : 320);
        case 3: return launch_one(gru_fwd_lean_kernel<3, 3, false, WLO, false, GIB>, a, grid, lds, s, "gru_seq_fwd", (WLO && 3 > 3) ? 256 : 320);
        case 4: return launch_one(gru_fwd_lean_kernel<4, 4, false, WLO, false, GIB>, a, grid, lds, s, "gru_seq_fwd", (WLO && 4 > 3) ? 256 : 320);
        default: return launch_one(gru_fwd_lean_kernel<5, 5, false, WLO, false, GIB>, a, grid, lds, s, "gru_seq_fwd", (WLO && 5 > 3) ? 256 : 320);
    }
}
int dispatch_fwd_lean(const GruArgs& a, int grid, size_t lds, hipStream_t s) {
    // Hg = 640: the K-split step on the tag-free hand-off with the register-direct sweep (library option gru_tf = 0: the tagged hand-off)
    const bool tf640 = a.Hg == 640 && !fwd_wlo(a.Hg) && a.h0 == nullptr && cruse_opt("gru_tf", 1) != 0;
    if (a.gi_bf16 == 2) {
        // f16 gi rows: built for the bench step's kernel only (Hg = 640, chains of 8, tag-free register-direct hand-off, h0 = 0)
        if (!tf640) { cruse_set_error("gru_seq_fwd: f16 gi rows are served at Hg = 640 with h0 = NULL on chains of 8 only (cruse_gru_plan)"); return CRUSE_E_SHAPE; }
        return launch_one(gru_fwd_lean_kernel<5, 5, true, false, false, true, true, true, true>, a, grid, lds, s, "gru_seq_fwd", 320);
    }
    if (tf640) {
        if (a.dbg == 32) return launch_one(gru_fwd_lean_kernel<5, 5, true, false, true, false, true, true>, a, grid, lds, s, "gru_seq_fwd", 320);
        return launch_one(gru_fwd_lean_kernel<5, 5, true, false, false, false, true, true>, a, grid, lds, s, "gru_seq_fwd", 320);
    }
    if (a.dbg == 32 && a.Hg == 640 && !fwd_wlo(a.Hg))
        return launch_one(gru_fwd_lean_kernel<5, 5, true, false, true>, a, grid, lds, s, "gru_seq_fwd", 320);
    return fwd_wlo(a.Hg) ? dispatch_fwd_lean_w<true, false>(a, grid, lds, s) : dispatch_fwd_lean_w<false, false>(a, grid, lds, s);
}

template <int PREC>
int dispatch_fwd(const GruArgs& a, int grid, size_t lds, hipStream_t s) {
    const int nkw = cdiv(a.Hg / 32, 4);
    if (nkw <= 2) return launch_one(gru_fwd_kernel<PREC, 2>, a, grid, lds, s, "gru_seq_fwd");
    if (nkw <= 3) return launch_one(gru_fwd_kernel<PREC, 3>, a, grid, lds, s, "gru_seq_fwd");
    if (nkw <= 5) return launch_one(gru_fwd_kernel<PREC, 5>, a, grid, lds, s, "gru_seq_fwd");
    return launch_one(gru_fwd_kernel<PREC, 8>, a, grid, lds, s, "gru_seq_fwd");
}
template <int PREC>
int dispatch_bwd(const GruArgs& a, int grid, size_t lds, hipStream_t s) {
    const int nkw = cdiv(3 * a.Hg / 32, 4);
    if (nkw <= 4) return launch_one(gru_bwd_kernel<PREC, 4>, a, grid, lds, s, "gru_seq_bwd");
    if (nkw <= 8) return launch_one(gru_bwd_kernel<PREC, 8>, a, grid, lds, s, "gru_seq_bwd");
    if (nkw <= 15) return launch_one(gru_bwd_kernel<PREC, 15>, a, grid, lds, s, "gru_seq_bwd");
    return launch_one(gru_bwd_kernel<PREC, 24>, a, grid, lds, s, "gru_seq_bwd");
}

int dispatch_bwd_rs(const GruArgs& a, int grid, hipStream_t s) {
    const int P = a.Hg / 32, np = (P + 3) / 4;   // tile pairs per wavefront
    if (a.dbg >= 32 && a.dbg <= 35 && a.Hg == 640) return launch_one(gru_bwd_rs_kernel<5, true, true>, a, grid, 0, s, "gru_seq_bwd", 320);
    if (P % 4 == 0) {
        switch (np) {
            case 1: return launch_one(gru_bwd_rs_kernel<1, true>, a, grid, 0, s, "gru_seq_bwd", 320);
            case 2: return launch_one(gru_bwd_rs_kernel<2, true>, a, grid, 0, s, "gru_seq_bwd", 320);
            case 3: return launch_one(gru_bwd_rs_kernel<3, true>, a, grid, 0, s, "gru_seq_bwd", 320);
            case 4: return launch_one(gru_bwd_rs_kernel<4, true>, a, grid, 0, s, "gru_seq_bwd", 320);
            default: return launch_one(gru_bwd_rs_kernel<5, true>, a, grid, 0, s, "gru_seq_bwd", 320);
        }
    }
    switch (np) {
        case 1: return launch_one(gru_bwd_rs_kernel<1, false>, a, grid, 0, s, "gru_seq_bwd", 320);
        case 2: return launch_one(gru_bwd_rs_kernel<2, false>, a, grid, 0, s, "gru_seq_bwd", 320);
        case 3: return launch_one(gru_bwd_rs_kernel<3, false>, a, grid, 0, s, "gru_seq_bwd", 320);
        case 4: return launch_one(gru_bwd_rs_kernel<4, false>, a, grid, 0, s, "gru_seq_bwd", 320);
        default: return launch_one(gru_bwd_rs_kernel<5, false>, a, grid, 0, s, "gru_seq_bwd", 320);
    }
}

int check_common(int B, int T, int G, int Hg, int prec, const char* name) {
    CRUSE_REQUIRE(B > 0 && T > 0 && G > 0 && G <= MAXG, CRUSE_E_SHAPE, "%s: bad shape B=%d T=%d G=%d", name, B, T, G);
    CRUSE_REQUIRE(Hg % 32 == 0 && Hg >= 32 && Hg <= 1024, CRUSE_E_SHAPE,
                  "%s: hidden size per group %d must be a multiple of 32 in [32,1024]", name, Hg);
    CRUSE_REQUIRE(prec == CRUSE_PREC_F32 || prec == CRUSE_PREC_BF16X3 || prec == CRUSE_PREC_BF16, CRUSE_E_DTYPE,
                  "%s: unknown precision %d", name, prec);
    return CRUSE_OK;
}

template <bool FWD>
int run_launches(GruArgs& a, const Plan& pl, int G, int Hg, int prec, void* panels, unsigned* status, int xcd_rot,
                 size_t lds, hipStream_t s) {
    a.status = status;
    a.xcd_rot = xcd_rot & 7;
    char* xid_base = (char*)panels;
    char* xg_base = xid_base + xid_bytes_total(a.B, G);
    unsigned* tickets_base = (unsigned*)(xg_base + xg_bytes_total(a.B, G, Hg));        // [launch][8]
    a.Bg = pl.Bg; a.P = pl.P;
    a.dbg = cruse_opt("gru_dbg", 0);
    int rc = CRUSE_OK;
    CRUSE_REQUIRE(pl.nlaunch <= MAX_LAUNCH_TICKETS, CRUSE_E_SHAPE, "gru_seq: batch %d needs %d launches (max %d)", a.B, pl.nlaunch,
                  MAX_LAUNCH_TICKETS);
    for (int L = 0; L < pl.nlaunch; ++L) {
        const int bg_off = L * pl.bg_per_launch;
        const int nbg_here = (pl.nbg - bg_off) < pl.bg_per_launch ? (pl.nbg - bg_off) : pl.bg_per_launch;
        a.bg_off = bg_off;
        a.nchains = nbg_here * G;
        // every launch gets its own panel region: chain index inside the launch + offset
        a.xid = (unsigned long long*)xid_base + (size_t)bg_off * G * 64;
        a.tickets = tickets_base + (size_t)L * 8;
        const bool rs_form = !FWD && !pl.wide && bwd_rs_eligible(pl.Bg, Hg, prec);
        const size_t gpp = pl.wide ? w16_panel_bytes_per_parity(Hg, FWD) / 8 : rs_form ? rs_gran_per_parity(Hg) : (size_t)pl.Bg * Hg;      // granules per parity and chain
        a.xg = (unsigned long long*)xg_base + (size_t)bg_off * G * 2 * gpp;
        a.xg_bytes = (unsigned)((size_t)a.nchains * 2 * gpp * 8);
        const int grid = cdiv(a.nchains, 8) * 8 * pl.P;
        if (FWD && a.gi_bf16 == 2 && (pl.wide || !fwd_lean_eligible(pl.Bg, Hg, prec))) {
            cruse_set_error("gru_seq_fwd: f16 gi rows are served at Hg = 640, CRUSE_PREC_BF16, on chains of 8 clips only (B <= 96: cruse_gru_plan)");
            return CRUSE_E_SHAPE;
        }
        if (FWD && pl.wide) {
            a.poll_delay = 0;
            rc = dispatch_fwd_w16(a, grid, s);
        } else if (FWD && fwd_lean_eligible(pl.Bg, Hg, prec)) {
            if (fwd_tf_eligible(pl.Bg, Hg, prec, a.h0 != nullptr, a.gi_bf16 != 0)) { a.poll_delay = cruse_opt("gru_poll_fwd", 8); rc = dispatch_fwd_tf(a, grid, fwd_wlo(Hg), s); }
            else { a.poll_delay = cruse_opt("gru_poll_fwd", 0); rc = dispatch_fwd_lean(a, grid, lds, s); }
        } else if (FWD) {
            if (prec == CRUSE_PREC_F32) rc = dispatch_fwd<CRUSE_PREC_F32>(a, grid, lds, s);
            else if (prec == CRUSE_PREC_BF16X3) rc = dispatch_fwd<CRUSE_PREC_BF16X3>(a, grid, lds, s);
            else rc = dispatch_fwd<CRUSE_PREC_BF16>(a, grid, lds, s);
        } else if (pl.wide) {
            a.poll_delay = 6;                         // (tools/gru_wide_probe.py: 1.62 us per step at 5..8 periods, 1.67 at 0, 1.71 at 12)
            rc = dispatch_bwd_w16(a, grid, s);
        } else if (rs_form) {
            // (the tag-free kernel's panels are half the size of the reduce-scatter kernel's: the same regions hold them)
            if (bwd_tf_eligible(pl.Bg, Hg, prec)) { a.poll_delay = cruse_opt("gru_poll_bwd", Hg == 640 ? 5 : 7); rc = dispatch_bwd_tf(a, grid, s); }
            else rc = dispatch_bwd_rs(a, grid, s);
        } else {
            if (prec == CRUSE_PREC_F32) rc = dispatch_bwd<CRUSE_PREC_F32>(a, grid, lds, s);
            else if (prec == CRUSE_PREC_BF16X3) rc = dispatch_bwd<CRUSE_PREC_BF16X3>(a, grid, lds, s);
            else rc = dispatch_bwd<CRUSE_PREC_BF16>(a, grid, lds, s);
        }
        if (rc) return rc;
    }
    return rc;
}


}  // namespace

extern "C" int cruse_gru_plan(int B, int G, int Hg, int prec, int fwd, int* out) {
    CRUSE_REQUIRE(out != nullptr && B >= 1 && G >= 1 && Hg >= 32 && Hg % 32 == 0, CRUSE_E_SHAPE, "gru_plan: B = %d, G = %d, Hg = %d", B, G, Hg);
    Plan pl;
    CRUSE_REQUIRE(make_plan(B, G, Hg, prec, fwd != 0, 0, true, pl) == 0, CRUSE_E_SHAPE, "gru_plan: G*Hg/32 exceeds the CU count");
    out[0] = pl.Bg; out[1] = pl.nbg; out[2] = pl.bg_per_launch; out[3] = pl.nlaunch; out[4] = pl.P; out[5] = pl.wide ? 1 : 0;
    return CRUSE_OK;
}

extern "C" size_t cruse_gru_ws_bytes(int B, int G, int Hg) {
    return 256 + xid_bytes_total(B, G) + xg_bytes_total(B, G, Hg) + (size_t)MAX_LAUNCH_TICKETS * 8 * sizeof(unsigned);
}

static int gru_seq_fwd_impl(const float* gi, const float* const* w_hh, const float* const* b_hh,
                            float* h, void* coef, float* an, float* z, const float* h0, long long h0_bstride,
                            int B, int T, int TS, int G, int Hg, int prec, int chain_clips, void* panels,
                            int panels_zeroed, unsigned* status, int xcd_rot, void* stream, int gi_dtype) {
    int rc = check_common(B, T, G, Hg, prec, "gru_seq_fwd");
    CRUSE_REQUIRE(chain_clips == 0 || chain_clips == 8 || chain_clips == 16, CRUSE_E_SHAPE, "gru_seq_fwd: chain_clips = %d (0, 8, 16)", chain_clips);
    if (rc) return rc;
    CRUSE_REQUIRE(TS >= T, CRUSE_E_SHAPE, "gru_seq_fwd: clip stride %d frames < %d steps", TS, T);
    CRUSE_REQUIRE(h0 == nullptr || (h0_bstride >= (long long)G * Hg && ((uintptr_t)h0 % 16) == 0 && h0_bstride % 4 == 0), CRUSE_E_SHAPE,
                  "gru_seq_fwd: h0 needs a clip stride >= G*Hg (multiple of 4 floats) and 16-byte alignment");
    CRUSE_REQUIRE((coef == nullptr) == (an == nullptr) && (coef == nullptr) == (z == nullptr), CRUSE_E_SHAPE,
                  "gru_seq_fwd: coef, an, z must all be given or all be NULL");
    Plan pl;
    // wide chains: f32 gi rows; their tag-free hand-off needs |h| < 1, so an initial state is taken only where the caller asks for
    // wide chains outright and thereby vouches for |h0| < 1 (the continuation of a sequence that started from zero)
    const bool wide_ok = h0 == nullptr || chain_clips == 16;
    CRUSE_REQUIRE(make_plan(B, G, Hg, prec, true, chain_clips, wide_ok, pl) == 0, CRUSE_E_SHAPE, "gru_seq_fwd: G*Hg/32 exceeds the CU count");
    hipStream_t s = (hipStream_t)stream;
    CRUSE_REQUIRE(panels != nullptr && status != nullptr, CRUSE_E_SHAPE, "gru_seq_fwd: workspace / status pointer is NULL");
    // (the status word is sticky: never cleared here.  panels_zeroed: the caller cleared the scratch itself -- e.g. the scratches of
    //  all four recurrences of a training step with one launch at the top of the step instead of a memset in front of each)
    if (!panels_zeroed) { int zrc = cruse_zero_async(panels, cruse_gru_ws_bytes(B, G, Hg) - 256, s, "gru_seq_fwd memset"); if (zrc) return zrc; }
    GruArgs a = {};
    a.gi = gi; a.h = h; a.coef = coef; a.an = an; a.z = z;
    for (int g = 0; g < G; ++g) { a.p.w_hh[g] = w_hh[g]; a.p.b_hh[g] = b_hh[g]; }
    a.B = B; a.T = T; a.G = G; a.Hg = Hg;
    a.TS = TS; a.h0 = h0; a.h0_bs = h0_bstride;
    a.gi_bf16 = gi_dtype == CRUSE_DT_F16 ? 2 : 0;
    const size_t esz = prec == CRUSE_PREC_F32 ? 4 : 2, npl = prec == CRUSE_PREC_BF16X3 ? 2 : 1;
    const size_t lds = (size_t)16 * (Hg + (prec == CRUSE_PREC_F32 ? 4 : 8)) * esz * npl + 4 * 6 * RED_TS * sizeof(float);
    return run_launches<true>(a, pl, G, Hg, prec, panels, status, xcd_rot, lds, s);
}

extern "C" int cruse_gru_seq_fwd_ex(const float* gi, const float* const* w_hh, const float* const* b_hh,
                                    float* h, void* coef, float* an, float* z, const float* h0, long long h0_bstride,
                                    int B, int T, int TS, int G, int Hg, int prec, int chain_clips, void* panels,
                                    int panels_zeroed, unsigned* status, int xcd_rot, void* stream) {
    return gru_seq_fwd_impl(gi, w_hh, b_hh, h, coef, an, z, h0, h0_bstride, B, T, TS, G, Hg, prec, chain_clips, panels, panels_zeroed, status,
                            xcd_rot, stream, CRUSE_DT_F32);
}

// gi rows stored as IEEE f16 ([B,T,G,3*Hg] halves: cruse_gemm_nt_out16 writes them): half the bytes of the largest tensor of the forward pass,
// written once and read once.  Served where the bench step runs -- CRUSE_PREC_BF16, Hg = 640, h0 = NULL, chains of 8 clips (B <= 96) --
// and refused with CRUSE_E_SHAPE elsewhere (the caller keeps f32 rows there).  gi_dtype: CRUSE_DT_F16 or CRUSE_DT_F32 (= cruse_gru_seq_fwd_ex).
extern "C" int cruse_gru_seq_fwd_gi16(const void* gi, int gi_dtype, const float* const* w_hh, const float* const* b_hh,
                                      float* h, void* coef, float* an, float* z,
                                      int B, int T, int TS, int G, int Hg, int prec, void* panels,
                                      int panels_zeroed, unsigned* status, int xcd_rot, void* stream) {
    CRUSE_REQUIRE(gi_dtype == CRUSE_DT_F16 || gi_dtype == CRUSE_DT_F32, CRUSE_E_DTYPE, "gru_seq_fwd_gi16: gi_dtype %d (CRUSE_DT_F32, CRUSE_DT_F16)", gi_dtype);
    CRUSE_REQUIRE(gi_dtype == CRUSE_DT_F32 || (prec == CRUSE_PREC_BF16 && Hg == 640), CRUSE_E_SHAPE,
                  "gru_seq_fwd_gi16: f16 gi rows need CRUSE_PREC_BF16 and Hg = 640");
    return gru_seq_fwd_impl(reinterpret_cast<const float*>(gi), w_hh, b_hh, h, coef, an, z, nullptr, 0, B, T, TS, G, Hg, prec, 0, panels,
                            panels_zeroed, status, xcd_rot, stream, gi_dtype);
}

extern "C" int cruse_gru_seq_fwd_on(const float* gi, const float* const* w_hh, const float* const* b_hh,
                                    float* h, void* coef, float* an, float* z,
                                    int B, int T, int G, int Hg, int prec, void* panels, unsigned* status, int xcd_rot,
                                    void* stream) {
    return cruse_gru_seq_fwd_ex(gi, w_hh, b_hh, h, coef, an, z, nullptr, 0, B, T, T, G, Hg, prec, 0, panels, 0, status, xcd_rot, stream);
}

extern "C" int cruse_gru_seq_fwd(const float* gi, const float* const* w_hh, const float* const* b_hh,
                                 float* h, void* coef, float* an, float* z,
                                 int B, int T, int G, int Hg, int prec, void* ws, void* stream) {
    return cruse_gru_seq_fwd_on(gi, w_hh, b_hh, h, coef, an, z, B, T, G, Hg, prec, ws ? (char*)ws + 256 : nullptr, (unsigned*)ws, 0,
                                stream);
}

extern "C" int cruse_gru_gate_grads_bf16(const float* dh, const void* coef, const float* an, void* dgi, void* dgT,
                                         long long ldT, float* const* db_ih, float* const* db_hh,
                                         long long rows, int G, int Hg, void* stream);

extern "C" int cruse_gru_seq_bwd_ex(const float* dout, const float* const* w_hh, const void* coef, const float* z,
                                    float* dh, const float* an, void* dgi, int dg_slabs, int carry, int B, int T, int TS, int G,
                                    int Hg, int prec, int chain_clips, void* panels, int panels_zeroed, unsigned* status, int xcd_rot,
                                    void* stream) {
    int rc = check_common(B, T, G, Hg, prec, "gru_seq_bwd");
    CRUSE_REQUIRE(chain_clips == 0 || chain_clips == 8 || chain_clips == 16, CRUSE_E_SHAPE, "gru_seq_bwd: chain_clips = %d (0, 8, 16)", chain_clips);
    if (rc) return rc;
    CRUSE_REQUIRE(TS >= T, CRUSE_E_SHAPE, "gru_seq_bwd: clip stride %d frames < %d steps", TS, T);
    Plan pl;
    CRUSE_REQUIRE(make_plan(B, G, Hg, prec, false, chain_clips, true, pl) == 0, CRUSE_E_SHAPE, "gru_seq_bwd: G*Hg/32 exceeds the CU count");
    hipStream_t s = (hipStream_t)stream;
    CRUSE_REQUIRE(panels != nullptr && status != nullptr, CRUSE_E_SHAPE, "gru_seq_bwd: workspace / status pointer is NULL");
    if (!panels_zeroed) { int zrc = cruse_zero_async(panels, cruse_gru_ws_bytes(B, G, Hg) - 256, s, "gru_seq_bwd memset"); if (zrc) return zrc; }   // the status word is sticky: never cleared here
    GruArgs a = {};
    a.dout = dout; a.coefs = coef; a.zs = z; a.dh = dh;
    for (int g = 0; g < G; ++g) { a.p.w_hh[g] = w_hh[g]; a.p.b_hh[g] = nullptr; }
    a.B = B; a.T = T; a.G = G; a.Hg = Hg;
    a.TS = TS; a.carry = carry ? 1 : 0;
    const size_t esz = prec == CRUSE_PREC_F32 ? 4 : 2, npl = prec == CRUSE_PREC_BF16X3 ? 2 : 1;
    const size_t lds = (size_t)16 * (3 * Hg + (prec == CRUSE_PREC_F32 ? 4 : 8)) * esz * npl + 4 * 2 * 64 * 4 * sizeof(float);
    CRUSE_REQUIRE(lds <= 160 * 1024, CRUSE_E_SHAPE, "gru_seq_bwd: Hg=%d needs %zu B of LDS", Hg, lds);
    CRUSE_REQUIRE(dgi == nullptr || (an != nullptr && prec == CRUSE_PREC_BF16), CRUSE_E_SHAPE,
                  "gru_seq_bwd: dgi needs the a_n rows and CRUSE_PREC_BF16");
    // the reduce-scatter kernel's loader wave writes dgi itself; the other kernels are followed by the gate-gradient pass
    const bool in_kernel = dgi != nullptr && (pl.wide || bwd_rs_eligible(pl.Bg, Hg, prec));
    CRUSE_REQUIRE(dg_slabs == 3 || (dg_slabs == 4 && (dgi == nullptr || in_kernel)), CRUSE_E_SHAPE,
                  "gru_seq_bwd: dg_slabs = %d (3, or 4 with the reduce-scatter kernels: bf16, Hg <= 640)", dg_slabs);
    a.ans = in_kernel ? an : nullptr;
    a.dgi = in_kernel ? dgi : nullptr;
    a.dg_slabs = dg_slabs;
    rc = run_launches<false>(a, pl, G, Hg, prec, panels, status, xcd_rot, lds, s);
    if (rc || dgi == nullptr || in_kernel) return rc;
    CRUSE_REQUIRE(TS == T, CRUSE_E_SHAPE, "gru_seq_bwd: dgi on a sub-sequence needs the reduce-scatter kernel (bf16, Hg <= 640)");
    const long long rows = (long long)B * T;
    return cruse_gru_gate_grads_bf16(dh, coef, an, dgi, nullptr, (rows + 63) / 64 * 64, nullptr, nullptr, rows, G, Hg, stream);
}

extern "C" int cruse_gru_seq_bwd_on(const float* dout, const float* const* w_hh, const void* coef, const float* z,
                                    float* dh, const float* an, void* dgi, int B, int T, int G, int Hg, int prec,
                                    void* panels, unsigned* status, int xcd_rot, void* stream) {
    return cruse_gru_seq_bwd_ex(dout, w_hh, coef, z, dh, an, dgi, 3, 0, B, T, T, G, Hg, prec, 0, panels, 0, status, xcd_rot, stream);
}

extern "C" int cruse_gru_seq_bwd(const float* dout, const float* const* w_hh, const void* coef, const float* z,
                                 float* dh, int B, int T, int G, int Hg, int prec, void* ws, void* stream) {
    return cruse_gru_seq_bwd_on(dout, w_hh, coef, z, dh, nullptr, nullptr, B, T, G, Hg, prec, ws ? (char*)ws + 256 : nullptr,
                                (unsigned*)ws, 0, stream);
}

extern "C" int cruse_gru_gate_grads(const float* dh, const void* coef, const float* an, float* dgi, float* dgh,
                                    long long rows, int G, int Hg, int prec, void* stream) {
    CRUSE_REQUIRE(rows > 0 && G > 0 && Hg > 0, CRUSE_E_SHAPE, "gru_gate_grads: bad shape");
    long long nblk = (rows * G * Hg + 1023) / 1024;
    if (nblk > 4096) nblk = 4096;
    if (prec == CRUSE_PREC_BF16)
        hipLaunchKernelGGL(gru_gate_grads_kernel<__bf16>, dim3((int)nblk), dim3(256), 0, (hipStream_t)stream, dh,
                           (const __bf16*)coef, an, dgi, dgh, rows, G, Hg);
    else
        hipLaunchKernelGGL(gru_gate_grads_kernel<float>, dim3((int)nblk), dim3(256), 0, (hipStream_t)stream, dh,
                           (const float*)coef, an, dgi, dgh, rows, G, Hg);
    CRUSE_LAUNCH_CHECK("gru_gate_grads");
    return CRUSE_OK;
}

extern "C" int cruse_gru_gate_grads_bf16(const float* dh, const void* coef, const float* an, void* dgi, void* dgT,
                                         long long ldT, float* const* db_ih, float* const* db_hh,
                                         long long rows, int G, int Hg, void* stream) {
    CRUSE_REQUIRE(rows > 0 && G > 0 && G <= MAXG && Hg > 0 && Hg % 32 == 0, CRUSE_E_SHAPE,
                  "gru_gate_grads_bf16: bad shape rows=%lld G=%d Hg=%d", rows, G, Hg);
    CRUSE_REQUIRE(ldT % 64 == 0 && ldT >= rows && ldT < rows + 64, CRUSE_E_SHAPE,
                  "gru_gate_grads_bf16: ldT=%lld must be rows=%lld rounded up to a multiple of 64", ldT, rows);
    CRUSE_REQUIRE(dgi != nullptr || dgT != nullptr, CRUSE_E_SHAPE, "gru_gate_grads_bf16: dgi and dgT are both NULL");
    CRUSE_REQUIRE(((uintptr_t)dh % 16) == 0 && ((uintptr_t)an % 16) == 0 && ((uintptr_t)coef % 8) == 0 &&
                  ((uintptr_t)dgi % 8) == 0 && ((uintptr_t)dgT % 16) == 0, CRUSE_E_ALIGN,
                  "gru_gate_grads_bf16: unaligned buffers");
    GateBiasPtrs bp = {};
    for (int g = 0; g < G; ++g) { bp.ih[g] = db_ih ? db_ih[g] : nullptr; bp.hh[g] = db_hh ? db_hh[g] : nullptr; }
    hipStream_t s = (hipStream_t)stream;
    const unsigned nrt = (unsigned)cdivl(ldT / 64, GG_RT);
    if (Hg % 64 == 0)
        hipLaunchKernelGGL(gru_gate_grads_bf16_kernel<64>, dim3(nrt, G * (Hg / 64)), dim3(256), 0, s, dh,
                           (const __bf16*)coef, an, (__bf16*)dgi, (__bf16*)dgT, ldT, bp, rows, G, Hg);
    else
        hipLaunchKernelGGL(gru_gate_grads_bf16_kernel<32>, dim3(nrt, G * (Hg / 32)), dim3(256), 0, s, dh,
                           (const __bf16*)coef, an, (__bf16*)dgi, (__bf16*)dgT, ldT, bp, rows, G, Hg);
    CRUSE_LAUNCH_CHECK("gru_gate_grads_bf16");
    return CRUSE_OK;
}

