// Shared pieces of the persistent GRU recurrence kernels (gru.hip, gru_tf.hip): launch arguments, hand-off helpers,
// team placement, gate nonlinearities.  See gru.hip for the design notes.
#pragma once
#include "common.h"
#include <stdlib.h>

namespace cruse_gru {

constexpr int U = 32;                 // hidden units per workgroup
constexpr int MAXG = 8;
constexpr int CH = 10;                // 16-byte granule pairs in flight per thread and sweep batch
constexpr unsigned SPIN_LIMIT = 1u << 21;
// K-reduction buffer of the forward kernels: [4 K groups][6 tiles][RED_TS floats]; the f32x4 accumulator vector of lane L sits at
// float offset (L + (L >> 4)) * 4 of its tile.  A gate thread (clip bl, unit u) reads element (u & 3) of lane ((u & 15) >> 2) * 16
// + bl of tile gate*2 + (u >> 4): with the plain [64 lanes][4] image all 32 lanes of a half-wave fell on FOUR banks (tile and
// 16-lane strides are multiples of 32 banks): an 8-way conflict on each of the 12 reads of every thread, ~700 of the ~3200
// cycles of a step (s_memtime stamps, gru_dbg = 32).  The extra 16 bytes per 16 lanes and the 272-float tile stride (68 vector
// slots = 4 mod 8) spread them over all 32.
constexpr int RED_TS = 272;
__device__ __forceinline__ int red_vec(int lane) { return (lane + (lane >> 4)) * 4; }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct GruPtrs { const float* w_hh[MAXG]; const float* b_hh[MAXG]; };

struct GruArgs {
    // forward
    const float* gi; float* h; void* coef; float* an; float* z;
    int gi_bf16;                      // gi rows are bf16 (written by cruse_gemm_bf16_nt_obf16): half the bytes; widened on load
    // backward
    const float* dout; const void* coefs; const float* zs; float* dh;
    const float* ans; void* dgi;      // optional (reduce-scatter kernel): a_n rows in, dgi = dh * (c_r, c_z, a_n) rows out
    int dg_slabs;                     // 3: dgi rows [G][3][Hg]; 4: [G][4][Hg] with slab 3 = dh * c_n (the n gate of dgh): one
                                      // row-major tensor then feeds dX and both TN weight-gradient GEMMs (gemm_bf16_tn)
    GruPtrs p;
    int B, T, G, Hg, Bg, nchains, P, bg_off;
    unsigned long long* xid;          // XCD-id handshake granules [chain][64]
    unsigned long long* xg;           // granule panels [chain][parity][Bg][Hg]
    unsigned xg_bytes;
    unsigned* status;
    int xcd_rot;                      // chain c of a launch runs on PHYSICAL XCD (c + xcd_rot) % 8 (concurrent launches: disjoint XCDs)
    unsigned* tickets;                // [8] per-XCD workgroup counters of this launch (zeroed with the panels)
    int prio;                         // s_setprio level of the compute waves (option gru_prio): the recurrence's instructions issue ahead of
                                      // co-resident side-stream waves on the same SIMD
    int poll_delay;                   // tag-free kernels: s_sleep(1) periods between a step's publish and its first poll
    int xsweep;                       // profiling: extra sweep / publish repetitions per step of the forward lean kernel (gru_xsweep)
    int poll_stagger;                 // backward tag-free kernel: > 0 = two polls in flight, this many s_sleep(1) periods apart
    int dbg;                          // profiling only (CRUSE_GRU_DBG): 1 = do not wait for tags, 2 = also skip MFMA
    // sub-sequences (cruse_gru_seq_*_ex): T steps of tensors whose clips are TS frames apart (TS >= T; the pointers are
    // already advanced to the first frame of the run)
    int TS;
    const float* h0; long long h0_bs; // forward: initial state [B][G*Hg] ("cat" layout), clips h0_bs floats apart; NULL = 0
    int carry;                        // backward: the first iteration takes dh of frame T-1 from the dh buffer (a later
                                      // run of the same sequence wrote it) instead of forming it from dout
};

// LDS panel storage per precision: f32 keeps floats; bf16 / bf16x3 keep 1 / 2 planes of bf16 converted
// once when the granules arrive, so an MFMA B fragment is one ds_read_b128 per plane.
template <int PREC> struct Panel {
    typedef __bf16 elem;
    static constexpr int NPL = (PREC == CRUSE_PREC_BF16X3) ? 2 : 1;
    static constexpr int PAD = 8;
};
template <> struct Panel<CRUSE_PREC_F32> {
    typedef float elem;
    static constexpr int NPL = 1;
    static constexpr int PAD = 4;
};

template <int PREC>
__device__ __forceinline__ void panel_put2(typename Panel<PREC>::elem* base, int plane, int off, float v0, float v1) {
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
__device__ __forceinline__ Frag<PREC> panel_get(const typename Panel<PREC>::elem* base, int plane, int off) {
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

// Hand-off granule formats.  f32 / bf16x3: 8-byte {epoch u32, value f32}, a 16-byte load carries 2 values.
// bf16: 8-byte {epoch u32, 2 x bf16}, a 16-byte load carries 4 values (the recurrent MFMA rounds the
// panel to bf16 anyway), which halves the sweep bytes.  The backward coefficient tensor is stored in the
// matching type (f32 or bf16).
template <int PREC> struct Gran {
    static constexpr int VPL = 2;            // values per 16-byte load
    static constexpr int CHN = CH;           // 16-byte loads in flight per thread and sweep batch
    typedef float coef_t;
    typedef float2 creg_t;                   // one gate's coefficients of one load, as held in registers
};
template <> struct Gran<CRUSE_PREC_BF16> {
    static constexpr int VPL = 4;
    static constexpr int CHN = CH / 2;
    typedef __bf16 coef_t;
    typedef u32x2 creg_t;                    // 4 bf16, unpacked at use
};

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

// lane exchanges inside a 16-lane row as DPP moves (one VALU op) instead of ds_bpermute round trips
__device__ __forceinline__ unsigned dpp_xor1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ unsigned dpp_xor2(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ unsigned dpp_ror8(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); }  // row_ror:8 == lane ^ 8
__device__ __forceinline__ unsigned swz_xor16(unsigned v) { return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x401F); }        // ds_swizzle BITMASK_PERM: and 0x1f, xor 0x10
__device__ __forceinline__ float bf16lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2 h;
    h[0] = (__bf16)a; h[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, h);
}

// per-thread sweep bookkeeping for load slots i = 0..CH-1 (load index e = tid + 256*i)
template <int PREC> struct SweepIdx {
    int loff[Gran<PREC>::CHN];        // LDS element offset bl*ld + v
    int coff[Gran<PREC>::CHN];        // coefficient element offset bl*cf_row_stride + v (backward)
    unsigned valid;
};
template <int PREC>
__device__ __forceinline__ void make_idx(SweepIdx<PREC>& si, int nload, int Hg, int ld, long long cf_row_stride, int i0,
                                         int tid) {
    constexpr int VPL = Gran<PREC>::VPL, CHN = Gran<PREC>::CHN;
    const int per = Hg / VPL;
    si.valid = 0;
#pragma unroll
    for (int j = 0; j < CHN; ++j) {
        const int e = tid + 256 * (i0 + j);
        si.loff[j] = 0; si.coff[j] = 0;
        if (e < nload) {
            const int bl = e / per, v = VPL * (e - bl * per);
            si.loff[j] = bl * ld + v;
            si.coff[j] = (int)(bl * cf_row_stride) + v;
            si.valid |= 1u << j;
        }
    }
}

// coefficient rows of the backward sweep, prefetched one step ahead (plain loads): 3 gates x VPL values
template <int PREC> struct CoefRegs { typename Gran<PREC>::creg_t c[Gran<PREC>::CHN][3]; };

template <int PREC>
__device__ __forceinline__ void load_coefs(CoefRegs<PREC>& cr, const typename Gran<PREC>::coef_t* cf, int Hg,
                                           const SweepIdx<PREC>& si) {
#pragma unroll
    for (int j = 0; j < Gran<PREC>::CHN; ++j) {
        if (si.valid & (1u << j)) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
                cr.c[j][q] = *reinterpret_cast<const typename Gran<PREC>::creg_t*>(cf + si.coff[j] + q * Hg);
        }
    }
}

// Sweep a team panel of granules into LDS until every tag == epoch.
// BWD == false: lds[bl*ld + v] = value                       (h_{t-1} panel, ld = Hg+PAD)
// BWD == true : lds[bl*ld + g*Hg + v] = value * coef[g][v]   (dh_t * c_t, ld = 3Hg+PAD); the coefficient
//               registers of the first batch arrive preloaded in `cr0`.
template <int PREC, bool BWD>
__device__ __forceinline__ bool sweep_panel(typename Panel<PREC>::elem* lds, int plane, int ld, __amdgpu_buffer_rsrc_t rs,
                                            unsigned base_bytes, int nload, int Hg, unsigned epoch,
                                            const SweepIdx<PREC>& si0, const CoefRegs<PREC>* cr0,
                                            const typename Gran<PREC>::coef_t* cf, long long cf_row_stride,
                                            unsigned* status, int tid, bool nowait) {
    constexpr int VPL = Gran<PREC>::VPL, CHN = Gran<PREC>::CHN;
    const int ni = (nload + 255) >> 8;
    bool timed_out = false;
    for (int i0 = 0; i0 < ni; i0 += CHN) {
        SweepIdx<PREC> si;
        CoefRegs<PREC> cr;
        if (i0 == 0) {
            si = si0;
            if (BWD) cr = *cr0;
        } else {
            make_idx<PREC>(si, nload, Hg, ld, cf_row_stride, i0, tid);
            if (BWD) load_coefs<PREC>(cr, cf, Hg, si);
        }
        unsigned pend = si.valid;
        unsigned spins = 0;
        for (;;) {
            u32x4 g[CHN];
#pragma unroll
            for (int j = 0; j < CHN; ++j)
                if (pend & (1u << j))
                    g[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, base_bytes + (unsigned)(tid + 256 * (i0 + j)) * 16u, 0, 16);
#pragma unroll
            for (int j = 0; j < CHN; ++j) {
                if ((pend & (1u << j)) && (nowait || (g[j].x == epoch && g[j].z == epoch))) {
                    float val[VPL];
                    if constexpr (PREC == CRUSE_PREC_BF16) {
                        val[0] = bf16lo(g[j].y); val[1] = bf16hi(g[j].y); val[2] = bf16lo(g[j].w); val[3] = bf16hi(g[j].w);
                    } else {
                        val[0] = __uint_as_float(g[j].y); val[1] = __uint_as_float(g[j].w);
                    }
                    if (BWD) {
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            if constexpr (PREC == CRUSE_PREC_BF16) {
                                const u32x2 cw = cr.c[j][q];
                                u32x2 w;
                                w.x = pack2(val[0] * bf16lo(cw.x), val[1] * bf16hi(cw.x));
                                w.y = pack2(val[2] * bf16lo(cw.y), val[3] * bf16hi(cw.y));
                                *reinterpret_cast<u32x2*>(lds + si.loff[j] + q * Hg) = w;
                            } else {
                                panel_put2<PREC>(lds, plane, si.loff[j] + q * Hg, val[0] * cr.c[j][q].x, val[1] * cr.c[j][q].y);
                            }
                        }
                    } else {
                        if constexpr (PREC == CRUSE_PREC_BF16) {
                            u32x2 w;
                            w.x = g[j].y; w.y = g[j].w;           // already bf16 pairs: no conversion at all
                            *reinterpret_cast<u32x2*>(lds + si.loff[j]) = w;
                        } else {
                            panel_put2<PREC>(lds, plane, si.loff[j], val[0], val[1]);
                        }
                    }
                    pend &= ~(1u << j);
                }
            }
            if (__syncthreads_and(pend == 0)) break;
            if (++spins >= SPIN_LIMIT) {     // give up: flag it and let the caller run the remaining steps unsynchronised
                if (tid == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                timed_out = true;
                nowait = true;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return timed_out;
}

// One-time team handshake: every workgroup publishes the id of the XCD it runs on (write-through), reads
// its P team mates' ids, and all reach the same verdict.  If the whole team shares one XCD its L2 is a
// common coherence point: the step payload can then be published with PLAIN stores (the line stays in that
// L2, consumers read it with L1-bypassing sc1 loads) instead of write-through stores -- ~0.5 us per step
// faster.  Any other placement keeps the write-through form; correctness never depends on placement.
__device__ __forceinline__ bool team_shares_xcd(unsigned long long* slots, int P, int part, unsigned* status, int tid) {
    __shared__ int s_same;
    const unsigned my = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;      // HW_REG_XCC_ID
    constexpr unsigned MAGIC = 0xC0DE0001u;
    if (tid == 0)
        __hip_atomic_store(slots + part, ((unsigned long long)my << 32) | MAGIC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
        bool ok = false, same = true;
        for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
            unsigned long long v = ((unsigned long long)my << 32) | MAGIC;
            if (tid < P) v = __hip_atomic_load(slots + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)v == MAGIC)) { ok = true; same = __all((unsigned)(v >> 32) == my); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (tid == 0) {
            s_same = (ok && same) ? 1 : 0;
            if (!ok) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    return s_same != 0;
}

// Workgroup -> (chain, part) by the PHYSICAL XCD the workgroup landed on.  A dispatch deals its workgroups round-robin
// over the 8 XCDs, but from a base that differs from dispatch to dispatch (census, tools/cu_mask_probe.py: bases 6, 7, 0
// for consecutive launches), so block ids say which workgroups share an XCD, not which XCD that is.  Each workgroup
// therefore reads its XCC id and draws a ticket from that XCD's counter: ticket t on XCD x is part t % P of chain
// (t / P) * 8 + ((x - xcd_rot) & 7).  A team always shares one XCD (plain-store hand-off, team_shares_xcd), and two
// concurrent launches with xcd_rot 0 and 4 and <= 4 chains each use disjoint XCDs whatever their dispatch bases.
// If an XCD ever received more workgroups than it has places (not observed), the surplus ones take the places left
// on the other XCDs: every place is always filled, placement is speed only.
__device__ __forceinline__ bool claim_chain(const GruArgs& a, int P, int& chain, int& part) {
    __shared__ int s_claim[2];
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;      // HW_REG_XCC_ID
        const unsigned places = (unsigned)(gridDim.x >> 3);         // per XCD: chain groups x P
        unsigned x = xcc, t = 0;
        for (int k = 0; k < 8; ++k) {
            x = (xcc + k) & 7u;
            t = __hip_atomic_fetch_add(a.tickets + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < places) break;
        }
        const int c = (int)(t / (unsigned)P) * 8 + (int)((x - (unsigned)a.xcd_rot) & 7u);
        s_claim[0] = (t < places && c < a.nchains) ? c : -1;
        s_claim[1] = (int)(t % (unsigned)P);
    }
    __syncthreads();
    chain = s_claim[0]; part = s_claim[1];
    return chain >= 0;
}

// publish the own unit pair (u, u+1) of clip bl; pair_index = bl*Hg/2 + (u0+u)/2
template <int PREC>
__device__ __forceinline__ void publish_pair(__amdgpu_buffer_rsrc_t rs, unsigned panel_base, unsigned pair_index,
                                             unsigned epoch, float a, float b, bool plain = false) {
    if constexpr (PREC == CRUSE_PREC_BF16) {
        const u32x2 w = {epoch, pack2(a, b)};
        if (plain) __builtin_amdgcn_raw_buffer_store_b64(w, rs, panel_base + pair_index * 8u, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b64(w, rs, panel_base + pair_index * 8u, 0, 16);    // aux 16 = sc1
    } else {
        const u32x4 w = {epoch, __float_as_uint(a), epoch, __float_as_uint(b)};
        if (plain) __builtin_amdgcn_raw_buffer_store_b128(w, rs, panel_base + pair_index * 16u, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(w, rs, panel_base + pair_index * 16u, 0, 16);
    }
}

// ---------------------------------------------------------------------------------
// forward.  Memory-queue discipline (vmcnt is in-order and counts stores): when a step's sweep waits for
// its granule loads, the only older request still in flight is the previous step's publish store -- the gi
// rows are loaded one step AHEAD and a step's saves are issued after the NEXT step's sweep has returned.
// ---------------------------------------------------------------------------------
// gate nonlinearities on the serial path: v_exp_f32 + v_rcp_f32 forms (abs. error ~1e-7), not the
// libm expf/tanhf sequences (they cost ~1 us per step of pure VALU latency)
__device__ __forceinline__ float fast_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __expf(-2.0f * fabsf(x));          // in (0, 1]: no overflow
    const float t = (1.0f - e) * __frcp_rn(1.0f + e);
    return copysignf(t, x);
}

// bf16-mode forms: bare v_exp_f32 / v_rcp_f32 (1 ulp) -- __frcp_rn expands to the ~10-instruction IEEE division
// sequence, three of them back to back on the serial path of every step
__device__ __forceinline__ float lean_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float lean_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(-2.8853900817779268f * fabsf(x));      // in (0, 1]: no overflow
    return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
}

// Tag-free hand-off (gru_tf.hip): the epoch bit of a published bf16 rides in its bit 14 -- the top exponent bit, 0 for |v| < 2.
constexpr unsigned TAGM = 0x40004000u;            // bit 14 of both bf16 halves of a dword
__device__ __forceinline__ unsigned tag_bit(unsigned epoch) { return ((epoch + 1u) >> 1) & 1u; }
// (x & ~TAGM) | tag  in one VOP3 (v_and_or_b32)
__device__ __forceinline__ unsigned with_tag(unsigned x, unsigned tagm) { return (x & ~TAGM) | tagm; }

template <typename Kern>
inline int launch_one(Kern k, const GruArgs& a, int grid, size_t lds, hipStream_t s, const char* name, int threads = 256) {
    int rc0 = cruse_ensure_dyn_lds(reinterpret_cast<const void*>(k), lds, name);
    if (rc0) return rc0;
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, s, a);
    CRUSE_LAUNCH_CHECK(name);
    return CRUSE_OK;
}

}  // namespace cruse_gru
