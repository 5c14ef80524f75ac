// Shared helpers for the CRUSE gfx950 kernels (wave = 64 lanes, hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/cruse_hip.h"

// the BatchNorm(+ReLU) whose input gradient a data-gradient convolution produces (cruse_conv_*_bnbwd): its pre-BN tensor and
// per-channel statistics / affine parameters
struct CruseBnBwd { const float* y; const float* mean; const float* rstd; const float* gamma; const float* beta; int relu; };
// the BatchNorm(+ReLU) a forward convolution applies to its INPUT while staging it (cruse_conv_*_bnin): batch sums of the layer below
// ([nrep][2*Cin] f64), its affine parameters, where to publish mean / rstd / running statistics (nullable), an optional tensor added
// after the ReLU (the decoder's skip) and an optional bf16 copy of the transformed rows
struct CruseBnIn { const double* sums; int nrep; long long count; float eps, momentum; const float* gamma; const float* beta;
                   float* mean_o; float* rstd_o; float* rmean; float* rvar; const float* add; void* copy_bf16; };

// the BatchNorm(+ReLU) BACKWARD a data-gradient convolution applies to its INPUT while staging it (cruse_conv_*_bnbwd_in): x is the gradient
// wrt that BatchNorm's output (bf16), y its pre-BN tensor, sums its backward batch sums ([nrep][2*Cin] f64: sum g, sum g*xhat); copy_bf16
// receives dy (the tensor the separate cruse_bn_act_bwd_apply pass would store); dgamma / dbeta / dbias (nullable) are ADDED to once
struct CruseBnBwdIn { const float* y; const double* sums; int nrep; long long count; const float* mean; const float* rstd; const float* gamma;
                      const float* beta; int relu, training; void* copy_bf16; float* dgamma; float* dbeta; float* dbias; };

extern "C" void cruse_set_error(const char* fmt, ...);

#define CRUSE_REQUIRE(cond, code, ...)                                   \
    do {                                                                 \
        if (!(cond)) {                                                   \
            cruse_set_error(__VA_ARGS__);                                \
            return (code);                                               \
        }                                                                \
    } while (0)

#define CRUSE_LAUNCH_CHECK(name)                                         \
    do {                                                                 \
        hipError_t e__ = hipGetLastError();                              \
        if (e__ != hipSuccess) {                                         \
            cruse_set_error("%s: HIP launch failed: %s", name, hipGetErrorString(e__)); \
            return CRUSE_E_HIP;                                          \
        }                                                                \
    } while (0)

#define CRUSE_HIP(call, name)                                            \
    do {                                                                 \
        hipError_t e__ = (call);                                         \
        if (e__ != hipSuccess) {                                         \
            cruse_set_error("%s: %s", name, hipGetErrorString(e__));     \
            return CRUSE_E_HIP;                                          \
        }                                                                \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and size class, so that
// repeated launches (and launches recorded during hipGraph capture) issue no attribute calls.
int cruse_ensure_dyn_lds(const void* fn, size_t bytes, const char* name);

// Zero `bytes` (multiple of 4) of device memory with a KERNEL node on `stream` (not hipMemsetAsync:
// inside a captured hipGraph, memset nodes were observed to race with the neighbouring kernel nodes).
int cruse_zero_async(void* p, size_t bytes, hipStream_t stream, const char* name);

// library option `name` (cruse_set_option), or dflt when the host has not set it; the library reads no environment variables
int cruse_opt(const char* name, int dflt);

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivl(long long a, long long b) { return (a + b - 1) / b; }

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- wave / block reductions -------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- bf16 split helpers --------------------------------------------------------
// hi = RNE bf16(x); lo = RNE bf16(x - hi): x ~= hi + lo to ~2^-17 relative.
__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// accurate variants used where parity with the CPU oracle matters
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

// precision modes shared by the MFMA kernels
//   CRUSE_PREC_F32   : v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulate)
//   CRUSE_PREC_BF16X3: operands split hi+lo bf16, 3 bf16 MFMAs (hi*hi + hi*lo + lo*hi)
//   CRUSE_PREC_BF16  : operands rounded to bf16, 1 bf16 MFMA, f32 accumulate
//   CRUSE_PREC_F16   : operands rounded to f16, 1 f16 MFMA (v_mfma_f32_16x16x32_f16), f32 accumulate
//
// Operand fragment of a 16x16x32 tile product: lane l holds, for row/column (l & 15), the
// eight K-elements k = (l >> 4) * 8 + 0..7.  For bf16 this is exactly the
// v_mfma_f32_16x16x32_bf16 A/B layout; for f32 the 8 elements feed 8 successive
// v_mfma_f32_16x16x4_f32 (element q of every lane = the k-slice {q, 8+q, 16+q, 24+q}),
// which sums the same 32 products.
template <int PREC> struct Frag;
template <> struct Frag<CRUSE_PREC_F32> {
    float v[8];
    __device__ __forceinline__ void set(const float (&x)[8]) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = x[q];
    }
};
template <> struct Frag<CRUSE_PREC_BF16> {
    bf16x8 h;
    __device__ __forceinline__ void set(const float (&x)[8]) {
#pragma unroll
        for (int q = 0; q < 8; ++q) h[q] = (__bf16)x[q];
    }
};
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
template <> struct Frag<CRUSE_PREC_F16> {          // v_mfma_f32_16x16x32_f16: the same fragment layout as bf16
    f16x8_t h;
    __device__ __forceinline__ void set(const float (&x)[8]) {
#pragma unroll
        for (int q = 0; q < 8; ++q) h[q] = (_Float16)x[q];
    }
};
template <> struct Frag<CRUSE_PREC_BF16X3> {
    bf16x8 h, l;
    __device__ __forceinline__ void set(const float (&x)[8]) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { __bf16 a, b; split_bf16(x[q], a, b); h[q] = a; l[q] = b; }
    }
};

__device__ __forceinline__ f32x4 mma(const Frag<CRUSE_PREC_F32>& a, const Frag<CRUSE_PREC_F32>& b, f32x4 c) {
#pragma unroll
    for (int q = 0; q < 8; ++q) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[q], b.v[q], c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ f32x4 mma(const Frag<CRUSE_PREC_BF16>& a, const Frag<CRUSE_PREC_BF16>& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(const Frag<CRUSE_PREC_F16>& a, const Frag<CRUSE_PREC_F16>& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, b.h, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(const Frag<CRUSE_PREC_BF16X3>& a, const Frag<CRUSE_PREC_BF16X3>& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, c, 0, 0, 0);
    return c;
}

// mma with the operand ROLES swapped (the transposed product: rows of the result follow `b`): the same three split-bf16 terms
// in the same order as mma(a, b), so a kernel that swaps roles for some launches produces the bits of the one that does not
template <int PREC>
__device__ __forceinline__ f32x4 mma_t(const Frag<PREC>& b, const Frag<PREC>& a, f32x4 c) { return mma(b, a, c); }
template <>
__device__ __forceinline__ f32x4 mma_t<CRUSE_PREC_BF16X3>(const Frag<CRUSE_PREC_BF16X3>& b, const Frag<CRUSE_PREC_BF16X3>& a, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.h, a.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.l, a.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.h, a.h, c, 0, 0, 0);
    return c;
}
