// Shared helpers for the gfx950 kernels of libirm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define IRM_OK 0
#define IRM_EINVAL (-1)
#define IRM_ELAUNCH (-2)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// First statement of every kernel.  Product build: nothing.  -DIRM_ACQUIRE_ENTRY (diagnostic variant, tools/build_variant.sh):
// an agent-scope acquire (buffer_inv sc1: this CU's vector L1) on every wave before its first load - the round-3
// experiment on the two-stream stale read.
#ifdef IRM_ACQUIRE_ENTRY
#define IRM_KERNEL_ENTRY() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define IRM_KERNEL_ENTRY() do { } while (0)
#endif

// activation codes shared by the GEMM / conv epilogues
#define IRM_ACT_NONE 0
#define IRM_ACT_RELU 1
#define IRM_ACT_GELU 2   // exact erf GELU (torch F.gelu default)
#define IRM_ACT_SILU 3

// LayerNorm prologue modes (restormer.py:25-70)
#define IRM_LN_NONE 0
#define IRM_LN_WITHBIAS 1
#define IRM_LN_BIASFREE 2

// Experiment switches (environment variables that select kernel variants, timing-only modes that skip work) exist only
// in -DIRM_PROBES builds (tools/build_variant.sh); in the product build they are compile-time constants: no environment
// variable can change what a kernel computes or skip part of it.
#ifdef IRM_PROBES
#include <stdlib.h>
static inline int irm_probe_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline bool irm_probe_set(const char* name) { return getenv(name) != nullptr; }
#define IRM_DBG(v, bit) (((v) & (bit)) != 0)
#else
static inline int irm_probe_int(const char*, int dflt) { return dflt; }
static inline bool irm_probe_set(const char*) { return false; }
#define IRM_DBG(v, bit) false
#endif

// Kernels that use more than 64 KiB of LDS: raise the limit once per (kernel instantiation, device).
#define IRM_ALLOW_BIG_LDS(kernel_ptr)                                                                              \
    do {                                                                                                           \
        static unsigned char irm_done_[64] = {0};                                                                  \
        int irm_dev_ = 0;                                                                                          \
        if (hipGetDevice(&irm_dev_) != hipSuccess || irm_dev_ < 0 || irm_dev_ >= 64) return IRM_ELAUNCH;           \
        if (!irm_done_[irm_dev_]) {                                                                                \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_ptr),                                     \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)         \
                return IRM_ELAUNCH;                                                                                \
            irm_done_[irm_dev_] = 1;                                                                               \
        }                                                                                                          \
    } while (0)

static inline int irm_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? IRM_OK : IRM_ELAUNCH;
}

__device__ __forceinline__ float irm_gelu(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ float irm_act(float v, int act) {
    if (act == IRM_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == IRM_ACT_GELU) return irm_gelu(v);
    if (act == IRM_ACT_SILU) return v / (1.0f + __expf(-v));
    return v;
}

// D(16x16) += A(16x4) * B(4x16), exact f32.  Lane l supplies A[l&15][l>>4] and
// B[l>>4][l&15]; it receives D[(l>>4)*4 + r][l&15] in element r of the result.
__device__ __forceinline__ f32x4 irm_mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// 4 consecutive floats at row[n..n+3]; VEC: one 16-byte access (row + n 16-byte aligned, n + 3 < N
// whenever n < N), else guarded scalars (elements at or beyond N read as 0 / are not written).
template <bool VEC>
__device__ __forceinline__ float4 irm_ld4(const float* row, int n, int N) {
    if (VEC) return *reinterpret_cast<const float4*>(row + n);
    float4 v;
    v.x = n < N ? row[n] : 0.0f;
    v.y = n + 1 < N ? row[n + 1] : 0.0f;
    v.z = n + 2 < N ? row[n + 2] : 0.0f;
    v.w = n + 3 < N ? row[n + 3] : 0.0f;
    return v;
}
template <bool VEC>
__device__ __forceinline__ void irm_st4(float* row, int n, int N, float4 v) {
    if (VEC) { *reinterpret_cast<float4*>(row + n) = v; return; }
    if (n < N) row[n] = v.x;
    if (n + 1 < N) row[n + 1] = v.y;
    if (n + 2 < N) row[n + 2] = v.z;
    if (n + 3 < N) row[n + 3] = v.w;
}
// fp16 hi/lo splits behind a FIXED scale (2^-4: gated activations, v, un-normalised GEMM inputs): the scaled value is
// saturated at +-65000 before the split, so an out-of-range activation (|x| > ~1e6) gives a clamped, finite operand instead
// of fp16 infinities and a NaN tile (one v_med3_f32; the scaled splits behind a LayerNorm or a pack-time bound cannot
// overflow and do not clamp).
__device__ __forceinline__ float irm_sat_h(float x) { return __builtin_amdgcn_fmed3f(x, -65000.0f, 65000.0f); }

// fp16 hi/lo split of fp32 values that are ALREADY rounded (register operands, opaque to the compiler): hi = rn16(x),
// lo = rn16(x - hi).  The difference is exact in fp32, so v_fma_mix{lo,hi}_f16 (f16 source widened, one rounding of the
// result) gives exactly the two-step value: cvt_pk + 2 mix instructions per PAIR instead of 2 x (cvt, cvt back, sub, cvt);
// hipcc does not form it from the source expression, and left alone it may fuse a preceding multiply into the lo part
// only (hi from the rounded product, lo from the exact one: 2^-11 outliers on double-rounding ties).
__device__ __forceinline__ void irm_split2(float a, float b, unsigned& hi, unsigned& lo) {
    unsigned h, l;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(a));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(b));
    hi = h;
    lo = l;
}
typedef unsigned irm_u2 __attribute__((ext_vector_type(2)));
typedef unsigned irm_u4 __attribute__((ext_vector_type(4)));
template <typename H8>
__device__ __forceinline__ void irm_split8(const float (&x)[8], H8& hi, H8& lo) {
    static_assert(sizeof(H8) == 16, "8 halves");
    irm_u4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsigned hh, ll; irm_split2(x[2 * e], x[2 * e + 1], hh, ll); h[e] = hh; l[e] = ll; }
    hi = __builtin_bit_cast(H8, h);
    lo = __builtin_bit_cast(H8, l);
}
template <typename H4>
__device__ __forceinline__ void irm_split4(const float (&x)[4], H4& hi, H4& lo) {
    static_assert(sizeof(H4) == 8, "4 halves");
    irm_u2 h, l;
#pragma unroll
    for (int e = 0; e < 2; ++e) { unsigned hh, ll; irm_split2(x[2 * e], x[2 * e + 1], hh, ll); h[e] = hh; l[e] = ll; }
    hi = __builtin_bit_cast(H4, h);
    lo = __builtin_bit_cast(H4, l);
}

static inline bool irm_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
