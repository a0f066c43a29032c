// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/jlm_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define JLM_NEG_BIG (-3.0e38f)

#define JLM_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

// Kernel attributes (the dynamic-LDS size a kernel may be launched with) are per DEVICE, and the launchers are called from several
// host threads: every launch site's record of what it has already been granted is keyed by the current device and guarded by
// one lock.  -3: the size could not be set (include/jlm_hip.h).
#include <mutex>
#define JLM_MAX_DEVICES 16
struct JlmLdsGrant { int bytes[JLM_MAX_DEVICES]; };          // one zero-initialised static per (launch site, kernel)
static inline int jlm_grant_lds(JlmLdsGrant &g, const void *fn, int bytes) {
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= JLM_MAX_DEVICES) return -3;
    const std::lock_guard<std::mutex> lock(mu);
    if (bytes > g.bytes[dev]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -3;
        g.bytes[dev] = bytes;
    }
    return 0;
}

struct SegTable {
    int n;
    jlm_segment s[JLM_MAX_SEGMENTS];
};

// sigma(x) = 1/(exp(-x)+1), the reference's formula (decoder/model.py:12-13); exp overflow gives
// 1/inf = 0, as in numpy.  Hardware exp2 / rcp (1 ulp each): two transcendental issues instead of the
// ~40-instruction libm expansions, which made the gate epilogue as long as the gate GEMM's mainloop.
__device__ __forceinline__ float jlm_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * -1.4426950408889634f) + 1.0f);
}
// tanh(x) = 2 sigma(2x) - 1 (np.tanh in model.py:129-130): exact limits +-1, absolute error ~1e-7
__device__ __forceinline__ float jlm_tanh(float x) {
    return fmaf(2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * -2.8853900817779268f) + 1.0f), -1.0f);
}

// The same with the pre-activation's scale folded into the exponent's: sigma(x * ds) = 1 / (1 + 2^(x * ks)), ks = ds * -log2(e), and
// tanh(x * ds) with kt = 2 ks.  The LSTM-step kernels' ds is a power of two (2^-S: jlm_amd/model.py gate_descale), so ds * c is c with another
// exponent and x * (ds * c) rounds exactly as (x * ds) * c did: bit-identical results, one multiply less per gate (hipcc cannot know).
__device__ __forceinline__ float jlm_sigmoid_k(float x, float ks) { return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * ks) + 1.0f); }
__device__ __forceinline__ float jlm_tanh_k(float x, float kt) { return fmaf(2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * kt) + 1.0f), -1.0f); }

// online log-sum-exp pair merge: (m, s) <- (m, s) (+) (m2, s2); empty = (NEG_BIG, 0)
__device__ __forceinline__ void lse_merge(float &m, float &s, float m2, float s2) {
    float mm = fmaxf(m, m2);
    s = s * expf(m - mm) + s2 * expf(m2 - mm);
    m = mm;
}

// ---- split-f16 operands (include/jlm_hip.h "f16x3"): x * scale = hi + lo, both f16.
// The scaled value and hi are pinned in registers before they are used twice: left to itself the
// compiler folds the multiply into ONE of the two uses of the f32->f16 conversion (v_fma_mix*,
// single rounding from the exact product) and not the other, and hi and lo then disagree about
// what hi is whenever the f32 product sits on an f16 rounding tie -- a 2^-11 error in that element
// (tools/probes/split_debug2.py finds them).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void jlm_split2(float x0, float x1, float scale, f16x2 &hi, f16x2 &lo) {
#pragma clang fp contract(off)
    f32x2 v = {x0 * scale, x1 * scale};
    asm volatile("" : "+v"(v));
    f16x2 h = __builtin_convertvector(v, f16x2);              // v_cvt_pk_f16_f32, round to nearest even
    asm volatile("" : "+v"(h));
    const f32x2 r = v - __builtin_convertvector(h, f32x2);     // exact: h is within half an f16 ulp of v
    hi = h;
    lo = __builtin_convertvector(r, f16x2);
}

template <class V4>
__device__ __forceinline__ void jlm_split4(const f32x4 &x, float scale, V4 &hi, V4 &lo) {
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
        f16x2 h, l;
        jlm_split2(x[i], x[i + 1], scale, h, l);
        hi[i] = h[0]; hi[i + 1] = h[1];
        lo[i] = l[0]; lo[i + 1] = l[1];
    }
}

__device__ __forceinline__ void jlm_split8(const float *x, float scale, f16x8 &hi, f16x8 &lo) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        f16x2 h, l;
        jlm_split2(x[i], x[i + 1], scale, h, l);
        hi[i] = h[0]; hi[i + 1] = h[1];
        lo[i] = l[0]; lo[i + 1] = l[1];
    }
}

// Cycle accounting of the rows-stationary LSE kernels (tools/lse_profile.sh builds a -DJLM_PROFILE copy
// of the library; the shipped build has none of it): per wave, cycles in the prologue (T fragments), waiting
// at the k-step barrier (skew + DMA landing), in the fold, and in total.
#ifdef JLM_PROFILE
// one counter block per translation unit; JLM_PROF_READER(name) defines its extern "C" reader
static __device__ unsigned long long jlm_prof[8];
#define JLM_PROF_READER(name)                                                                                   \
    extern "C" int name(unsigned long long *out, int reset) {                                                   \
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(jlm_prof), sizeof(jlm_prof)) != hipSuccess) return -1;          \
        if (reset) {                                                                                            \
            unsigned long long z[8] = {0};                                                                      \
            if (hipMemcpyToSymbol(HIP_SYMBOL(jlm_prof), z, sizeof(z)) != hipSuccess) return -1;                 \
        }                                                                                                       \
        return 0;                                                                                               \
    }
#define JLM_PROF_DECL() unsigned long long p_t0 = clock64(), p_w0 = wall_clock64(), p_t1 = 0, p_t2 = 0, p_x = 0, p_bar = 0, p_fold = 0
#define JLM_PROF_MARK(v) v = clock64()
#define JLM_PROF_ADD(acc_, since) acc_ += clock64() - since
#define JLM_PROF_FLUSH()                                                                   \
    if ((threadIdx.x & 63) == 0) {                                                         \
        const unsigned long long now = clock64();                                          \
        atomicAdd(&jlm_prof[0], now - p_t0); atomicAdd(&jlm_prof[1], p_t1 - p_t0);         \
        atomicAdd(&jlm_prof[2], p_t2 - p_t1); atomicAdd(&jlm_prof[3], p_bar);              \
        atomicAdd(&jlm_prof[4], p_fold); atomicAdd(&jlm_prof[5], 1ull);                    \
        atomicAdd(&jlm_prof[6], wall_clock64() - p_w0);                                    \
    }
#else
#define JLM_PROF_READER(name)
#define JLM_PROF_DECL() (void)0
#define JLM_PROF_MARK(v) (void)0
#define JLM_PROF_ADD(acc_, since) (void)0
#define JLM_PROF_FLUSH() (void)0
#endif

