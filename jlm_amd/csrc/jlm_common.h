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

struct SegTable {
    int n;
    jlm_segment s[JLM_MAX_SEGMENTS];
};

// sigma(x) = 1/(exp(-x)+1), the reference's formula (decoder/model.py:12-13);
// exp overflow gives 1/inf = 0, as in numpy.
__device__ __forceinline__ float jlm_sigmoid(float x) { return 1.0f / (expf(-x) + 1.0f); }

// online log-sum-exp pair merge: (m, s) <- (m, s) (+) (m2, s2); empty = (NEG_BIG, 0)
__device__ __forceinline__ void lse_merge(float &m, float &s, float m2, float s2) {
    float mm = fmaxf(m, m2);
    s = s * expf(m - mm) + s2 * expf(m2 - mm);
    m = mm;
}
