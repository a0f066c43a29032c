// Probe: how well does VALU work of one wave overlap with back-to-back f16 MFMAs of another wave on
// the same SIMD?  8 waves per workgroup (2 per SIMD): waves 0-3 run MFMAs, waves 4-7 run a fold-like
// VALU mix (fma, max, sub, v_exp, add per element).  Times each role alone and both together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int n_mfma, int n_valu) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long t0 = clock64();
    float r = 0;
    if (wave < 4) {
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int it = 0; it < n_mfma; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        }
        f32x16 s = c0 + c1 + c2 + c3;
        for (int i = 0; i < 16; ++i) r += s[i];
    } else {
        float v[16], m = -1e30f, s0 = 0, s1 = 0;
        for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
        for (int it = 0; it < n_valu; ++it) {       // 16 elements: fma, max | sub, exp, add  (= the LSE fold)
            float tm = m;
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i] = fmaf(v[i], 0.999f, 0.001f); tm = fmaxf(tm, v[i]); }
#pragma unroll
            for (int i = 0; i < 16; i += 2) { s0 += __builtin_amdgcn_exp2f(v[i] - tm); s1 += __builtin_amdgcn_exp2f(v[i + 1] - tm); }
            m = tm;
        }
        r = s0 + s1 + m;
    }
    unsigned long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

int main() {
    float *out; unsigned long long *cyc, h[8];
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 64);
    const int NM = 4000, NV = 1000;                 // 16000 MFMAs / 16000 folded elements per wave
    int cfg[3][2] = {{NM, 0}, {0, NV}, {NM, NV}};
    const char *nm[3] = {"MFMA waves alone", "VALU waves alone", "both"};
    for (int c = 0; c < 3; ++c) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, cyc, cfg[c][0], cfg[c][1]);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("%-18s MFMA wave %8llu cyc (%.1f / MFMA)   VALU wave %8llu cyc (%.1f / folded element)\n", nm[c], h[0],
               cfg[c][0] ? (double)h[0] / (4.0 * cfg[c][0]) : 0.0, h[4], cfg[c][1] ? (double)h[4] / (16.0 * cfg[c][1]) : 0.0);
    }
    return 0;
}
