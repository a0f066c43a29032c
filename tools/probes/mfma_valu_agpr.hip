// Probe: does it matter for the OTHER wave's VALU stream whether a wave's MFMA accumulators live in the architectural VGPRs or in
// the accumulation registers?  8 waves per workgroup (2 per SIMD): waves 0-3 run back-to-back 32x32x16 f16 MFMAs (MODE 0: "+v"
// accumulators, MODE 1: "+a"), waves 4-7 the fold-like VALU mix of mfma_valu_overlap.hip.  Also the same-wave form (MODE 2 / 3):
// every wave runs MFMAs with 4 VALU ops of the fold behind each (acc "+v" / "+a").
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int n_mfma, int n_valu) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long t0 = clock64();
    float r = 0;
    if (MODE >= 2) {
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
        f32x16 c0 = {0}, c1 = {0};
        float v[16], m = -1e30f, s0 = 0, s1 = 0;
        for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
        for (int it = 0; it < n_mfma; ++it) {
            float tm = m;
#pragma unroll
            for (int q = 0; q < 8; ++q) {           // 8 MFMAs, behind each: 2 elements of the fold (fma, max, sub, exp, add) = 10 VALU
                if (MODE == 2) {
                    if (q & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
                    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
                } else {
                    if (q & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
                    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
                }
                if (n_valu) {
                    const int i = 2 * q;
                    v[i] = fmaf(v[i], 0.999f, 0.001f); v[i + 1] = fmaf(v[i + 1], 0.999f, 0.001f);
                    tm = fmaxf(tm, fmaxf(v[i], v[i + 1]));
                    s0 += __builtin_amdgcn_exp2f(v[i] - m); s1 += __builtin_amdgcn_exp2f(v[i + 1] - m);
                    asm volatile("" : "+v"(v[i]), "+v"(v[i + 1]), "+v"(s0), "+v"(s1), "+v"(tm));
                }
            }
            m = tm;
        }
        f32x16 s = c0 + c1;
        for (int i = 0; i < 16; ++i) r += s[i];
        r += s0 + s1 + m;
    } else if (wave < 4) {
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int it = 0; it < n_mfma; ++it) {
            if (MODE == 0)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n"
                             "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
            else
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n"
                             "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n"
                             : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
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

template <int MODE>
void run(const char *title, float *out, unsigned long long *cyc) {
    unsigned long long h[8];
    const int NM = 4000, NV = 1000;
    printf("%s\n", title);
    if (MODE < 2) {
        int cfg[3][2] = {{NM, 0}, {0, NV}, {NM, NV}};
        const char *nm[3] = {"MFMA waves alone", "VALU waves alone", "both"};
        for (int c = 0; c < 3; ++c) {
            hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, cyc, cfg[c][0], cfg[c][1]);
            hipDeviceSynchronize();
            hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
            printf("  %-18s MFMA wave %8llu cyc (%.1f / MFMA)   VALU wave %8llu cyc (%.1f / folded element)\n", nm[c], h[0],
                   cfg[c][0] ? (double)h[0] / (4.0 * cfg[c][0]) : 0.0, h[4], cfg[c][1] ? (double)h[4] / (16.0 * cfg[c][1]) : 0.0);
        }
    } else {
        for (int nv = 0; nv < 2; ++nv) {
            hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, cyc, 2000, nv);
            hipDeviceSynchronize();
            hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
            printf("  8 waves, MFMAs %s: wave 0 %8llu cyc, wave 4 %8llu cyc (%.1f / MFMA and wave; 2 waves per SIMD)\n",
                   nv ? "+ 10 VALU of the fold behind each" : "alone", h[0], h[4], (double)h[0] / (8.0 * 2000));
        }
    }
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 64);
    run<0>("accumulators in architectural VGPRs, VALU on the other wave", out, cyc);
    run<1>("accumulators in AGPRs, VALU on the other wave", out, cyc);
    run<2>("same-wave interleave, accumulators in architectural VGPRs", out, cyc);
    run<3>("same-wave interleave, accumulators in AGPRs", out, cyc);
    return 0;
}
