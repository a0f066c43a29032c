// Probe: how many cycles of VALU issue does one matrix instruction take away from its SIMD?  8 waves per workgroup (2 per SIMD), every
// wave runs NV VALU instructions of the fold (fma, max, sub, exp, add mix) behind each matrix instruction, for several instruction
// forms; VALU-bound on purpose (NV = 10).  blocked = (cycles per slot and SIMD) - (the same VALU stream without matrix instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE 0 no matrix instruction, 1 f16 32x32x16 (2 accumulators in turn), 2 i8 32x32x32, 3 f16 16x16x32 (two per slot), 4 f16 32x32x16 with
// C = 0 (no accumulator read), 5 f16 32x32x16 with A and B operands in AGPRs, 6 alternating f16 / i8 as the vocabulary kernel does,
// 7 f16 32x32x16 with the accumulators in AGPRs
template <int MODE, int NV>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int n) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    i32x4 ia = {1, 2, 3, (int)threadIdx.x}, ib = {4, 5, 6, 7};
    f32x16 c0 = {0}, c1 = {0};
    i32x16 d0 = {0}, d1 = {0};
    f32x4 e0 = {0}, e1 = {0};
    float v[16], m = -1e30f, s0 = 0, s1 = 0;
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
    unsigned long long t0 = clock64();
    for (int it = 0; it < n; ++it) {
        float tm = m;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (MODE == 1) {
                if (q & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            } else if (MODE == 2) {
                if (q & 1) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(d1) : "v"(ia), "v"(ib));
                else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(d0) : "v"(ia), "v"(ib));
            } else if (MODE == 3) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_f16 %1, %2, %3, %1" : "+v"(e0), "+v"(e1) : "v"(a), "v"(b));
            } else if (MODE == 4) {
                if (q & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(c1) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(c0) : "v"(a), "v"(b));
            } else if (MODE == 5) {
                if (q & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "a"(a), "a"(b));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "a"(a), "a"(b));
            } else if (MODE == 6) {
                if (q & 1) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(d0) : "v"(ia), "v"(ib));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            } else if (MODE == 7) {
                if (q & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
            }
            // NV VALU instructions: per pair of elements fma, fma, max3-like (2 max), sub, sub, exp, exp, add, add = 10
            const int i = 2 * q;
            if (NV >= 2) { v[i] = fmaf(v[i], 0.999f, 0.001f); v[i + 1] = fmaf(v[i + 1], 0.999f, 0.001f); }
            if (NV >= 4) { tm = fmaxf(tm, v[i]); tm = fmaxf(tm, v[i + 1]); }
            if (NV >= 10) { s0 += __builtin_amdgcn_exp2f(v[i] - m); s1 += __builtin_amdgcn_exp2f(v[i + 1] - m); }
            if (NV >= 6 && NV < 10) { s0 += v[i]; s1 += v[i + 1]; }
            asm volatile("" : "+v"(v[i]), "+v"(v[i + 1]), "+v"(s0), "+v"(s1), "+v"(tm));
        }
        m = tm;
    }
    unsigned long long t1 = clock64();
    float r = s0 + s1 + m;
    f32x16 s = c0 + c1;
    for (int i = 0; i < 16; ++i) r += s[i] + (float)(d0[i] + d1[i]);
    r += e0[0] + e1[0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int MODE, int NV>
double run(float *out, unsigned long long *cyc) {
    unsigned long long h[8];
    const int n = 2000;
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(512), 0, 0, out, cyc, n);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int i = 0; i < 8; ++i) mx = h[i] > mx ? h[i] : mx;
    return (double)mx / (2.0 * 8.0 * n);      // cycles per slot and SIMD (two waves per SIMD, one after the other's slots)
}

template <int NV>
void table(float *out, unsigned long long *cyc) {
    const char *nm[8] = {"no matrix instruction", "f16 32x32x16", "i8 32x32x32", "2 x f16 16x16x32", "f16 32x32x16, C = 0", "f16 32x32x16, A/B in AGPRs",
                         "f16 / i8 alternating", "f16 32x32x16, C/D in AGPRs"};
    double t[8] = {run<0, NV>(out, cyc), run<1, NV>(out, cyc), run<2, NV>(out, cyc), run<3, NV>(out, cyc), run<4, NV>(out, cyc), run<5, NV>(out, cyc),
                   run<6, NV>(out, cyc), run<7, NV>(out, cyc)};
    printf("%d VALU instructions behind each matrix instruction (slot), two waves per SIMD:\n", NV);
    for (int i = 0; i < 8; ++i) printf("  %-30s %6.1f cycles per slot and SIMD   (+%.1f over the VALU stream alone)\n", nm[i], t[i], t[i] - t[0]);
}

int main() {
    float *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, 1 << 22)); CHECK(hipMalloc(&cyc, 64));
    table<10>(out, cyc);
    table<6>(out, cyc);
    table<4>(out, cyc);
    table<2>(out, cyc);
    return 0;
}
