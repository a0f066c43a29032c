// Probe: issue rate of v_mfma_f32_32x32x16_f16 with the accumulators in arch VGPRs vs AGPRs,
// 1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O2 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k(float *out, unsigned long long *cyc, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {           // compiler's choice (arch VGPR accumulators under a 256-register budget)
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        } else if (MODE == 1) {    // accumulators pinned to AGPRs
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n"
                         "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n"
                         : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
        } else if (MODE == 2) {    // accumulators pinned to arch VGPRs
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n"
                         "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        } else {                   // f32 32x32x2 for reference
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, 2.0f, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, 2.0f, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, 2.0f, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, 2.0f, c3, 0, 0, 0);
        }
    }
    unsigned long long t1 = clock64();
    f32x16 s = c0 + c1 + c2 + c3;
    float r = 0;
    for (int i = 0; i < 16; ++i) r += s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    const int iters = 20000;
    const char *names[] = {"builtin (compiler-allocated)", "asm, acc in AGPR", "asm, acc in arch VGPR", "f32 32x32x2 builtin"};
    for (int waves = 4; waves <= 8; waves += 4)
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const double n = 4.0 * iters;            // MFMAs per wave
            const double flops = (mode == 3 ? 4096.0 : 32768.0) * n * waves * 256;
            printf("%d waves/CU  %-30s %.1f cycles/MFMA/wave  %.0f TFLOP/s  (%.2f GHz effective)\n", waves, names[mode],
                   (double)h / n, flops / ms / 1e9, (double)h / (ms * 1e6));
        }
    return 0;
}
