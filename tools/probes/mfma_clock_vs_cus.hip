// Probe: shader clock (clock64 ticks / wall time) and matrix throughput of a pure v_mfma_f32_32x32x16_f16 loop as a function of the
// number of CUs it runs on (one 8-wave workgroup per CU).   hipcc --offload-arch=gfx950 -O2 -o mfma_clock_vs_cus mfma_clock_vs_cus.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int DUTY>
__global__ __launch_bounds__(512, 1) void k(float *out, unsigned long long *cyc, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        if (DUTY) __builtin_amdgcn_s_sleep(DUTY);       // idle part of the loop: lower matrix-pipe duty cycle
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
    const int iters = 40000;
    for (int duty = 0; duty <= 8; duty += 4)
        for (int blocks : {32, 64, 96, 128, 160, 192, 224, 256}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&](int n) {
                if (duty == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, out, cyc, n);
                else if (duty == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(512), 0, 0, out, cyc, n);
                else hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(512), 0, 0, out, cyc, n);
            };
            launch(iters / 4);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            launch(iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const double flops = 32768.0 * 4.0 * iters * 8 * blocks;
            printf("s_sleep %d  %3d CUs: %.2f ms  %.2f GHz  %7.0f TFLOP/s  (%.2f TF per CU)\n", duty, blocks, ms, (double)h / (ms * 1e6),
                   flops / ms / 1e9, flops / ms / 1e9 / blocks);
        }
    return 0;
}
