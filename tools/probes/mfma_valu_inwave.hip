// Probe: how many VALU instructions (the vocabulary kernel's fold: fma, exp2, add) ride for free BETWEEN the MFMAs of the same
// wave, two waves per SIMD both doing the same (8 waves per CU).  Per MFMA: N fold elements = N x (v_fma_f32, v_exp_f32, v_add_f32).
//   hipcc --offload-arch=gfx950 -O3 -o build_prof/mfma_valu_inwave tools/probes/mfma_valu_inwave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int N>
__global__ __launch_bounds__(512, 1) void k(float *out, unsigned long long *cyc, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 c[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) c[j][i] = 0.0f;
    f32x16 prev;                                 // the "previous tile's" logits being folded
    for (int i = 0; i < 16; ++i) prev[i] = threadIdx.x * 0.01f + i;
    float s0 = 0.f, s1 = 0.f, m = 3.0f, ds = 0.25f;
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const float v = __builtin_fmaf(prev[(4 * j + e) & 15], ds, -m);
                const float x = __builtin_amdgcn_exp2f(v);
                if (e & 1) s1 += x; else s0 += x;
            }
            // one MFMA, then its VALU group
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (N) __builtin_amdgcn_sched_group_barrier(0x002, 3 * N, 0);
        }
        asm volatile("" : "+v"(prev));            // keep the fold inputs opaque across iterations
    }
    unsigned long long t1 = clock64();
    float r = s0 + s1;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) r += c[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int N> void run(float *out, unsigned long long *cyc) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<N>, dim3(256), dim3(512), 0, 0, out, cyc, 2000);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<N>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 4.0 * iters;
    printf("%d fold elements (%2d VALU) per MFMA: %6.1f ns per MFMA and SIMD-pair  %5.1f clk per MFMA (wave 0)  %.2f GHz  %6.0f TF executed\n", N, 3 * N,
           ms * 1e6 / n / 2, (double)h / n, (double)h / (ms * 1e6), n * 8 * 256 * 32768.0 / ms / 1e9);
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    run<0>(out, cyc); run<1>(out, cyc); run<2>(out, cyc); run<3>(out, cyc); run<4>(out, cyc); run<6>(out, cyc); run<8>(out, cyc);
    return 0;
}
