// Probe: issue cost of the fold's instructions on gfx950 -- v_fma_f32, v_exp_f32 (transcendental), v_max3_f32,
// v_pk_fma_f32 and the fold's own mix (max3/2 + fma + exp + add per element), 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters, float seed) {
    float x[16], acc0 = 0.f, acc1 = 0.f, m = -1e30f;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = seed + threadIdx.x * 1e-3f + i;
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {            // 16 independent fma
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
        } else if (MODE == 1) {     // 16 independent exp2
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
        } else if (MODE == 2) {     // the fold of 16 elements: max3 x8, then fma + exp2 + add each
            float t = m;
#pragma unroll
            for (int i = 0; i < 16; i += 2) t = fmaxf(fmaxf(t, x[i]), x[i + 1]);
            const float nm = -t;
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                acc0 += __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], 0.001f, nm));
                acc1 += __builtin_amdgcn_exp2f(__builtin_fmaf(x[i + 1], 0.001f, nm));
            }
            m = t * 0.999f;
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(x[i]));
        } else if (MODE == 3) {     // same without the max (fixed shift)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                acc0 += __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], 0.001f, m));
                acc1 += __builtin_amdgcn_exp2f(__builtin_fmaf(x[i + 1], 0.001f, m));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(x[i]));
        } else if (MODE == 4) {     // 8 packed fma (16 elements)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2 v = {x[i], x[i + 1]}, a = {1.0001f, 1.0001f}, b = {0.5f, 0.5f};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
                x[i] = v[0]; x[i + 1] = v[1];
            }
        } else if (MODE == 5) {     // exp2 only half of the elements + the fma/add of all (is the trans unit the limit?)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                acc0 += __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], 0.001f, m));
                acc1 += __builtin_fmaf(x[i + 1], 0.001f, m);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(x[i]));
        }
    }
    unsigned long long t1 = clock64();
    float r = acc0 + acc1 + m;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    const int iters = 20000;
    const char *names[] = {"v_fma_f32 x16", "v_exp_f32 x16", "fold of 16 (max3/2 + fma + exp + add)", "fold of 16 without the max",
                           "v_pk_fma_f32 x8 (16 elements)", "fold of 16, exp on half the elements"};
    for (int waves = 4; waves <= 8; waves += 4)
        for (int mode = 0; mode < 6; ++mode) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters, 1.0f);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters, 1.0f);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters, 1.0f);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters, 1.0f);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters, 1.0f);
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters, 1.0f);
            hipDeviceSynchronize();
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("%d waves/SIMD  %-44s %.1f cycles per element and wave\n", waves / 4, names[mode], (double)h / iters / 16.0);
        }
    return 0;
}
