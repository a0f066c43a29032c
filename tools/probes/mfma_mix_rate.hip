// Probe: what mixed f16 / i8 MFMA streams cost against the three f16 passes of the split scheme, on RANDOM operands with every
// SIMD busy (the clock the governor grants is part of the answer).  Units of 32 k-values per 32x32 block:
//   mode 0  6 x v_mfma_f32_32x32x16_f16                      (hi.hi, hi.lo, lo.hi over two 16-wide steps: today)
//   mode 1  2 x f16 32x32x16 + 2 x v_mfma_i32_32x32x32_i8    (hi.hi in f16, the two cross terms in int8)
//   mode 2  2 x f16 32x32x16                                 (hi.hi alone)
//   mode 3  6 x v_mfma_i32_32x32x32_i8
//   mode 4  4 x v_mfma_f32_32x32x8_f16  (legacy half-width step: what a k tail of 8 would cost)
// hipcc --offload-arch=gfx950 -O2 -o mfma_mix_rate mfma_mix_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(const int *__restrict__ rnd, float *out, unsigned long long *cyc, int iters) {
    // eight operand sets of random bits per lane (f16 values scaled into a sane range by masking the exponent)
    i32x4 ra[8], rb[8];
    for (int s = 0; s < 8; ++s) {
        ra[s] = *reinterpret_cast<const i32x4 *>(rnd + ((threadIdx.x * 8 + s) * 8) % 65536);
        rb[s] = *reinterpret_cast<const i32x4 *>(rnd + ((threadIdx.x * 8 + s) * 8 + 4) % 65536);
        if (MODE != 3) for (int j = 0; j < 4; ++j) { ra[s][j] = (ra[s][j] & 0xbbffbbff) | 0x20002000; rb[s][j] = (rb[s][j] & 0xbbffbbff) | 0x20002000; }
    }
    f32x16 c[4] = {};
    i32x16 ci[2] = {};
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {           // 4 blocks per iteration, operand sets rotate
            const f16x8 a0 = __builtin_bit_cast(f16x8, ra[(2 * u) & 7]), a1 = __builtin_bit_cast(f16x8, ra[(2 * u + 1) & 7]);
            const f16x8 b0 = __builtin_bit_cast(f16x8, rb[(2 * u) & 7]), b1 = __builtin_bit_cast(f16x8, rb[(2 * u + 1) & 7]);
            if (MODE == 0) {
                c[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c[u], 0, 0, 0);
                c[(u + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c[(u + 1) & 3], 0, 0, 0);
                c[(u + 2) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c[(u + 2) & 3], 0, 0, 0);
                c[(u + 3) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c[(u + 3) & 3], 0, 0, 0);
                c[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c[u], 0, 0, 0);
                c[(u + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c[(u + 1) & 3], 0, 0, 0);
            } else if (MODE == 1) {
                c[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c[u], 0, 0, 0);
                ci[u & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ra[(2 * u + 1) & 7], rb[(2 * u) & 7], ci[u & 1], 0, 0, 0);
                c[(u + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c[(u + 1) & 3], 0, 0, 0);
                ci[(u + 1) & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ra[(2 * u) & 7], rb[(2 * u + 1) & 7], ci[(u + 1) & 1], 0, 0, 0);
            } else if (MODE == 2) {
                c[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c[u], 0, 0, 0);
                c[(u + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c[(u + 1) & 3], 0, 0, 0);
            } else if (MODE == 3) {
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    ci[q & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ra[(2 * u + q) & 7], rb[(2 * u + q + 3) & 7], ci[q & 1], 0, 0, 0);
            } else {
                const f16x4 x0 = {a0[0], a0[1], a0[2], a0[3]}, y0 = {b0[0], b0[1], b0[2], b0[3]};
                const f16x4 x1 = {a1[0], a1[1], a1[2], a1[3]}, y1 = {b1[0], b1[1], b1[2], b1[3]};
                c[u] = __builtin_amdgcn_mfma_f32_32x32x8f16(x0, y0, c[u], 0, 0, 0);
                c[(u + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x8f16(x1, y1, c[(u + 1) & 3], 0, 0, 0);
                c[(u + 2) & 3] = __builtin_amdgcn_mfma_f32_32x32x8f16(x0, y1, c[(u + 2) & 3], 0, 0, 0);
                c[(u + 3) & 3] = __builtin_amdgcn_mfma_f32_32x32x8f16(x1, y0, c[(u + 3) & 3], 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = clock64();
    float r = 0;
    for (int u = 0; u < 4; ++u) for (int i = 0; i < 16; ++i) r += c[u][i];
    for (int u = 0; u < 2; ++u) for (int i = 0; i < 16; ++i) r += (float)ci[u][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float *out; unsigned long long *cyc, h; int *rnd;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8); hipMalloc(&rnd, 65536 * 4 + 64);
    int *hr = (int *)malloc(65536 * 4 + 64);
    srand(7);
    for (int i = 0; i < 65536 + 16; ++i) hr[i] = (rand() << 16) ^ rand();
    hipMemcpy(rnd, hr, 65536 * 4 + 64, hipMemcpyHostToDevice);
    const int iters = 4000;
    const char *names[] = {"6 x f16 32x32x16 (three passes, today)", "2 x f16 + 2 x i8 32x32x32 (int8 cross terms)", "2 x f16 (hi.hi alone)",
                           "6 x i8 32x32x32", "4 x f16 32x32x8 (legacy half step)"};
    const int per_unit[] = {6, 4, 2, 6, 4};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 5; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int w = 0; w < 2; ++w) {
                if (w) hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, rnd, out, cyc, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, rnd, out, cyc, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, rnd, out, cyc, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, rnd, out, cyc, iters);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, rnd, out, cyc, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const double units = 4.0 * iters;            // 32-k units per wave
            printf("%-48s %6.1f cycles per unit and wave (%4.1f per MFMA)  %.3f ms  %.2f GHz  -> %.1f ns per unit and SIMD\n", names[mode],
                   (double)h / units, (double)h / units / per_unit[mode], ms, (double)h / (ms * 1e6), ms * 1e6 / units / 2);
        }
    return 0;
}
