// Probe: how many bytes per cycle and CU does LDS deliver to ds_read_b128 / ds_read_b64 / ds_read_b32 (conflict-free addresses),
// alone and with one 32x32x16 f16 matrix instruction per read beside it?  8 waves per CU (2 per SIMD), one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 -o lds_read_rate lds_read_rate.hip && ./lds_read_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int W, bool MFMA>          // W = bytes per lane per read (16, 8, 4)
__global__ __launch_bounds__(512, 1) void k(int iters, float *sink, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += 512) reinterpret_cast<float *>(lds)[i] = (float)i;
    __syncthreads();
    // lane-linear addresses: consecutive lanes read consecutive W-byte words (conflict-free for every width)
    const unsigned char *p = lds + lane * W;
    f32x4 a = {0, 0, 0, 0};
    f32x16 acc = {0};
    const f16x8 t = {1, 1, 1, 1, 1, 1, 1, 1};
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned char *q = p + ((it * 8 + u) & 31) * 1024;
            f32x4 v = {0, 0, 0, 0};
            if (W == 16) v = *reinterpret_cast<const f32x4 *>(q);
            else if (W == 8) { const f32x2 w = *reinterpret_cast<const f32x2 *>(q); v[0] = w[0]; v[1] = w[1]; }
            else v[0] = *reinterpret_cast<const float *>(q);
            if (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, v), t, acc, 0, 0, 0);
            else asm volatile("" :: "v"(v));
        }
    }
    const unsigned long long c1 = clock64();
    if (tid == 0) cyc[blockIdx.x] = c1 - c0;
    if (acc[0] == 12345.678f || a[0] == 1.5f) sink[0] = acc[0];
}

template <int W, bool MFMA>
static void run(const char *name, float *sink, unsigned long long *cyc) {
    const int iters = 2048;
    auto kern = k<W, MFMA>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    unsigned long long h[256];
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 64 * 1024, 0, iters, sink, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; ++i) c += h[i]; c /= 256;
    const double reads = 8.0 * iters * 8;         // wave-instructions per CU
    printf("%-28s %8.1f us  %9.0f cycles/CU  %6.1f cycles per wave-read and CU  %6.1f B/cycle/CU  (%.2f GHz)\n", name, best * 1e3, c,
           c / reads, reads * 64 * W / c, c / (best * 1e6));
}

int main() {
    float *sink; unsigned long long *cyc;
    hipMalloc(&sink, 64); hipMalloc(&cyc, 256 * 8);
    run<16, false>("ds_read_b128", sink, cyc);
    run<8, false>("ds_read_b64", sink, cyc);
    run<4, false>("ds_read_b32", sink, cyc);
    run<16, true>("ds_read_b128 + mfma", sink, cyc);
    run<8, true>("ds_read_b64 + mfma", sink, cyc);
    return 0;
}
