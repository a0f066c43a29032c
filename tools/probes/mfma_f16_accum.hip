// Probe: how does v_mfma_f32_32x32x16_f16 add 16 small products into a LARGE accumulator?
// (exactly-summed-then-rounded once, or each product aligned to the accumulator and truncated?)
//   hipcc --offload-arch=gfx950 -O2 -o mfma_f16_accum mfma_f16_accum.hip && ./mfma_f16_accum
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const _Float16 *A, const _Float16 *B, const float *C, float *D) {
    const int lane = threadIdx.x, li = lane & 31, h = lane >> 5;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = A[li * 16 + 8 * h + i]; b[i] = B[li * 16 + 8 * h + i]; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + li];
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + li] = c[r];
}

int main() {
    _Float16 hA[32 * 16], hB[32 * 16];
    float hC[32 * 32], hD[32 * 32];
    for (int cmag = 0; cmag <= 20; cmag += 4) {
        srand(1);
        for (int i = 0; i < 512; ++i) { hA[i] = (_Float16)((rand() % 2001 - 1000) / 1000.0f); hB[i] = (_Float16)((rand() % 2001 - 1000) / 1000.0f); }
        for (int i = 0; i < 1024; ++i) hC[i] = ldexpf(1.0f + (rand() % 1000) / 1000.0f, cmag);
        _Float16 *dA, *dB; float *dC, *dD;
        hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC)); hipMalloc(&dD, sizeof(hD));
        hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
        hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
        double worst = 0, bias = 0, worst_rn = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double s = 0;
                for (int kk = 0; kk < 16; ++kk) s += (double)(float)hA[i * 16 + kk] * (double)(float)hB[j * 16 + kk];
                const double exact = s + (double)hC[i * 32 + j];
                const double ulp = ldexp(1.0, cmag - 23);
                const double e = ((double)hD[i * 32 + j] - exact) / ulp;
                const double ern = ((double)(float)exact - exact) / ulp;
                if (fabs(e) > worst) worst = fabs(e);
                if (fabs(ern) > worst_rn) worst_rn = fabs(ern);
                bias += e;
            }
        printf("C ~ 2^%-2d : max |D - exact| = %.3f ulp(C), mean signed error %.3f ulp (correctly rounded sum would be <= %.3f)\n",
               cmag, worst, bias / 1024, worst_rn);
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
    }
    return 0;
}
