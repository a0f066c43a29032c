// layout check of v_mfma_i32_32x32x32_i8: A[m][k] from lane (m = l & 31, k = 16 (l >> 5) + byte), B[k][n] likewise with n = l & 31;
// D[m][n]: lane n + 32 h holds rows m = (r & 3) + 8 (r >> 2) + 4 h in register r (the 32x32 f32 layout)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const signed char *A, const signed char *B, int *D) {
    const int l = threadIdx.x, m = l & 31, h = l >> 5;
    i32x4 a, b;
    for (int q = 0; q < 4; ++q) {
        int va = 0, vb = 0;
        for (int e = 0; e < 4; ++e) {
            const int kk = 16 * h + 4 * q + e;
            va |= ((int)(unsigned char)A[m * 32 + kk]) << (8 * e);
            vb |= ((int)(unsigned char)B[kk * 32 + m]) << (8 * e);
        }
        a[q] = va; b[q] = vb;
    }
    i32x16 c = {};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + m] = c[r];
}
int main() {
    signed char hA[1024], hB[1024]; int hD[1024], ref[1024];
    srand(3);
    for (int i = 0; i < 1024; ++i) { hA[i] = (signed char)(rand() % 255 - 127); hB[i] = (signed char)(rand() % 255 - 127); }
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { int s = 0; for (int kk = 0; kk < 32; ++kk) s += (int)hA[m * 32 + kk] * (int)hB[kk * 32 + n]; ref[m * 32 + n] = s; }
    signed char *A, *B; int *D;
    hipMalloc(&A, 1024); hipMalloc(&B, 1024); hipMalloc(&D, 4096);
    hipMemcpy(A, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(B, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, D);
    hipMemcpy(hD, D, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += hD[i] != ref[i];
    printf("i8 32x32x32 layout: %d of 1024 outputs differ\n", bad);
    return bad != 0;
}
