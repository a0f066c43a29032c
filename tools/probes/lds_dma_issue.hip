// Probe: what does it cost a wave to ISSUE one LDS-DMA instruction (1 KB: 64 lanes x 16 B, L2-resident source)?  One wave per SIMD
// (256 threads), bursts of NB instructions back to back, the shader clock read in front of and behind the burst (not behind the data:
// the wait follows outside the timed span); forms: global_load_lds, raw buffer (offen), struct buffer (idxen + offen), 4 bytes per
// lane instead of 16, an ordinary global_load_dwordx4 into registers and a ds_read_b128 for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, int NB>
__global__ __launch_bounds__(256) void k(const float *src, int n_rows, unsigned long long *out, float *sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), (short)1024, 0x40000000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, n_rows * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t r16 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), (short)16, 0x7fffffff, 0x00020000);
    unsigned long long total = 0;
    f32x4 acc = {0, 0, 0, 0};
    float *my = lds + wave * (NB * 256);
    for (int it = 0; it < iters; ++it) {
        const int row0 = ((blockIdx.x * 37 + it * 11 + wave * 3) * NB) % (n_rows - NB);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t0 = clock64();
        f32x4 v[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (MODE == 0)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)(row0 + b) * 256 + lane * 4),
                                                 (__attribute__((address_space(3))) void *)(my + b * 256), 16, 0, 0);
            else if (MODE == 1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (__attribute__((address_space(3))) void *)(my + b * 256), 16, (row0 + b) * 1024 + lane * 16, 0, 0, 0);
            else if (MODE == 2)
                __builtin_amdgcn_struct_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(my + b * 256), 16, row0 + b, lane * 16, 0, 0, 0);
            else if (MODE == 3)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (__attribute__((address_space(3))) void *)(my + b * 64), 4, (row0 + b) * 1024 + lane * 4, 0, 0, 0);
            else if (MODE == 6)      // struct form, 16-byte records, index only
                __builtin_amdgcn_struct_ptr_buffer_load_lds(r16, (__attribute__((address_space(3))) void *)(my + b * 256), 16, (row0 + b) * 64 + lane, 0, 0, 0, 0);
            else if (MODE == 4)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v[b]) : "v"(src + (size_t)(row0 + b) * 256 + lane * 4) : "memory");
            else
                asm volatile("ds_read_b128 %0, %1" : "=&v"(v[b]) : "v"((int)((wave * NB + b) * 1024 + lane * 16)) : "memory");
        }
        const unsigned long long t1 = clock64();
        total += t1 - t0;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (MODE == 4 || MODE == 5) {
#pragma unroll
            for (int b = 0; b < NB; ++b) { asm volatile("" : "+v"(v[b])); acc += v[b]; }
        }
    }
    __syncthreads();
    acc[0] += lds[tid];
    if (acc[0] == 12345.678f) sink[0] = acc[0];
    if (lane == 0 && blockIdx.x == 0) out[wave] = total;
}

template <int MODE, int NB>
int run(const char *name, const float *src, int n_rows, unsigned long long *out, float *sink, int grid) {
    const int iters = 2000;
    auto kern = k<MODE, NB>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 64 * 1024, 0, src, n_rows, out, sink, iters);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[4];
    CHECK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
    printf("  %-44s burst of %2d, %3d workgroups: %6.1f cycles per instruction (wave 0; %.1f wave 3)\n", name, NB, grid, (double)h[0] / iters / NB, (double)h[3] / iters / NB);
    return 0;
}

int main() {
    float *src, *sink; unsigned long long *out;
    const int n_rows = 4096;                        // 4 MB: L2-resident
    CHECK(hipMalloc(&src, (size_t)n_rows * 1024)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(src, 0, (size_t)n_rows * 1024));
    for (int grid : {1, 256}) {
        run<0, 4>("global_load_lds 16 B", src, n_rows, out, sink, grid);
        run<0, 8>("global_load_lds 16 B", src, n_rows, out, sink, grid);
        run<1, 4>("raw buffer_load ... lds 16 B (offen)", src, n_rows, out, sink, grid);
        run<1, 8>("raw buffer_load ... lds 16 B (offen)", src, n_rows, out, sink, grid);
        run<2, 8>("struct buffer_load ... lds 16 B (idxen offen)", src, n_rows, out, sink, grid);
        run<6, 8>("struct buffer_load ... lds 16 B (idxen only)", src, n_rows, out, sink, grid);
        run<3, 8>("raw buffer_load ... lds 4 B", src, n_rows, out, sink, grid);
        run<4, 8>("global_load_dwordx4 -> registers", src, n_rows, out, sink, grid);
        run<5, 8>("ds_read_b128", src, n_rows, out, sink, grid);
    }
    return 0;
}
