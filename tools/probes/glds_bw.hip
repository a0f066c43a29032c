// Probe: how fast can a CU pull L2-resident data into LDS?  global_load_lds (16 B per lane, LDS-DMA) against
// global_load_dwordx4 into registers (+ ds_write), 1..3 workgroups of 256 threads per CU, every CU streaming its own
// slice of a buffer that fits L2 / MALL.   hipcc --offload-arch=gfx950 -O3 -o glds_bw glds_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define GLDS16(gp, lp) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp), \
                                                        (__attribute__((address_space(3))) void *)(lp), 16, 0, 0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// each workgroup reads `bytes_per_wg` bytes (a private contiguous slice, wrapped over `span` bytes) in pieces of
// 4 KB per instruction group (256 threads x 16 B); PIECES groups in flight
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void k(const float *src, size_t span_f, int iters, float *sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    size_t base = ((size_t)blockIdx.x * 7919u * 1024u) % span_f;
    f32x4 accv = {0, 0, 0, 0};
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                size_t off = (base + (size_t)(it * DEPTH + d) * 1024) % span_f;      // 4 KB per group
                GLDS16(src + off + wave * 256 + lane * 4, lds + (d * 4 + wave) * 256);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        accv[0] = lds[tid];
    } else {
        for (int it = 0; it < iters; ++it) {
            f32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                size_t off = (base + (size_t)(it * DEPTH + d) * 1024) % span_f;
                v[d] = *reinterpret_cast<const f32x4 *>(src + off + tid * 4);
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (MODE == 2) *reinterpret_cast<f32x4 *>(lds + d * 1024 + tid * 4) = v[d];
                else accv += v[d];
            }
        }
        if (MODE == 2) { __syncthreads(); accv[0] += lds[tid]; }
    }
    if (accv[0] == 12345.678f) sink[0] = accv[0] + accv[1];
}

template <int MODE, int DEPTH>
static void run(const char *name, const float *src, size_t span_bytes, int wg_per_cu, float *sink) {
    const int iters = 4096 / DEPTH;
    const int grid = 256 * wg_per_cu;
    auto kern = k<MODE, DEPTH>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 48 * 1024, 0, src, span_bytes / 4, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)grid * iters * DEPTH * 4096.0;
    printf("%-34s depth %2d  %d WG/CU  span %5zu MB: %6.2f TB/s  (%.1f GB/s per CU)\n", name, DEPTH, wg_per_cu, span_bytes >> 20,
           bytes / best / 1e9, bytes / best / 1e6 / 256);
}

int main() {
    float *src, *sink;
    const size_t big = 512ull << 20;
    hipMalloc(&src, big); hipMalloc(&sink, 64);
    hipMemset(src, 0, big);
    for (size_t span : {(size_t)8 << 20, (size_t)64 << 20, big})
        for (int w : {1, 3}) {
            run<0, 4>("global_load_lds x4", src, span, w, sink);
            run<0, 12>("global_load_lds x4", src, span, w, sink);
            run<1, 4>("global_load_dwordx4 -> regs", src, span, w, sink);
            run<1, 12>("global_load_dwordx4 -> regs", src, span, w, sink);
            run<2, 8>("global_load_dwordx4 -> ds_write", src, span, w, sink);
        }
    return 0;
}
