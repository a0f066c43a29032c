// Probe: the k-step loop of jlm_lstm_step_xg in isolation (no global memory): 8 waves per CU, waves 0-3 with 3 accumulator
// blocks, waves 4-7 with 2; per k-step 2 half steps of 3 x NB MFMAs.  Features: fragment reads from LDS (prefetched one
// half step ahead, as the kernel does), one s_barrier per k-step, 4 fake "DMA issue" scalar sections.
//   hipcc --offload-arch=gfx950 -O3 -o build_prof/gate_loop tools/probes/gate_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NB, bool READS, bool BAR, int ILV>
__device__ __forceinline__ void body(float *out, unsigned long long *cyc, int iters, float *smem, int wave, int lane) {
    const int li = lane & 31, hf = lane >> 5;
    int goff[2][2];
    for (int st = 0; st < 2; ++st)
        for (int p = 0; p < 2; ++p) goff[st][p] = li * 32 + (((4 * st + 2 * hf + p) ^ ((li >> 1) & 7)) * 4);
    const int w_off = (wave & 3) * 1024, h_off = (128 + ((wave >> 2) ? 3 : 0) * 32) * 32;
    f32x16 acc[NB];
    for (int nb = 0; nb < NB; ++nb)
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
    struct Frag { f16x8 aw[2]; f16x8 bh[NB][2]; };
    Frag fa, fb;
    for (int p = 0; p < 2; ++p) {
        for (int i = 0; i < 8; ++i) { fa.aw[p][i] = (_Float16)(lane * 0.001f); fb.aw[p][i] = (_Float16)(0.5f); }
        for (int nb = 0; nb < NB; ++nb) { fa.bh[nb][p] = fa.aw[p]; fb.bh[nb][p] = fb.aw[p]; }
    }
    auto read_half = [&](int stage, int st, Frag &f) {
        if (!READS) return;
        const float *ws = smem + (stage & 3) * 9216 + w_off;
        const float *hs = smem + (stage & 3) * 9216 + h_off;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            f.aw[p] = *reinterpret_cast<const f16x8 *>(ws + goff[st][p]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) f.bh[nb][p] = *reinterpret_cast<const f16x8 *>(hs + nb * 1024 + goff[st][p]);
        }
    };
    auto mfmas = [&](const Frag &f) {
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.aw[pr == 0 ? 1 : 0], f.bh[nb][pr == 1 ? 1 : 0], acc[nb], 0, 0, 0);
    };
    auto touch = [&](Frag &f) {
        if (!READS) return;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            asm volatile("" : "+v"(f.aw[p]));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) asm volatile("" : "+v"(f.bh[nb][p]));
        }
    };
    __syncthreads();
    unsigned long long t0 = clock64();
    auto half = [&](int stage, int st, Frag &nxt, const Frag &cur) {
        if (ILV == 0) {
            read_half(stage, st, nxt);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(cur);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // one MFMA, one ds_read, one MFMA, ... (sched_group_barrier: mask 0x8 = MFMA, 0x100 = DS read)
            read_half(stage, st, nxt);
            mfmas(cur);
#pragma unroll
            for (int i = 0; i < 3 * NB; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (READS && i < 2 + 2 * NB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int kt = 0; kt < iters; ++kt) {
        if (ILV == 2) __builtin_amdgcn_s_setprio(1);
        half(kt, 1, fb, fa);
        touch(fb);
        if (ILV == 2) __builtin_amdgcn_s_setprio(0);
        if (BAR) asm volatile("s_barrier" ::: "memory");
        if (ILV == 2) __builtin_amdgcn_s_setprio(1);
        half(kt + 1, 0, fa, fb);
        touch(fa);
        if (ILV == 2) __builtin_amdgcn_s_setprio(0);
    }
    unsigned long long t1 = clock64();
    float r = 0;
    for (int nb = 0; nb < NB; ++nb)
        for (int i = 0; i < 16; ++i) r += acc[nb][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <bool READS, bool BAR, int NBA, int NBB, int ILV = 0>
__global__ __launch_bounds__(512, 1) void k(float *out, unsigned long long *cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int i = threadIdx.x; i < 4 * 9216; i += 512) smem[i] = 0.001f * (i & 255);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (wave < 4) body<NBA, READS, BAR, ILV>(out, cyc, iters, smem, wave, lane);
    else body<NBB, READS, BAR, ILV>(out, cyc, iters, smem, wave, lane);
}

template <bool READS, bool BAR, int NBA, int NBB, int ILV = 0>
void run(const char *name, float *out, unsigned long long *cyc) {
    const int iters = 4000;
    auto kern = k<READS, BAR, NBA, NBB, ILV>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 147456, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 147456, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const double mf = 6.0 * (NBA + NBB);          // MFMAs per SIMD and k-step
    printf("%-44s %7.1f ns per k-step  wave0 %6.0f clk  wave4 %6.0f clk per k-step  (pipe-bound: %4.0f clk)  %.2f GHz  %6.0f TF executed\n",
           name, ms * 1e6 / iters, (double)h[0] / iters, (double)h[4] / iters, mf * 32, (double)h[0] / (ms * 1e6),
           mf * 4 * 256 * 32768.0 * iters / ms / 1e9);
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 64);
    run<false, false, 3, 2>("MFMAs only, 3 + 2 blocks", out, cyc);
    run<false, true, 3, 2>("MFMAs + barrier, 3 + 2 blocks", out, cyc);
    run<true, false, 3, 2>("MFMAs + reads, 3 + 2 blocks", out, cyc);
    run<true, true, 3, 2>("MFMAs + reads + barrier, 3 + 2 blocks", out, cyc);
    run<false, false, 4, 4>("MFMAs only, 4 + 4 blocks", out, cyc);
    run<true, true, 4, 4>("MFMAs + reads + barrier, 4 + 4 blocks", out, cyc);
    run<false, false, 2, 2>("MFMAs only, 2 + 2 blocks", out, cyc);
    run<true, true, 2, 2>("MFMAs + reads + barrier, 2 + 2 blocks", out, cyc);
    run<false, false, 5, 1>("MFMAs only, 5 + 1 blocks", out, cyc);
    run<true, true, 3, 2, 1>("reads interleaved 1:1, barrier, 3 + 2", out, cyc);
    run<true, false, 3, 2, 1>("reads interleaved 1:1, no barrier, 3 + 2", out, cyc);
    run<true, true, 4, 4, 1>("reads interleaved 1:1, barrier, 4 + 4", out, cyc);
    run<true, true, 3, 2, 2>("grouped + setprio(1) around compute, 3 + 2", out, cyc);
    return 0;
}
