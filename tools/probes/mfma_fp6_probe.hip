// Probe (round 6, verdict item 1b): the block-scaled matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4 as the carrier of the split
// product's cross terms.
//   part A  operand layout and numerics of the FP6 (e2m3) form, decoded on the device and checked against a host emulation:
//           which k-slot a lane's i-th 6-bit field is, which lanes' scale bytes apply to which half of K, what op_sel selects;
//   part B  what the instruction mixes cost on RANDOM operands with every SIMD busy (the clock granted is part of the answer),
//           per unit of 32 k-values and 32 x 32 block:
//             mode 1  2 x f16 32x32x16 + 2 x i8 32x32x32                 (the shipped mixed rows)
//             mode 5  2 x f16 32x32x16 + 1 x f8f6f4 K=64 as FP6 x FP6    (both cross terms in one instruction)
//             mode 6  2 x f16 32x32x16 + 1 x f8f6f4 K=64 as FP8 x FP8
//             mode 7  2 x f16 32x32x16 + 1 x f8f6f4 K=64 as FP4 x FP4
//             mode 8  1 x f8f6f4 FP6 alone        mode 2  2 x f16 alone
// hipcc --offload-arch=gfx950 -O2 -o mfma_fp6_probe mfma_fp6_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------ part A
// one wave: D = A . B with per-lane operand dwords and scale dwords given by the host
template <int OPA, int OPB>
__global__ void mm_fp6(const int *a, const int *b, const int *sa, const int *sb, float *d) {
    const int l = threadIdx.x;
    i32x8 va = {}, vb = {};
    for (int i = 0; i < 6; ++i) { va[i] = a[l * 6 + i]; vb[i] = b[l * 6 + i]; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 2, 2, OPA, sa[l], OPB, sb[l]);
    for (int r = 0; r < 16; ++r) d[l * 16 + r] = c[r];
}

static float e2m3(int code) {                 // OCP MX FP6 e2m3: sign, 2 exponent bits (bias 1), 3 mantissa bits; no inf / nan
    const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
    const float v = e == 0 ? m * 0.125f : ldexpf(1.0f + m * 0.125f, e - 1);
    return s ? -v : v;
}

// hypothesis under test: lane l holds row (l & 31), k-slots 32 (l >> 5) + i, i = 0 .. 31, field i at bits [6 i, 6 i + 6) of its 192 bits;
// its scale byte (selected by op_sel from its scale dword) applies to exactly those 32 slots
static void pack6(const int *codes /* [32 rows][64 slots] */, int *dw /* [64 lanes][6] */) {
    memset(dw, 0, 64 * 6 * 4);
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 32; ++i) {
            const unsigned c = (unsigned)codes[(l & 31) * 64 + 32 * (l >> 5) + i] & 63u;
            const int bit = 6 * i;
            dw[l * 6 + bit / 32] |= (int)(c << (bit % 32));
            if (bit % 32 > 26) dw[l * 6 + bit / 32 + 1] |= (int)(c >> (32 - bit % 32));
        }
}

static int part_a() {
    int *ca = (int *)malloc(32 * 64 * 4), *cb = (int *)malloc(32 * 64 * 4);
    int ha[64 * 6], hb[64 * 6], hsa[64], hsb[64];
    int *da, *db, *dsa, *dsb; float *dd;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dsa, sizeof hsa); hipMalloc(&dsb, sizeof hsb); hipMalloc(&dd, 64 * 16 * 4);
    float hd[64 * 16];
    int bad_total = 0;
    for (int test = 0; test < 4; ++test) {
        srand(11 + test);
        for (int i = 0; i < 32 * 64; ++i) { ca[i] = rand() & 63; cb[i] = rand() & 63; }
        // scale dwords: four different bytes; the one op_sel = test picks is the lane's real scale (row- and half-dependent)
        int ea[64], eb[64];
        for (int l = 0; l < 64; ++l) {
            ea[l] = 120 + (l * 7) % 13; eb[l] = 125 + (l * 5) % 11;
            unsigned wa = 0x01010101u * 90u, wb = 0x01010101u * 200u;      // decoys: 2^-37, 2^73
            wa = (wa & ~(0xffu << (8 * test))) | ((unsigned)ea[l] << (8 * test));
            wb = (wb & ~(0xffu << (8 * ((test + 1) & 3)))) | ((unsigned)eb[l] << (8 * ((test + 1) & 3)));
            hsa[l] = (int)wa; hsb[l] = (int)wb;
        }
        pack6(ca, ha); pack6(cb, hb);
        hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
        hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
        if (test == 0) hipLaunchKernelGGL((mm_fp6<0, 1>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
        if (test == 1) hipLaunchKernelGGL((mm_fp6<1, 2>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
        if (test == 2) hipLaunchKernelGGL((mm_fp6<2, 3>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
        if (test == 3) hipLaunchKernelGGL((mm_fp6<3, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
        hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
        // D[i][j]: i = A row (vocabulary word in the kernel), j = B row; C/D map: lane l holds column j = l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)
        double worst = 0.0, scale = 0.0;
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
                double want = 0.0;
                for (int h = 0; h < 2; ++h) {
                    double sum = 0.0;
                    for (int s = 0; s < 32; ++s) sum += (double)e2m3(ca[i * 64 + 32 * h + s]) * (double)e2m3(cb[j * 64 + 32 * h + s]);
                    want += ldexp(sum, (ea[i + 32 * h] - 127) + (eb[j + 32 * h] - 127));
                }
                const double err = fabs(want - hd[l * 16 + r]);
                if (err > worst) worst = err;
                if (fabs(want) > scale) scale = fabs(want);
                if (err > 1e-6 * fabs(want) + 1e-30) ++bad;
            }
        printf("part A test %d (op_sel a = %d, b = %d): max |D - emulation| = %.3e of max |D| = %.3e, %d of 1024 elements off\n", test, test, (test + 1) & 3,
               worst, scale, bad);
        bad_total += bad;
    }
    if (bad_total) {
        // decode: a single 1.0 in A at (lane la, field fa) against a single 1.0 in B at (lane lb, field fb): which pairs meet, and where in D
        printf("layout hypothesis FAILED: singleton decode follows (A lane, field) x (B lane, field) -> non-zero D entries\n");
        const int lanes[4] = {0, 1, 32, 33}, fields[4] = {0, 1, 5, 31};
        for (int x = 0; x < 4; ++x)
            for (int y = 0; y < 4; ++y) {
                memset(ha, 0, sizeof ha);
                const int bit = 6 * fields[y];
                ha[lanes[x] * 6 + bit / 32] |= 8 << (bit % 32);
                if (bit % 32 > 26) ha[lanes[x] * 6 + bit / 32 + 1] |= 8 >> (32 - bit % 32);
                for (int l = 0; l < 64; ++l) { hsa[l] = 0x7f7f7f7f; hsb[l] = 0x7f7f7f7f; }
                // B: every field of every lane = 1.0 x a value that encodes (lane half, field): use B all 1.0 first, just to see the row
                for (int l = 0; l < 64; ++l) for (int i = 0; i < 6; ++i) hb[l * 6 + i] = 0;
                for (int l = 0; l < 64; ++l) for (int f = 0; f < 32; ++f) { const int bb = 6 * f; hb[l * 6 + bb / 32] |= 8 << (bb % 32); if (bb % 32 > 26) hb[l * 6 + bb / 32 + 1] |= 8 >> (32 - bb % 32); }
                hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
                hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
                hipLaunchKernelGGL((mm_fp6<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
                hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
                int nz = 0, first = -1; float val = 0;
                for (int q = 0; q < 1024; ++q) if (hd[q] != 0.0f) { if (first < 0) { first = q; val = hd[q]; } ++nz; }
                printf("  A lane %2d field %2d: %d non-zero D entries, first at lane %d reg %d = %g\n", lanes[x], fields[y], nz, first / 16, first % 16, val);
            }
    }
    return bad_total;
}

// ------------------------------------------------------------------------------------------------ part B
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(const int *__restrict__ rnd, float *out, unsigned long long *cyc, int iters) {
    i32x4 ra[8], rb[8];
    i32x8 wa[4], wb[4];
    for (int s = 0; s < 8; ++s) {
        ra[s] = *reinterpret_cast<const i32x4 *>(rnd + ((threadIdx.x * 8 + s) * 8) % 65536);
        rb[s] = *reinterpret_cast<const i32x4 *>(rnd + ((threadIdx.x * 8 + s) * 8 + 4) % 65536);
    }
    for (int s = 0; s < 4; ++s)
        for (int j = 0; j < 8; ++j) {
            wa[s][j] = rnd[(threadIdx.x * 64 + s * 16 + j + 1000) % 65536];
            wb[s][j] = rnd[(threadIdx.x * 64 + s * 16 + j + 8 + 1000) % 65536];
            if (MODE == 6) { wa[s][j] &= 0x77777777 | 0x80808080; wb[s][j] &= 0xf7f7f7f7; wa[s][j] &= 0xbfbfbfbf; wb[s][j] &= 0xbfbfbfbf; }   // FP8 e4m3: keep off nan
        }
    for (int s = 0; s < 8; ++s)
        for (int j = 0; j < 4; ++j) { ra[s][j] = (ra[s][j] & 0xbbffbbff) | 0x20002000; rb[s][j] = (rb[s][j] & 0xbbffbbff) | 0x20002000; }
    const int sc_a = 0x60606060 + (threadIdx.x & 3), sc_b = 0x5e5e5e5e + ((threadIdx.x >> 2) & 3);
    f32x16 c[4] = {};
    i32x16 ci[2] = {};
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f16x8 a0 = __builtin_bit_cast(f16x8, ra[(2 * u) & 7]), a1 = __builtin_bit_cast(f16x8, ra[(2 * u + 1) & 7]);
            const f16x8 b0 = __builtin_bit_cast(f16x8, rb[(2 * u) & 7]), b1 = __builtin_bit_cast(f16x8, rb[(2 * u + 1) & 7]);
            if (MODE == 1) {
                c[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c[u], 0, 0, 0);
                ci[u & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ra[(2 * u + 1) & 7], rb[(2 * u) & 7], ci[u & 1], 0, 0, 0);
                c[(u + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c[(u + 1) & 3], 0, 0, 0);
                ci[(u + 1) & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ra[(2 * u) & 7], rb[(2 * u + 1) & 7], ci[(u + 1) & 1], 0, 0, 0);
            } else if (MODE == 2) {
                c[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c[u], 0, 0, 0);
                c[(u + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c[(u + 1) & 3], 0, 0, 0);
            } else if (MODE == 5 || MODE == 6 || MODE == 7) {
                constexpr int FMT = MODE == 5 ? 2 : MODE == 6 ? 0 : 4;
                c[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c[u], 0, 0, 0);
                c[(u + 1) & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa[u], wb[u], c[(u + 1) & 3], FMT, FMT, 0, sc_a, 1, sc_b);
                c[(u + 2) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c[(u + 2) & 3], 0, 0, 0);
            } else if (MODE == 8) {
                c[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa[u], wb[u], c[u], 2, 2, 0, sc_a, 1, sc_b);
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
    const int bad = part_a();
    float *out; unsigned long long *cyc, h; int *rnd;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8); hipMalloc(&rnd, 65536 * 4 + 64);
    int *hr = (int *)malloc(65536 * 4 + 64);
    srand(7);
    for (int i = 0; i < 65536 + 16; ++i) hr[i] = (rand() << 16) ^ rand();
    hipMemcpy(rnd, hr, 65536 * 4 + 64, hipMemcpyHostToDevice);
    const int iters = 4000;
    const int modes[] = {1, 5, 6, 7, 8, 2};
    const char *names[] = {"2 x f16 + 2 x i8 32x32x32 (shipped mixed rows)", "2 x f16 + 1 x f8f6f4 K=64 FP6", "2 x f16 + 1 x f8f6f4 K=64 FP8",
                           "2 x f16 + 1 x f8f6f4 K=64 FP4", "1 x f8f6f4 K=64 FP6 alone", "2 x f16 alone"};
    const int per_unit[] = {4, 3, 3, 3, 1, 2};
    double ms_mode[6] = {};
    for (int rep = 0; rep < 3; ++rep)
        for (int mi = 0; mi < 6; ++mi) {
            const int mode = modes[mi];
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int w = 0; w < 2; ++w) {
                if (w) hipEventRecord(e0);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, rnd, out, cyc, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, rnd, out, cyc, iters);
                if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, rnd, out, cyc, iters);
                if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(256), dim3(512), 0, 0, rnd, out, cyc, iters);
                if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(256), dim3(512), 0, 0, rnd, out, cyc, iters);
                if (mode == 8) hipLaunchKernelGGL(k<8>, dim3(256), dim3(512), 0, 0, rnd, out, cyc, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const double units = 4.0 * iters;
            ms_mode[mi] = ms;
            printf("%-48s %6.1f cycles per unit and wave (%4.1f per MFMA)  %.3f ms  %.2f GHz  -> %.1f ns per unit and SIMD\n", names[mi],
                   (double)h / units, (double)h / units / per_unit[mi], ms, (double)h / (ms * 1e6), ms * 1e6 / units / 2);
        }
    printf("ratio {2 f16 + 1 FP6} / {2 f16 + 2 i8} = %.3f (kill above 0.85); FP8 %.3f; FP4 %.3f\n", ms_mode[1] / ms_mode[0], ms_mode[2] / ms_mode[0],
           ms_mode[3] / ms_mode[0]);
    return bad ? 1 : 0;
}
