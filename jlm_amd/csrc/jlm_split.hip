// Split-f16 ("f16x3") matrix kernels for gfx950.
//
// The f32 matrix pipe of CDNA4 runs at the f32 VECTOR rate (v_mfma_f32_32x32x2_f32: 64 cycles per
// instruction and SIMD, 1/16 of the f16 rate) and a kernel that keeps it >90 % busy is clocked down
// by the power governor (measured: tools/lse_profile.py).  The kernels here get f32-grade products
// out of the f16 pipe instead: every f32 operand x is carried as two halves
//     x * 2^e = hi + lo + r,   hi = f16(x 2^e),  lo = f16(x 2^e - hi),  |r| <= 2^-22 |x 2^e|
// (e = a per-matrix power of two that keeps hi inside the f16 range and lo out of the subnormals)
// and a product a.b is formed as  hi_a hi_b + hi_a lo_b + lo_a hi_b  in THREE
// v_mfma_f32_32x32x16_f16 instructions accumulating in f32: 3 x 32 cycles per 32x32x16 block instead of
// 8 x 64.  Every partial product of two f16 values is exact in f32; what is dropped
// (lo_a lo_b, hi r, r hi) is <= 3 x 2^-22 |a b| per term, the size of one f32 rounding step of the
// accumulation itself.  tests/test_gpu_kernels.py pins the error against an f64 evaluation next to
// that of the plain f32 kernels.
//
// Packed operand layout ("split rows"): a row of K values (K padded to a multiple of 16) is stored
// as K/8 blocks of 32 bytes: [8 x f16 hi][8 x f16 lo] -- the same bytes per element as f32, and one
// 16-byte granule = one MFMA operand (8 consecutive k of one plane) for the LDS-DMA.
#include "jlm_common.h"
#include "jlm_mixed_body.h"
#include <stdlib.h>

#define GLDS16(gp, lp)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp),      \
                                     (__attribute__((address_space(3))) void *)(lp), 16, 0, 0)

// LDS-DMA as a BUFFER load (buffer_load_dwordx4 ... offen lds): address = descriptor base + per-lane byte offset (one VGPR)
// + scalar byte offset (SGPR).  Same data path as global_load_lds; what differs is hipcc's wait bookkeeping: it treats
// global_load_lds (FLAT encoding) as possibly outstanding on BOTH counters and, while one is in flight -- in these
// kernels always --, turns every wait for a fragment read into s_waitcnt lgkmcnt(0) (a drain); behind a buffer load it
// counts (lgkmcnt(n), in-order LDS returns).  JLM_LSE_BUFDMA=0 keeps the FLAT form for A/B builds.
#ifndef JLM_LSE_BUFDMA
#define JLM_LSE_BUFDMA 1
#endif
#define BLDS16(rsrc, voff, soff, lp)                                                                                   \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void *)(lp), 16, (int)(voff), (int)(soff), 0, 0)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t jlm_raw_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7ffffff0, 0x00020000);
}

#define split8 jlm_split8
#ifndef JLM_LSE_PRIO
#define JLM_LSE_PRIO 0
#endif

// ---------------------------------------------------------------------------------------------
// f32 [rows, k] (ld) -> split rows (stride ld_dst 4-byte units): the blocks covering k rounded up to
// 16 values are written (zero padded), the rest of a destination row is left alone.
__global__ void pack_split_kernel(const float *__restrict__ src, int rows, int k, int ld, float scale,
                                  float *__restrict__ dst, int ld_dst) {
    const int nblk = ((k + 15) / 16) * 2;              // 8-value blocks written per row
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * nblk) return;
    const int r = (int)(i / nblk), kb = (int)(i % nblk);
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (kb * 8 + j < k) ? src[(size_t)r * ld + kb * 8 + j] : 0.0f;
    f16x8 hi, lo;
    split8(x, scale, hi, lo);
    f16x8 *d = reinterpret_cast<f16x8 *>(dst + (size_t)r * ld_dst + kb * 8);
    d[0] = hi;
    d[1] = lo;
}

extern "C" int jlm_pack_split_f16(const float *src, int rows, int k, int ld, float scale, void *dst, int ld_dst,
                                  void *stream) {
    if (rows < 0 || k <= 0 || ld < k || ld_dst % 16 || ld_dst < (k + 15) / 16 * 16) return -1;
    if (rows == 0) return 0;
    const long n = (long)rows * (((k + 15) / 16) * 2);
    hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, rows, k,
                       ld, scale, reinterpret_cast<float *>(dst), ld_dst);
    JLM_LAUNCH_CHECK();
    return 0;
}

// One column (index col) of split rows from a vector: dst[r][col] = split(v[r] * scale).
__global__ void pack_split_col_kernel(const float *__restrict__ v, int rows, float scale, float *__restrict__ dst, int ld_dst,
                                      int col) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    f16x2 h, l;
    jlm_split2(v[r], 0.0f, scale, h, l);
    _Float16 *blk = reinterpret_cast<_Float16 *>(dst + (size_t)r * ld_dst + (col & ~7)) + (col & 7);
    blk[0] = h[0];
    blk[8] = l[0];
}

extern "C" int jlm_pack_split_f16_col(const float *v, int rows, float scale, void *dst, int ld_dst, int col, void *stream) {
    if (rows < 0 || ld_dst % 16 || col < 0 || col >= ld_dst) return -1;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(pack_split_col_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, v, rows, scale,
                       reinterpret_cast<float *>(dst), ld_dst, col);
    JLM_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// k-means compressed weights (train/comp.py:52-80: per tensor a uint8 code array of the tensor's shape and a float32
// codebook): the codes stay resident in HBM, the panels the kernels consume are expanded from them ON THE DEVICE --
// dst[r][c] = codebook[code[r][c]], what `np.take(codebook, code)` does on the host in train/comp.py:70 -- with the
// codebook (<= 256 entries) in LDS.  A quarter of the upload, and the decoded float copy never exists on the host.
__global__ __launch_bounds__(256) void dequant_u8_kernel(const uint8_t *__restrict__ code, int rows, int k, int ld_code,
                                                         const float *__restrict__ book, int n_codes, float *__restrict__ dst,
                                                         int ld_dst) {
    __shared__ float lut[256];
    lut[threadIdx.x] = (int)threadIdx.x < n_codes ? book[threadIdx.x] : 0.0f;
    __syncthreads();
    const int k4 = (k + 3) >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * k4) return;
    const int r = (int)(i / k4), c = (int)(i % k4) * 4;
    const uint8_t *src = code + (size_t)r * ld_code + c;
    float *d = dst + (size_t)r * ld_dst + c;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (c + j < k) d[j] = lut[src[j]];
}

extern "C" int jlm_dequant_u8(const uint8_t *code, int rows, int k, int ld_code, const float *codebook, int n_codes, float *dst,
                              int ld_dst, void *stream) {
    if (rows < 0 || k <= 0 || ld_code < k || ld_dst < k || n_codes < 1 || n_codes > 256) return -1;
    if (rows == 0) return 0;
    const long n = (long)rows * ((k + 3) / 4);
    hipLaunchKernelGGL(dequant_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, code, rows, k, ld_code,
                       codebook, n_codes, dst, ld_dst);
    JLM_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Rows-stationary vocabulary log-sum-exp, split-f16 form.  Same structure as
// vocab_lse_stationary_kernel (jlm_gemm.hip): a workgroup keeps 128 hypothesis rows' operands in
// registers (32 rows per wave, both planes, the whole contraction) and streams a sub-range of
// vocabulary tiles (32*MT rows) past them; per row and sub-range one (max, sum exp) partial.
//
// NS = k-steps of 16 the contraction is split into (k <= 16 NS); a tile is consumed in
// NC = ceil(NS / 4) chunks of <= 4 steps (64 k-values, 256 bytes per vocabulary row), one workgroup
// barrier per chunk.  LDS tile: [32*MT rows][16 granules of 16 B]; granule g of a chunk =
// (step g >> 2, lane half (g >> 1) & 1, plane g & 1); stored at g ^ (row & 15) -- the XOR is applied
// to the SOURCE address of the lane-linear LDS-DMA -- so that the 16 lanes a ds_read_b128 serves
// together hit 16 distinct granules of the 256-byte bank line.
template <int NS>
struct SplitChunks {
    static constexpr int NC = (NS + 3) / 4;
    static constexpr int size(int c) { return NS / NC + (c < NS % NC ? 1 : 0); }
    static constexpr int start(int c) { return c * (NS / NC) + (c < NS % NC ? c : NS % NC); }
};

#define LSES_MAX_PARTS 96
#define LSES_MAX_SUB (LSES_MAX_PARTS + JLM_MAX_SEGMENTS)

// The vocabulary is cut into n_cols COLUMNS of equal cost (the launcher's cost model); a column is one workgroup per row tile.
// A column that straddles a segment boundary walks two (or more) SUB-RANGES, one per segment it touches; every sub-range
// produces its own slice of (max, sum exp) partials.
struct LseSplitArgs {
    int n_cols, n_sub, n_segs;
    jlm_segment seg[JLM_MAX_SEGMENTS];          // B = split rows, ldb in 4-byte units
    const float *bias[JLM_MAX_SEGMENTS];
    float t_scale[JLM_MAX_SEGMENTS];            // log2(e) * 2^eT: applied to T before it is split
    float descale[JLM_MAX_SEGMENTS];            // 2^-(eT + eB): accumulator -> base-2 logit
    short bias_col[JLM_MAX_SEGMENTS];           // >= 0: the segment's bias is column bias_col of its split rows
    unsigned char col_first[LSES_MAX_PARTS + 1];        // column c walks sub-ranges [col_first[c], col_first[c + 1])
    unsigned char sub_seg[LSES_MAX_SUB];
    unsigned short sub_t0[LSES_MAX_SUB], sub_t1[LSES_MAX_SUB];      // vocabulary tiles [t0, t1) of the segment
};

JLM_PROF_READER(jlm_prof_read_split)
#if defined(JLM_PROFILE) || defined(JLM_WGTIME)
// per-workgroup timeline (-DJLM_WGTIME: these stamps only, none of JLM_PROFILE's probes inside the loops) (constant 100 MHz clock): start, end, segment (of the last sub-range), shader-clock cycles in between
static __device__ unsigned long long jlm_prof_wg[1024][4];
extern "C" int jlm_prof_read_wg(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(jlm_prof_wg), sizeof(jlm_prof_wg)) == hipSuccess ? 0 : -1;
}
#endif

// -DJLM_TILETRACE: ONE wave (workgroup JLM_TT_BLOCK, wave 0) stamps the shader clock along its 4th tile: after every
// k-step's last MFMA issue, before / after the chunk barriers, around the fold (tools/probes/lse_tile_trace.py)
#ifdef JLM_TILETRACE
#ifndef JLM_TT_BLOCK
#define JLM_TT_BLOCK 8
#endif
static __device__ unsigned long long jlm_tile_trace[3][128];     // per k-step class: [0] = count, then (tag, clock) pairs
extern "C" int jlm_tile_trace_read(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(jlm_tile_trace), sizeof(jlm_tile_trace)) == hipSuccess ? 0 : -1;
}
#define JLM_TT(tag)                                                                     \
    do {                                                                                \
        if (tt_on && tt_n < 60) {                                                       \
            jlm_tile_trace[tt_cls][1 + 2 * tt_n] = (unsigned long long)(tag);           \
            jlm_tile_trace[tt_cls][2 + 2 * tt_n] = clock64();                           \
            ++tt_n;                                                                     \
            jlm_tile_trace[tt_cls][0] = tt_n;                                           \
        }                                                                               \
    } while (0)
#else
#define JLM_TT(tag) (void)0
#endif

// BG: the bias rides in the GEMM -- column sg.k of the split rows holds bias * 2^eB and the row operand
// gets 1.0 there -- so the fold is max3 / fma / exp / add per logit (the fma forms acc * 2^-(eT+eB) - max
// in one go) and stages no bias through LDS.  Needs a spare column (k % 16 != 0).
// -DJLM_ABL=<bits>: measurement builds only (tools/probes/lse_ablate.sh): 1 = no fold, 2 = no MFMAs, 4 = no fragment reads from LDS,
// 8 = no LDS-DMA.  Results are wrong by construction; the point is the time and the shader clock of what is left.
#ifndef JLM_LSE_HALF
#define JLM_LSE_HALF 1          // bias-column segments: the fold rides between the MFMAs (lse_split_body_h)
#endif
#ifndef JLM_ABL
#define JLM_ABL 0
#endif
#if JLM_ABL & 2
static __device__ __forceinline__ f32x16 lse_mfma_off(f16x8 a, f16x8 b, f32x16 c) { asm volatile("" :: "v"(a), "v"(b)); return c; }
#define LSE_MFMA(a, b, c) lse_mfma_off(a, b, c)
#else
#define LSE_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif
template <int NS, int MT, bool BG, int NW>
__device__ __forceinline__ void lse_split_body(
    const jlm_segment &sg, const float *__restrict__ bias, float t_scale, float descale, int vt0, int vt1, int pt,
    int n_paths, const float *__restrict__ T, int ldt, const int *__restrict__ rows, float2 *__restrict__ part_row,
    float *smem) {
    using CH = SplitChunks<NS>;
    constexpr int NC = CH::NC;
    constexpr int BMV = 32 * MT;                   // vocabulary rows per tile
    constexpr int NINST = BMV / (4 * NW);          // LDS-DMA instructions per wave per chunk (4 rows each, NW waves)
    constexpr float LN2 = 0.6931471805599453f, LOG2E = 1.4426950408889634f;
    JLM_PROF_DECL();
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));                 // per call: nothing derived from the lane id is hoisted over the sub-range loop
    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // an SGPR: LDS-DMA destinations (M0) stay scalar
    const int h = lane >> 5, li = lane & 31;
    const int K = sg.k, n_vocab = sg.v_end - sg.v_start, ldb = sg.ldb;
    const float *__restrict__ Bp = sg.B;
    const __amdgpu_buffer_rsrc_t rs_b = jlm_raw_rsrc(Bp);
    // 1. this lane's row operands: for step s the lane half h owns k = 16 s + 8 h .. + 7, both planes
    const int prow = pt * (32 * NW) + wave * 32 + li;
    const bool row_ok = prow < n_paths;
    const float *trow = T + (size_t)(row_ok ? (rows ? rows[prow] : prow) : 0) * ldt + sg.t_off;
    f16x8 thi[NS], tlo[NS];
    {   // loads of the next 4 steps in flight while 4 are split: bounded register use, one exposed latency
        constexpr int NG = (NS + 3) / 4;
        f32x4 xb[2][4][2];
        auto load_group = [&](int g, int par) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int k = 16 * (4 * g + j) + 8 * h + 4 * q;
                    xb[par][j][q] = *reinterpret_cast<const f32x4 *>(trow + (k < K ? k : 0));
                }
        };
        load_group(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) load_group(g + 1, (g + 1) & 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int st = 4 * g + j;
                if (st >= NS) break;
                float x[8];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = 16 * st + 8 * h + 4 * q + e;
                        x[4 * q + e] = (row_ok && k < K) ? xb[g & 1][j][q][e] : (BG && row_ok && k == K) ? 1.0f : 0.0f;
                    }
                split8(x, t_scale, thi[st], tlo[st]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float *Bs = smem;                              // [2][BMV][64]
    float *bias_s = smem + 2 * BMV * 64;           // [3][BMV], base-2 units
    // LDS-DMA: one wave instruction fills 4 tile rows x 16 granules, lane-linear.  Source of lane
    // (row r, slot) = granule g = slot ^ (r & 15) of that row.  Nothing is masked: rows past the
    // segment re-read its last row (their bias is -3e38), granules past the chunk / the padded row
    // re-read the row's last granule (no fragment read ever touches them).
    // The address of a DMA piece is a SCALAR part (segment base + tile + 4 i rows, all wave-uniform:
    // SALU work) plus a small per-lane offset (the lane's row inside the instruction's 4 rows and its
    // granule): one VALU add per piece instead of ten -- VALU issue slots are what this kernel runs
    // out of beside the MFMAs (tools/probes/mfma_valu_overlap.hip).  Only a segment's last, partial
    // tile takes the slow path that clamps the row index.
    const int lrow = lane >> 4, pslot = lane & 15;
    const int dma_g0 = pslot ^ ((wave * NINST * 4 + lrow) & 15);   // granule of piece i: dma_g0 ^ (4 (i & 3))
    const unsigned lane_row_b = (unsigned)lrow * (unsigned)ldb * 4u;
    const unsigned row4_b = 16u * (unsigned)ldb;                   // bytes between the rows of consecutive pieces
    auto issue = [&](int t, int c_start, int buf) {                // chunk = steps [c_start, ...)
        if (JLM_ABL & 8) return;
        int g0 = dma_g0;
        asm volatile("" : "+v"(g0));               // keep the per-chunk offsets out of the loop-invariant registers
        unsigned voff[4];
#pragma unroll
        for (int q = 0; q < 4 && q < NINST; ++q)
            voff[q] = lane_row_b + 4u * (unsigned)min(c_start * 16 + ((g0 ^ (4 * q)) << 2), ldb - 4);
        const int row0 = t * BMV + wave * NINST * 4;               // scalar: first row of this wave's pieces
        if (row0 + NINST * 4 <= n_vocab) {
#if JLM_LSE_BUFDMA
            const unsigned sb = (unsigned)row0 * (unsigned)ldb * 4u;                // scalar: < 2 GB per segment block (checked at launch)
#pragma unroll
            for (int i = 0; i < NINST; ++i) BLDS16(rs_b, voff[i & 3], sb + (unsigned)i * row4_b, Bs + (buf * BMV + (wave * NINST + i) * 4) * 64);
#else
            const char *sbase = reinterpret_cast<const char *>(Bp) + (size_t)row0 * ldb * 4;
#pragma unroll
            for (int i = 0; i < NINST; ++i)
                GLDS16(sbase + (size_t)i * row4_b + voff[i & 3], Bs + (buf * BMV + (wave * NINST + i) * 4) * 64);
#endif
        } else {
#pragma unroll
            for (int i = 0; i < NINST; ++i) {
                const int vrow = min(row0 + 4 * i + lrow, n_vocab - 1);
                const unsigned off = (unsigned)vrow * (unsigned)ldb * 4u + (voff[i & 3] - lane_row_b);
#if JLM_LSE_BUFDMA
                BLDS16(rs_b, off, 0, Bs + (buf * BMV + (wave * NINST + i) * 4) * 64);
#else
                GLDS16(reinterpret_cast<const char *>(Bp) + off, Bs + (buf * BMV + (wave * NINST + i) * 4) * 64);
#endif
            }
        }
    };
    auto bias_stage = [&](int t) {
        if (tid < BMV) {
            const int vrow = t * BMV + tid;
            bias_s[((t - vt0) % 3) * BMV + tid] = vrow < n_vocab ? bias[vrow] * LOG2E : JLM_NEG_BIG;
        }
    };
    float m = JLM_NEG_BIG, s = 0.0f;
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
    // fragment offsets (floats) inside a tile for step-in-chunk j, plane p: granule 4 j + 2 h + p of row li (+32 mt)
    int goff[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) goff[j][p] = li * 64 + (((4 * j + 2 * h + p) ^ (li & 15)) * 4);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    JLM_PROF_MARK(p_t1);
    issue(vt0, 0, 0);
    if (!BG) bias_stage(vt0);
    __syncthreads();
    JLM_PROF_MARK(p_t2);
    int buf = 0;
#if JLM_LSE_PRIO == 2
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
#ifdef JLM_TILETRACE
    int tt_n = 0;
    const int tt_cls = NS <= 4 ? 2 : (NS <= 7 ? 1 : 0);
    // the first sub-range of each segment, row tile 0
    const bool tt_wg = (vt0 == 0 && pt == 0 && wave == 0);
#endif
    for (int t = vt0; t < vt1; ++t) {
#ifdef JLM_TILETRACE
        const bool tt_on = tt_wg && t == vt0 + 3;
#endif
        JLM_TT(1);
#if JLM_LSE_PRIO == 1
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool last_c = (c == NC - 1);
            if (last_c) issue(t + 1, 0, buf ^ 1);                    // past the range: harmless (next range / last row)
            else issue(t, CH::start(c + 1 < NC ? c + 1 : 0), buf ^ 1);
            const float *bs = Bs + buf * BMV * 64;
            // One k-step = 3 groups of MT MFMAs (lo.hi, hi.lo, hi.hi over all MT blocks): consecutive
            // MFMAs never share an accumulator and a dependent one is MT instructions away.  Fragment
            // reads run ahead of their use: both planes of step j+1 (lo refilled in place, hi double-
            // buffered) are requested right after step j's first group, 8 MFMAs before they are needed.
            // The sched_barriers pin that order: left alone hipcc shares one register set between the
            // planes and moves every read directly in front of its MFMAs (s_waitcnt lgkmcnt(0) before
            // each group: three exposed LDS round trips per k-step, half the kernel's MFMA time).
            f16x8 ah[2][MT], al[MT];
            const int csz = CH::size(c), cst = CH::start(c);
            auto load_plane = [&](f16x8 (&dst)[MT], int j, int p) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (JLM_ABL & 4) { asm volatile("" : "=v"(dst[mt])); continue; }
                    dst[mt] = *reinterpret_cast<const f16x8 *>(bs + mt * 32 * 64 + goff[j][p]);
                }
            };
            // (steps past the segment's own k, when NS is rounded up, multiply zero T operands: no
            //  run-time guards in here, they would split the chunk into basic blocks)
            load_plane(al, 0, 1);
            load_plane(ah[0], 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j >= csz) break;               // compile-time bound of this chunk
                const int st = cst + j;            // compile-time after unrolling (register index)
                const bool more = (j + 1 < csz);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = LSE_MFMA(al[mt], thi[st], (c == 0 && j == 0) ? zero16 : acc[mt]);
                if (j + 1 < 4 && more) {           // both planes of step j+1 (lo refilled in place: behind the lo.hi group)
                    load_plane(al, j + 1 < 4 ? j + 1 : 0, 1);
                    load_plane(ah[(j + 1) & 1], j + 1 < 4 ? j + 1 : 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = LSE_MFMA(ah[j & 1][mt], tlo[st], acc[mt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = LSE_MFMA(ah[j & 1][mt], thi[st], acc[mt]);
                // issue order: the MT lo.hi MFMAs, then ONE fragment read behind every following MFMA (tools/probes/gate_loop.hip:
                // reads grouped in front of an MFMA group cost 12 % of the k-step, interleaved 1:1 they are free)
                __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);
                if (j + 1 < 4 && more) {
#pragma unroll
                    for (int i = 0; i < 2 * MT; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * MT, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                JLM_TT(10 + st);
            }
            if (!BG && last_c) bias_stage(t + 1);
            JLM_PROF_MARK(p_x);
            JLM_TT(2);
#ifdef JLM_TILETRACE
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the next chunk's DMA has landed ...
            JLM_TT(5);                                                      // ... the rest of the wait is the other waves
#endif
            __syncthreads();
            JLM_TT(3);
            JLM_PROF_ADD(p_bar, p_x);
            buf ^= 1;
        }
#if JLM_LSE_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#endif
        JLM_PROF_MARK(p_x);
        // fold this tile's 16*MT logits of the lane's row into (m, s), base-2 units
        if (JLM_ABL & 1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" :: "v"(acc[mt]));     // every accumulator stays live
            s += acc[0][0];
        } else if (BG) {
            if ((t + 1) * BMV > n_vocab) {         // the segment's last, partial tile: rows past it re-read its last row
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * BMV + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= n_vocab) acc[mt][r] = JLM_NEG_BIG;
            }
            float tmax = JLM_NEG_BIG;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; r += 2) tmax = fmaxf(fmaxf(tmax, acc[mt][r]), acc[mt][r + 1]);
            const float mn = fmaxf(m, tmax * descale);         // descale > 0
            const float nmn = -mn;
            float add0 = 0.0f, add1 = 0.0f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    add0 += __builtin_amdgcn_exp2f(fmaf(acc[mt][r], descale, nmn));
                    add1 += __builtin_amdgcn_exp2f(fmaf(acc[mt][r + 1], descale, nmn));
                }
            s = s * __builtin_amdgcn_exp2f(m - mn) + (add0 + add1);
            m = mn;
        } else {
            const float *bt = bias_s + ((t - vt0) % 3) * BMV + 4 * h;
            float tmax = JLM_NEG_BIG;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bt + mt * 32 + 8 * j);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = fmaf(acc[mt][4 * j + e], descale, b4[e]);
                        acc[mt][4 * j + e] = v;
                        tmax = fmaxf(tmax, v);
                    }
                }
            const float mn = fmaxf(m, tmax);
            float add0 = 0.0f, add1 = 0.0f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    add0 += __builtin_amdgcn_exp2f(acc[mt][r] - mn);
                    add1 += __builtin_amdgcn_exp2f(acc[mt][r + 1] - mn);
                }
            s = s * __builtin_amdgcn_exp2f(m - mn) + (add0 + add1);
            m = mn;
        }
        JLM_PROF_ADD(p_fold, p_x);
        JLM_TT(4);
    }
    JLM_PROF_FLUSH();
    const float m2 = __shfl_xor(m, 32), s2 = __shfl_xor(s, 32);
    {
        const float mm = fmaxf(m, m2);
        s = s * __builtin_amdgcn_exp2f(m - mm) + s2 * __builtin_amdgcn_exp2f(m2 - mm);
        m = mm * LN2;
    }
    if (h == 0 && row_ok) part_row[prow] = make_float2(m, s);
}

// The same with the fold INSIDE the MFMA stream (bias-column segments).  A tile's MT blocks are walked as two halves:
// within a chunk first every k-step of blocks [0, MT/2), then every k-step of blocks [MT/2, MT).  Once a tile's last chunk
// has finished the first half, those accumulators are final: their fold (max3 / fma / exp / add per logit) is issued between
// the MFMAs of the second half; the second half's fold rides between the first-half MFMAs of the NEXT tile's first chunk.
// Same registers, same LDS layout, same DMA as lse_split_body -- only the order differs.  Why: folding after the tile, all
// eight waves at once, leaves the matrix pipe idle for 1 250 cycles per tile (19.5 cycles per logit and wave); between a
// wave's own MFMAs the same VALU work costs 4.4 SIMD cycles per logit (tools/probes/mfma_valu_inwave.hip).
template <int NS, int MT, int NW>
__device__ __forceinline__ void lse_split_body_h(
    const jlm_segment &sg, float t_scale, float descale, int vt0, int vt1, int pt, int n_paths, const float *__restrict__ T, int ldt,
    const int *__restrict__ rows, float2 *__restrict__ part_row, float *smem) {
    using CH = SplitChunks<NS>;
    constexpr int NC = CH::NC;
    constexpr int HM = MT / 2;                     // blocks per half
    constexpr int BMV = 32 * MT;                   // vocabulary rows per tile
    constexpr int NINST = BMV / (4 * NW);          // LDS-DMA instructions per wave per chunk (4 rows each, NW waves)
    constexpr int NE = 16 * HM;                    // logits per lane and half
    constexpr float LN2 = 0.6931471805599453f;
    static_assert(MT % 2 == 0, "two halves");
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));                 // per call: nothing derived from the lane id is hoisted over the sub-range loop
    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // an SGPR: LDS-DMA destinations (M0) stay scalar
    const int h = lane >> 5, li = lane & 31;
    const int K = sg.k, n_vocab = sg.v_end - sg.v_start, ldb = sg.ldb;
    const float *__restrict__ Bp = sg.B;
    const __amdgpu_buffer_rsrc_t rs_b = jlm_raw_rsrc(Bp);
    const int prow = pt * (32 * NW) + wave * 32 + li;
    const bool row_ok = prow < n_paths;
    const float *trow = T + (size_t)(row_ok ? (rows ? rows[prow] : prow) : 0) * ldt + sg.t_off;
    f16x8 thi[NS], tlo[NS];
    {   // the lane's row operands (the constant 1.0 of the bias column at k == K).  Every load of the row is requested before
        // any is used -- one memory round trip; in groups of four k-steps (lse_split_body) the prologue was four of them, 5 us
        // of an 85-us workgroup -- the accumulators are not live yet, the registers are there (16-step form: two groups of 8)
        constexpr int GS = NS <= 13 ? NS : 8;
        constexpr int NG = (NS + GS - 1) / GS;
        f32x4 xb[NG > 1 ? 2 : 1][GS][2];
        auto load_group = [&](int g, int par) {
#pragma unroll
            for (int j = 0; j < GS; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int k = 16 * (GS * g + j) + 8 * h + 4 * q;
                    xb[par][j][q] = *reinterpret_cast<const f32x4 *>(trow + (k < K ? k : 0));
                }
        };
        load_group(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) load_group(g + 1, (g + 1) & 1);
#pragma unroll
            for (int j = 0; j < GS; ++j) {
                const int st = GS * g + j;
                if (st >= NS) break;
                float x[8];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = 16 * st + 8 * h + 4 * q + e;
                        x[4 * q + e] = (row_ok && k < K) ? xb[NG > 1 ? (g & 1) : 0][j][q][e] : (row_ok && k == K) ? 1.0f : 0.0f;
                    }
                split8(x, t_scale, thi[st], tlo[st]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float *Bs = smem;                              // [2][BMV][64]
    const int lrow = lane >> 4, pslot = lane & 15;
    const int dma_g0 = pslot ^ ((wave * NINST * 4 + lrow) & 15);
    const unsigned lane_row_b = (unsigned)lrow * (unsigned)ldb * 4u;
    const unsigned row4_b = 16u * (unsigned)ldb;
    auto issue = [&](int t, int c_start, int buf) {                // as in lse_split_body
        int g0 = dma_g0;
        asm volatile("" : "+v"(g0));
        unsigned voff[4];
#pragma unroll
        for (int q = 0; q < 4 && q < NINST; ++q)
            voff[q] = lane_row_b + 4u * (unsigned)min(c_start * 16 + ((g0 ^ (4 * q)) << 2), ldb - 4);
        const int row0 = t * BMV + wave * NINST * 4;
        if (row0 + NINST * 4 <= n_vocab) {
#if JLM_LSE_BUFDMA
            const unsigned sb = (unsigned)row0 * (unsigned)ldb * 4u;                // scalar: < 2 GB per segment block (checked at launch)
#pragma unroll
            for (int i = 0; i < NINST; ++i) BLDS16(rs_b, voff[i & 3], sb + (unsigned)i * row4_b, Bs + (buf * BMV + (wave * NINST + i) * 4) * 64);
#else
            const char *sbase = reinterpret_cast<const char *>(Bp) + (size_t)row0 * ldb * 4;
#pragma unroll
            for (int i = 0; i < NINST; ++i)
                GLDS16(sbase + (size_t)i * row4_b + voff[i & 3], Bs + (buf * BMV + (wave * NINST + i) * 4) * 64);
#endif
        } else {
#pragma unroll
            for (int i = 0; i < NINST; ++i) {
                const int vrow = min(row0 + 4 * i + lrow, n_vocab - 1);
                const unsigned off = (unsigned)vrow * (unsigned)ldb * 4u + (voff[i & 3] - lane_row_b);
#if JLM_LSE_BUFDMA
                BLDS16(rs_b, off, 0, Bs + (buf * BMV + (wave * NINST + i) * 4) * 64);
#else
                GLDS16(reinterpret_cast<const char *>(Bp) + off, Bs + (buf * BMV + (wave * NINST + i) * 4) * 64);
#endif
            }
        }
    };
    float m = JLM_NEG_BIG, s = 0.0f;
    f32x16 acc[MT];
    // the first tile's first half folds "the previous tile's second half": finite, far below any logit -- whatever it leaves in
    // (m, s) is scaled by 2^(m - max) = 0 at the first real fold
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = -1.0e30f;
    int goff[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) goff[j][p] = li * 64 + (((4 * j + 2 * h + p) ^ (li & 15)) * 4);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // rows of the segment's last, partial tile that lie past its end re-read the last row: their logits are struck out
    auto mask_half = [&](int hf, int t) {
        if ((t + 1) * BMV > n_vocab) {
#pragma unroll
            for (int q = 0; q < HM; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * BMV + (hf * HM + q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= n_vocab) acc[hf * HM + q][r] = JLM_NEG_BIG;
        }
    };
    // the fold of one half, cut into `parts` slices that are issued between the MFMAs of `parts` consecutive k-steps:
    // slice 0 also finds the half's maximum and the new running maximum, the last slice closes the running sum
    float f_nmn = 0.0f, f_sc = 0.0f, f_a0 = 0.0f, f_a1 = 0.0f;
    auto fold_slice = [&](int hf, int i, int parts) {
        if (i == 0) {
            float tmax = JLM_NEG_BIG;
#pragma unroll
            for (int q = 0; q < HM; ++q)
#pragma unroll
                for (int r = 0; r < 16; r += 2) tmax = fmaxf(fmaxf(tmax, acc[hf * HM + q][r]), acc[hf * HM + q][r + 1]);
            const float mn = fmaxf(m, tmax * descale);             // descale > 0
            f_sc = __builtin_amdgcn_exp2f(m - mn);
            f_nmn = -mn;
            f_a0 = f_a1 = 0.0f;
            m = mn;
        }
        const int e0 = NE * i / parts, e1 = NE * (i + 1) / parts;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (e < e0 || e >= e1) continue;
            const float x = __builtin_amdgcn_exp2f(fmaf(acc[hf * HM + (e >> 4)][e & 15], descale, f_nmn));
            if (e & 1) f_a1 += x; else f_a0 += x;
        }
        if (i == parts - 1) s = s * f_sc + (f_a0 + f_a1);
    };
    issue(vt0, 0, 0);
    __syncthreads();
    int buf = 0;
    for (int t = vt0; t < vt1; ++t) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool last_c = (c == NC - 1);
            if (last_c) issue(t + 1, 0, buf ^ 1);                    // past the range: harmless (next range / last row)
            else issue(t, CH::start(c + 1 < NC ? c + 1 : 0), buf ^ 1);
            const float *bs = Bs + buf * BMV * 64;
            f16x8 ah[2][HM], al[HM];
            const int csz = CH::size(c), cst = CH::start(c);
            auto load_plane = [&](f16x8 (&dst)[HM], int hf, int j, int p) {
#pragma unroll
                for (int q = 0; q < HM; ++q) dst[q] = *reinterpret_cast<const f16x8 *>(bs + (hf * HM + q) * 32 * 64 + goff[j][p]);
            };
            // stages of the chunk: (half 0, steps 0 .. csz-1), (half 1, steps 0 .. csz-1); the fragments of stage i+1 are
            // requested behind the first MFMA group of stage i
            load_plane(al, 0, 0, 1);
            load_plane(ah[0], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i >= 2 * csz) break;                       // compile-time bound
                const int hf = i >= csz ? 1 : 0, j = hf ? i - csz : i;
                const int st = cst + j;
                const bool more = (i + 1 < 2 * csz);
                if (hf == 1 && j == 0 && last_c) mask_half(0, t);   // (uniform, rare) the first half is final here
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < HM; ++q)
                    acc[hf * HM + q] = LSE_MFMA(al[q], thi[st], (c == 0 && j == 0) ? zero16 : acc[hf * HM + q]);
                if (more) {
                    const int hn = (i + 1) >= csz ? 1 : 0, jn = hn ? i + 1 - csz : i + 1;
                    load_plane(al, hn, jn, 1);
                    load_plane(ah[(i + 1) & 1], hn, jn, 0);
                }
#pragma unroll
                for (int q = 0; q < HM; ++q) acc[hf * HM + q] = LSE_MFMA(ah[i & 1][q], tlo[st], acc[hf * HM + q]);
#pragma unroll
                for (int q = 0; q < HM; ++q) acc[hf * HM + q] = LSE_MFMA(ah[i & 1][q], thi[st], acc[hf * HM + q]);
                // the fold slice that rides in this stage
                const bool f_prev = (c == 0 && hf == 0), f_cur = (last_c && hf == 1);
                if (f_prev) fold_slice(1, j, csz);
                if (f_cur) fold_slice(0, j, csz);
                // issue order: one fragment read and a share of the fold behind every MFMA
                constexpr int VPM = (3 * NE + 24) / 3 / (3 * HM) + 1;      // VALU per MFMA when a 3-step chunk carries a fold
                if (f_prev || f_cur) {
#pragma unroll
                    for (int g = 0; g < HM; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                    }
#pragma unroll
                    for (int g = 0; g < 2 * HM; ++g) {
                        if (more) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                    }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, HM, 0);
                    if (more) {
#pragma unroll
                        for (int g = 0; g < 2 * HM; ++g) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        }
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2 * HM, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            buf ^= 1;
        }
        mask_half(1, t);                                           // (uniform, rare: the range's last tile at most)
    }
    // the last tile's second half
    if (vt1 > vt0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fold_slice(1, i, 2);
    }
    const float m2 = __shfl_xor(m, 32), s2 = __shfl_xor(s, 32);
    {
        const float mm = fmaxf(m, m2);
        s = s * __builtin_amdgcn_exp2f(m - mm) + s2 * __builtin_amdgcn_exp2f(m2 - mm);
        m = mm * LN2;
    }
    if (h == 0 && row_ok) part_row[prow] = make_float2(m, s);
}

template <int NW>
__device__ __forceinline__ void vocab_lse_split_main(const LseSplitArgs &a, const float *__restrict__ T, int ldt,
                                                     const int *__restrict__ rows, float2 *__restrict__ part, int ld_part,
                                                     int n_rows_max, const int *n_dev, int n_ptiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n_paths = n_dev ? min(*n_dev, n_rows_max) : n_rows_max;
    // XCD-aware order, as in vocab_lse_stationary_kernel: with n_parts a multiple of 8 one XCD walks
    // all row tiles of its vocabulary columns, which stay in its L2
    const int b = blockIdx.x;
    int p, pt;
    // n_cols = 8 q + r: the first 8 q columns are dealt to the XCDs (block b runs on XCD b % 8), column p on XCD p % 8 with
    // all of its row tiles; the r < 8 columns left over follow linearly -- their few workgroups land one or two per XCD, on the
    // CUs the 8 q columns leave idle (10 row tiles: 3 x 10 = 30 of an XCD's 32 CUs)
    const int nb8 = (a.n_cols & ~7) * n_ptiles;
    if (b < nb8) { const int x = b & 7, jb = b >> 3; p = (jb / n_ptiles) * 8 + x; pt = jb % n_ptiles; }
    else { const int bb = b - nb8; p = (a.n_cols & ~7) + bb / n_ptiles; pt = bb % n_ptiles; }
    if (p >= a.n_cols || pt * (32 * NW) >= n_paths) return;
#if defined(JLM_PROFILE) || defined(JLM_WGTIME)
    const unsigned long long wg_t0 = wall_clock64(), wg_c0 = clock64();
    int si_last = 0;
#endif
#define JLM_LSE_CASE(NS_, MT_)                                                                                           \
    do {                                                                                                                 \
        if (bg && JLM_LSE_HALF) lse_split_body_h<NS_, MT_, NW>(sg, ts, ds, vt0, vt1, pt, n_paths, T, ldt, rows, prow, smem);  \
        else if (bg) lse_split_body<NS_, MT_, true, NW>(sg, bias, ts, ds, vt0, vt1, pt, n_paths, T, ldt, rows, prow, smem); \
        else lse_split_body<NS_, MT_, false, NW>(sg, bias, ts, ds, vt0, vt1, pt, n_paths, T, ldt, rows, prow, smem);     \
    } while (0)
    for (int r = a.col_first[p]; r < a.col_first[p + 1]; ++r) {
        const int si = a.sub_seg[r];
        const jlm_segment sg = a.seg[si];
        const float *bias = a.bias[si];
        const float ts = a.t_scale[si], ds = a.descale[si];
        const int vt0 = a.sub_t0[r], vt1 = a.sub_t1[r];
        float2 *prow = part + (size_t)r * ld_part;
        const int ns = (sg.k + 15) >> 4;
        const bool bg = a.bias_col[si] >= 0;
        if (r != a.col_first[p]) __syncthreads();          // the previous sub-range's last fold may still read its staged biases
        if (ns <= 2) JLM_LSE_CASE(2, 4);
        else if (ns <= 4) JLM_LSE_CASE(4, 4);
        else if (ns <= 7) JLM_LSE_CASE(7, 4);
        else if (ns <= 10) JLM_LSE_CASE(10, 4);
        else if (ns <= 13) JLM_LSE_CASE(13, 4);
        else JLM_LSE_CASE(16, 2);
#if defined(JLM_PROFILE) || defined(JLM_WGTIME)
        si_last = si;
#endif
    }
#undef JLM_LSE_CASE
#if defined(JLM_PROFILE) || defined(JLM_WGTIME)
    if (threadIdx.x == 0 && b < 1024) {
        jlm_prof_wg[b][0] = wg_t0; jlm_prof_wg[b][1] = wall_clock64(); jlm_prof_wg[b][2] = si_last; jlm_prof_wg[b][3] = clock64() - wg_c0;
    }
#endif
}

// 4 waves x 32 rows, two workgroups per CU
__global__ __launch_bounds__(256, 2) void vocab_lse_split_kernel(LseSplitArgs a, const float *__restrict__ T, int ldt,
                                                                  const int *__restrict__ rows, float2 *__restrict__ part,
                                                                  int ld_part, int n_rows_max, const int *n_dev, int n_ptiles) {
    vocab_lse_split_main<4>(a, T, ldt, rows, part, ld_part, n_rows_max, n_dev, n_ptiles);
}
// 8 waves x 32 rows, one workgroup per CU: a staged vocabulary chunk serves 256 rows (half the L2 -> LDS
// traffic and half the DMA issues per MFMA)
__global__ __launch_bounds__(512, 1) void vocab_lse_split8_kernel(LseSplitArgs a, const float *__restrict__ T, int ldt,
                                                                   const int *__restrict__ rows, float2 *__restrict__ part,
                                                                   int ld_part, int n_rows_max, const int *n_dev, int n_ptiles) {
    vocab_lse_split_main<8>(a, T, ldt, rows, part, ld_part, n_rows_max, n_dev, n_ptiles);
}

// Host side: equal-cost columns over the concatenated segments, one resident round of workgroups.
// Returns the number of partial slices written (fold them with jlm_lse_combine), or <0.
extern "C" int jlm_vocab_lse_split(const jlm_segment *segs_host, const float *t_scale, const float *descale,
                                   const int *bias_col, int n_segs, const float *b2, const float *T, int ldt, const int *rows,
                                   float *part, int ld_part, int max_parts, int n_rows_max, const int *n_dev, void *stream) {
    if (n_segs < 1 || n_segs > JLM_MAX_SEGMENTS || ldt % 4 || n_rows_max <= 0) return -1;
    LseSplitArgs a;
    a.n_segs = n_segs;
    static int c0x2 = -1, np8 = -1;            // cost constant in half k-steps; JLM_LSE_NP8=1: column count a multiple of 8
    if (c0x2 < 0) { const char *e = getenv("JLM_LSE_C0"); c0x2 = e ? (int)(2.0 * atof(e) + 0.5) : 2; }
    // JLM_LSE_NP8=0 lets the column count use every CU (25 columns x 10 row tiles = 250 workgroups instead of 240: the kernel
    // alone runs 3 % faster), but the 16 CUs the multiple of 8 leaves idle are where the other batch in flight runs its
    // small kernels meanwhile: the decode is 2.8 % slower with them taken (2.59 vs 2.52 ms per step, tools/ab_engine.py)
    if (np8 < 0) { const char *e = getenv("JLM_LSE_NP8"); np8 = e ? atoi(e) : 1; }
    int ntiles[JLM_MAX_SEGMENTS];
    double ctile[JLM_MAX_SEGMENTS], total = 0.0;
    long n_tiles_all = 0;
    for (int i = 0; i < n_segs; ++i) {
        const jlm_segment &sg = segs_host[i];
        const int ns = (sg.k + 15) / 16;
        if (ns > 16 || sg.k % 4 || sg.ldb % 16 || sg.ldb < ns * 16 || sg.t_off % 4) return -2;
        const int bc = bias_col ? bias_col[i] : -1;
        if (bc >= 0 && (bc != sg.k || sg.k % 16 == 0)) return -2;          // the bias column is the first padded one
        a.bias_col[i] = (short)bc;
        a.seg[i] = sg;
        a.bias[i] = b2 + sg.v_start;
        a.t_scale[i] = t_scale[i] * 1.4426950408889634f;
        a.descale[i] = descale[i];
        const int bmv = ns > 13 ? 64 : 128;
        ntiles[i] = (sg.v_end - sg.v_start + bmv - 1) / bmv;
        if (ntiles[i] > 65535) return -2;
        // Cost of a vocabulary tile ~ (k-steps + c0): the MFMAs plus a per-tile constant (barrier, what the fold costs between
        // the MFMAs).  The per-workgroup timeline of the mixed launch (tools/probes/lse_wg_timeline.py on a -DJLM_WGTIME
        // build) gives 8.15 / 4.62 / 2.95 us per 128-word tile at 13 / 7 / 4 k-steps = 0.58 (k-steps + 1.1); a sweep of the
        // constant (JLM_LSE_C0 = 0.5 .. 2) is flat within the noise between 1 and 2.
        ctile[i] = (2 * ns + c0x2) * (bmv / 128.0);
        total += ctile[i] * ntiles[i];
        n_tiles_all += ntiles[i];
    }
    static int nw = -1;
    if (nw < 0) { const char *e = getenv("JLM_LSE_WAVES"); nw = (e && atoi(e) == 4) ? 4 : 8; }
    const int n_ptiles = (n_rows_max + 32 * nw - 1) / (32 * nw);
    int cap = max_parts < LSES_MAX_SUB ? max_parts : LSES_MAX_SUB;
    cap -= n_segs - 1;                          // slices: one per column + one per segment boundary a column straddles
    if (cap > LSES_MAX_PARTS) cap = LSES_MAX_PARTS;
    if (cap < 1) return -1;
    int np = ((nw == 8 ? 1 : 2) * 256) / n_ptiles; // one resident round: 2 four-wave / 1 eight-wave workgroup per CU
    if (np < 1) np = 1;
    if (np > cap) np = cap;
    if (np >= 8 && np8) np &= ~7;
    // JLM_LSE_NP=<columns> for launches that would use >= 8 (read on every call: tools/ab_np.py changes it in-process)
    int np_force = 0;
    { const char *e = getenv("JLM_LSE_NP"); np_force = e ? atoi(e) : 0; }
    if (np_force > 0 && np_force <= cap) np = np_force;
    if (np > n_tiles_all) np = (int)n_tiles_all;
    // Min-max cuts of the concatenated segments: the smallest column budget M (bisection) with which a greedy fill -- a
    // column takes whole tiles while they fit; at a segment's end it goes on in the next one when what is left pays for the
    // second start (the last fold, the T operands of the other segment, the first chunk's exposed latency: ~9 us = 17
    // k-steps in the timeline, JLM_LSE_PRO) and one tile -- needs <= np columns.  A column is one sub-range per segment
    // it touches.
    static double pro = -1.0;
    if (pro < 0) { const char *e = getenv("JLM_LSE_PRO"); pro = 2.0 * (e ? atof(e) : 17.0); }
    int n_sub = 0, n_cols = 0;
    auto fill = [&](double M, bool emit) -> int {
        int seg = 0, t = 0, cols = 0;
        n_sub = 0;
        while (seg < n_segs) {
            if (emit) { if (cols >= LSES_MAX_PARTS) return -1; a.col_first[cols] = (unsigned char)n_sub; }
            double budget = M;
            bool first = true;
            while (seg < n_segs) {
                if (!first) {
                    if (budget < pro + ctile[seg]) break;
                    budget -= pro;
                }
                const int avail = ntiles[seg] - t;
                int take = (int)(budget / ctile[seg] + 1e-9);
                if (take > avail) take = avail;
                if (take < 1) { if (!first) break; take = 1; }
                if (emit) {
                    if (n_sub >= LSES_MAX_SUB) return -1;
                    a.sub_seg[n_sub] = (unsigned char)seg;
                    a.sub_t0[n_sub] = (unsigned short)t;
                    a.sub_t1[n_sub] = (unsigned short)(t + take);
                }
                ++n_sub;
                budget -= take * ctile[seg];
                first = false;
                t += take;
                if (t < ntiles[seg]) break;
                ++seg;
                t = 0;
            }
            ++cols;
        }
        if (emit) a.col_first[cols] = (unsigned char)n_sub;
        return cols;
    };
    {
        double cmax = 0.0;
        for (int i = 0; i < n_segs; ++i) cmax = ctile[i] > cmax ? ctile[i] : cmax;
        double lo = total / np, hi = total / np + (2.0 * cmax + pro) * n_segs + 1.0;
        for (int it = 0; it < 32; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (fill(mid, false) <= np) hi = mid; else lo = mid;
        }
        n_cols = fill(hi, true);
        if (n_cols < 1 || n_cols > np) return -4;
    }
    np = n_cols;
    a.n_cols = np;
    a.n_sub = n_sub;
    const int given = n_sub;
    if (given > max_parts) return -1;
    const int lds = (2 * 128 * 64 + 3 * 128) * 4;
    static JlmLdsGrant grant4, grant8;
    if (int rc = jlm_grant_lds(grant4, reinterpret_cast<const void *>(vocab_lse_split_kernel), lds)) return rc;
    if (int rc = jlm_grant_lds(grant8, reinterpret_cast<const void *>(vocab_lse_split8_kernel), lds)) return rc;
    const int grid = np * n_ptiles;
    if (nw == 8)
        hipLaunchKernelGGL(vocab_lse_split8_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, a, T, ldt, rows,
                           reinterpret_cast<float2 *>(part), ld_part, n_rows_max, n_dev, n_ptiles);
    else
        hipLaunchKernelGGL(vocab_lse_split_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a, T, ldt, rows,
                           reinterpret_cast<float2 *>(part), ld_part, n_rows_max, n_dev, n_ptiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(int)e - 100;
    return given;
}

// ---------------------------------------------------------------------------------------------
// HYBRID: one launch whose columns walk segments of BOTH formats -- mixed rows (f16 hi.hi + int8 cross terms, jlm_mixed.hip) where
// the contraction is long enough for the matrix instructions to dominate (k = 200, 100: 27 / 15 instead of 39 / 21 instructions
// per 32 x 32 block), split rows with three f16 passes where the fold's VALU work dominates either way (k = 50: 8 vs 12
// instructions against ~90 VALU instructions of fold per block -- the int accumulator's combine makes the mixed form SLOWER there:
// 30.2 vs 27.2 us for the segment alone).  Same equal-cost columns, same partial slices.
struct LseHybridArgs {
    LseSplitArgs sp;                                 // the split view of every segment (unused fields for mixed ones)
    jlm_mx::MxSeg mx[JLM_MAX_SEGMENTS];
    // bit i: segment i runs on its mixed rows.  A scalar, NOT a byte array indexed by the segment: with `unsigned char
    // is_mixed[8]` hipcc based every per-segment kernarg load on &is_mixed[si] (s_load_dwordx2 s[80:81], s[8:9], s4 with
    // s[8:9] = kernarg + si, s4 = 31 si): base and offset each unaligned, their sum aligned -- the scalar unit drops the low
    // bits of the parts, segment 2's matrix pointer came out of the wrong dwords and the LDS-DMA faulted
    unsigned mixed_mask;
};

__global__ __launch_bounds__(512, 1) void vocab_lse_hybrid_kernel(LseHybridArgs a, const float *__restrict__ T, int ldt, const float *__restrict__ Tm,
                                                                  int ld_tm, const int *__restrict__ rows, float2 *__restrict__ part, int ld_part,
                                                                  int n_rows_max, const int *n_dev, int n_ptiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n_paths = n_dev ? min(*n_dev, n_rows_max) : n_rows_max;
    const int b = blockIdx.x;
    int p, pt;
    const int nb8 = (a.sp.n_cols & ~7) * n_ptiles;
    if (b < nb8) { const int x = b & 7, jb = b >> 3; p = (jb / n_ptiles) * 8 + x; pt = jb % n_ptiles; }
    else { const int bb = b - nb8; p = (a.sp.n_cols & ~7) + bb / n_ptiles; pt = bb % n_ptiles; }
    if (p >= a.sp.n_cols || pt * 256 >= n_paths) return;
    for (int r = a.sp.col_first[p]; r < a.sp.col_first[p + 1]; ++r) {
        const int si = a.sp.sub_seg[r];
        const int vt0 = a.sp.sub_t0[r], vt1 = a.sp.sub_t1[r];
        float2 *prow = part + (size_t)r * ld_part;
        if (r != a.sp.col_first[p]) __syncthreads();
        if ((a.mixed_mask >> si) & 1) {
            const jlm_mx::MxSeg sg = a.mx[si];
            const int ns16 = (sg.k + 2 + 15) >> 4;
            unsigned char *sm8 = reinterpret_cast<unsigned char *>(smem);
            if (sg.nb == 7 && ns16 == 13) jlm_mx::mx_body<7, 13, 2>(sg, vt0, vt1, pt, n_paths, Tm, ld_tm, nullptr, prow, sm8);
            else if (sg.nb == 4 && ns16 == 7) jlm_mx::mx_body<4, 7, 4>(sg, vt0, vt1, pt, n_paths, Tm, ld_tm, nullptr, prow, sm8);      // (packed rows: compact)
            else if (sg.nb == 2 && ns16 == 4) jlm_mx::mx_body<2, 4, 8>(sg, vt0, vt1, pt, n_paths, Tm, ld_tm, nullptr, prow, sm8);
        } else {
            const jlm_segment sg = a.sp.seg[si];
            const int ns = (sg.k + 15) >> 4;
            if (ns <= 2) lse_split_body_h<2, 4, 8>(sg, a.sp.t_scale[si], a.sp.descale[si], vt0, vt1, pt, n_paths, T, ldt, rows, prow, smem);
            else if (ns <= 4) lse_split_body_h<4, 4, 8>(sg, a.sp.t_scale[si], a.sp.descale[si], vt0, vt1, pt, n_paths, T, ldt, rows, prow, smem);
            else if (ns <= 7) lse_split_body_h<7, 4, 8>(sg, a.sp.t_scale[si], a.sp.descale[si], vt0, vt1, pt, n_paths, T, ldt, rows, prow, smem);
            else lse_split_body_h<13, 4, 8>(sg, a.sp.t_scale[si], a.sp.descale[si], vt0, vt1, pt, n_paths, T, ldt, rows, prow, smem);       // (round 5: a segment's head)
        }
    }
}

// segs / t_scale / descale / bias_col: as for jlm_vocab_lse_split, for EVERY segment (the T column offsets of the mixed ones are
// read from here too).  mixed[i].B != NULL: segment i runs on its mixed rows (mixed[i].ldb = 32 nb; mx_descale / mx_s8 as for
// jlm_vocab_lse_mixed), else on its split rows.  -2: a shape this kernel does not host (mixed: k + 2 in (192, 208], (96, 112] or
// (32, 64]; split: k <= 208 and no multiple of 16, with a bias column) -- the caller falls back to jlm_vocab_lse_split.
// ABI 10, head_split (NULL: none): the first head_split[i] words (a multiple of 128) of MIXED segment i run on its split rows -- the
// words that carry a trained model's probability mass, and so the int8 cross terms' share of the log-normaliser's error -- the rest
// on mixed rows.  The segment becomes two internal ones (same T columns, same packed hypothesis rows, same scale slot).
extern "C" int jlm_vocab_lse_hybrid(const jlm_segment *segs_host_in, const float *t_scale_in, const float *descale_in, const int *bias_col_in,
                                    const jlm_segment *mixed_in, const float *mx_descale_in, const float *mx_s8_in, const int *head_split,
                                    int n_segs, const float *b2,
                                    const float *T, int ldt, const void *Tm, int ld_tm, const int *rows, float *part, int ld_part,
                                    int max_parts, int n_rows_max, const int *n_dev, void *stream) {
    if (n_segs < 1 || n_segs > JLM_MAX_SEGMENTS || ldt % 4 || n_rows_max <= 0) return -1;
    // internal segments: a mixed segment with a head is listed twice -- its head as a split segment, the rest as a mixed one
    jlm_segment segs_host[JLM_MAX_SEGMENTS], mixed[JLM_MAX_SEGMENTS];
    float t_scale[JLM_MAX_SEGMENTS], descale[JLM_MAX_SEGMENTS], mx_descale[JLM_MAX_SEGMENTS], mx_s8[JLM_MAX_SEGMENTS];
    int bias_col_v[JLM_MAX_SEGMENTS], slot_of[JLM_MAX_SEGMENTS], tmoff_of[JLM_MAX_SEGMENTS];
    const int *bias_col = bias_col_in ? bias_col_v : nullptr;
    {
        int n = 0, slot = 0, tmo = 0;
        const jlm_segment none{};
        for (int i = 0; i < n_segs; ++i) {
            const bool mx = mixed_in && mixed_in[i].B;
            const int nv = segs_host_in[i].v_end - segs_host_in[i].v_start;
            int cut = (mx && head_split) ? head_split[i] : 0;
            if (cut < 0 || cut % 128) return -1;
            if (cut >= nv) return -1;                // (the whole segment on split rows is said with mixed[i].B == NULL)
            auto put = [&](const jlm_segment &sp, const jlm_segment &m) -> bool {
                if (n >= JLM_MAX_SEGMENTS) return false;
                segs_host[n] = sp; mixed[n] = m;
                t_scale[n] = t_scale_in[i]; descale[n] = descale_in[i];
                bias_col_v[n] = bias_col_in ? bias_col_in[i] : -1;
                mx_descale[n] = mx ? mx_descale_in[i] : 0.0f; mx_s8[n] = mx ? mx_s8_in[i] : 0.0f;
                slot_of[n] = slot; tmoff_of[n] = tmo;
                ++n;
                return true;
            };
            if (cut > 0) {
                jlm_segment hd = segs_host_in[i];
                hd.v_end = hd.v_start + cut;
                if (!put(hd, none)) return -2;
                jlm_segment sp = segs_host_in[i], m = mixed_in[i];
                sp.v_start += cut; m.v_start += cut;        // (B of the split view is not read for a mixed segment)
                m.B = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(m.B) + (size_t)cut * m.ldb * 4);
                if (!put(sp, m)) return -2;
            } else if (!put(segs_host_in[i], mx ? mixed_in[i] : none)) return -2;
            if (mx) { ++slot; tmo += ((segs_host_in[i].k + 2 + 31) / 32) * 128; }
        }
        n_segs = n;
    }
    LseHybridArgs h;
    unsigned char is_mixed[JLM_MAX_SEGMENTS];
    h.mixed_mask = 0;
    LseSplitArgs &a = h.sp;
    a.n_segs = n_segs;
    static int c0x2 = -1, np8 = -1;
    if (c0x2 < 0) { const char *e = getenv("JLM_LSE_C0"); c0x2 = e ? (int)(2.0 * atof(e) + 0.5) : 2; }
    if (np8 < 0) { const char *e = getenv("JLM_LSE_NP8"); np8 = e ? atoi(e) : 1; }
    int ntiles[JLM_MAX_SEGMENTS];
    double ctile[JLM_MAX_SEGMENTS], total = 0.0;
    long n_tiles_all = 0;
    int lds = (2 * 128 * 64 + 3 * 128) * 4, any_mixed = 0;
    for (int i = 0; i < n_segs; ++i) {
        const jlm_segment &sg = segs_host[i];
        is_mixed[i] = mixed[i].B ? 1 : 0;
        h.mixed_mask |= (unsigned)is_mixed[i] << i;
        a.seg[i] = sg;
        a.bias[i] = b2 + sg.v_start;
        a.t_scale[i] = t_scale[i] * 1.4426950408889634f;
        a.descale[i] = descale[i];
        a.bias_col[i] = (short)(bias_col ? bias_col[i] : -1);
        if (is_mixed[i]) {
            const int nb = (sg.k + 2 + 31) / 32, ns16 = (sg.k + 2 + 15) / 16;
            if (!((nb == 7 && ns16 == 13) || (nb == 4 && ns16 == 7) || (nb == 2 && ns16 == 4)) || mixed[i].ldb != 32 * nb || sg.k % 4 || sg.t_off % 4) return -2;
            if ((long)(sg.v_end - sg.v_start) * nb * 128 >= (1l << 31)) return -2;
            jlm_mx::MxSeg &m = h.mx[i];
            m.B = reinterpret_cast<const unsigned char *>(mixed[i].B);
            m.n_vocab = sg.v_end - sg.v_start; m.k = sg.k; m.t_off = sg.t_off; m.nb = nb;
            m.tm_off = tmoff_of[i]; m.seg = slot_of[i]; m.bias2 = nullptr;
            m.descale = mx_descale[i];
            if (!(mx_s8[i] > 0.0f)) return -2;          // (ABI 11: mx6 rows -- s8 = 0 -- have no body in this launch)
            m.cs = mx_s8[i] * (1.0f / 2048.0f);
            const int mtt = jlm_mx::mx_blocks_per_tile(nb);
            ntiles[i] = (m.n_vocab + 32 * mtt - 1) / (32 * mtt);
            ctile[i] = mtt * (ns16 + 2 * nb + 3.0 * c0x2) / 6.0;
            const int l = 2 * 32 * mtt * nb * 128;
            if (l > lds) lds = l;
            any_mixed = 1;
        } else {
            const int ns = (sg.k + 15) / 16;
            const int bc = bias_col ? bias_col[i] : -1;
            if (ns > 13 || (ns > 7 && ns < 13) || bc != sg.k || sg.k % 16 == 0 || sg.k % 4 || sg.ldb % 16 || sg.ldb < ns * 16 || sg.t_off % 4) return -2;
            ntiles[i] = (sg.v_end - sg.v_start + 127) / 128;
            ctile[i] = 2 * ns + c0x2;
        }
        if (ntiles[i] > 65535) return -2;
        total += ctile[i] * ntiles[i];
        n_tiles_all += ntiles[i];
    }
    if (!any_mixed) return -2;
    const int n_ptiles = (n_rows_max + 255) / 256;
    int cap = max_parts < LSES_MAX_SUB ? max_parts : LSES_MAX_SUB;
    cap -= n_segs - 1;
    if (cap > LSES_MAX_PARTS) cap = LSES_MAX_PARTS;
    if (cap < 1) return -1;
    int np = 256 / n_ptiles;
    if (np < 1) np = 1;
    if (np > cap) np = cap;
    if (np >= 8 && np8) np &= ~7;
    { const char *e = getenv("JLM_LSE_NP"); const int f = e ? atoi(e) : 0; if (f > 0 && f <= cap) np = f; }
    if (np > n_tiles_all) np = (int)n_tiles_all;
    static double pro = -1.0;
    if (pro < 0) { const char *e = getenv("JLM_LSE_PRO"); pro = 2.0 * (e ? atof(e) : 17.0); }
    int n_sub = 0, n_cols = 0;
    auto fill = [&](double M, bool emit) -> int {
        int seg = 0, t = 0, cols = 0;
        n_sub = 0;
        while (seg < n_segs) {
            if (emit) { if (cols >= LSES_MAX_PARTS) return -1; a.col_first[cols] = (unsigned char)n_sub; }
            double budget = M;
            bool first = true;
            while (seg < n_segs) {
                if (!first) {
                    if (budget < pro + ctile[seg]) break;
                    budget -= pro;
                }
                const int avail = ntiles[seg] - t;
                int take = (int)(budget / ctile[seg] + 1e-9);
                if (take > avail) take = avail;
                if (take < 1) { if (!first) break; take = 1; }
                if (emit) {
                    if (n_sub >= LSES_MAX_SUB) return -1;
                    a.sub_seg[n_sub] = (unsigned char)seg;
                    a.sub_t0[n_sub] = (unsigned short)t;
                    a.sub_t1[n_sub] = (unsigned short)(t + take);
                }
                ++n_sub;
                budget -= take * ctile[seg];
                first = false;
                t += take;
                if (t < ntiles[seg]) break;
                ++seg;
                t = 0;
            }
            ++cols;
        }
        if (emit) a.col_first[cols] = (unsigned char)n_sub;
        return cols;
    };
    {
        double cmax = 0.0;
        for (int i = 0; i < n_segs; ++i) cmax = ctile[i] > cmax ? ctile[i] : cmax;
        double lo = total / np, hi = total / np + (2.0 * cmax + pro) * n_segs + 1.0;
        for (int it = 0; it < 32; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (fill(mid, false) <= np) hi = mid; else lo = mid;
        }
        n_cols = fill(hi, true);
        if (n_cols < 1 || n_cols > np) return -4;
    }
    a.n_cols = n_cols;
    a.n_sub = n_sub;
    if (n_sub > max_parts) return -1;
    static JlmLdsGrant grant;
    if (int rc = jlm_grant_lds(grant, reinterpret_cast<const void *>(vocab_lse_hybrid_kernel), lds)) return rc;
    hipLaunchKernelGGL(vocab_lse_hybrid_kernel, dim3(n_cols * n_ptiles), dim3(512), lds, (hipStream_t)stream, h, T, ldt,
                       reinterpret_cast<const float *>(Tm), ld_tm, rows, reinterpret_cast<float2 *>(part), ld_part, n_rows_max, n_dev, n_ptiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(int)e - 100;
    return n_sub;
}

// ---------------------------------------------------------------------------------------------
// Word-list log-sum-exp on split rows (selected / incremental vocabulary, single-segment models):
// the split-f16 form of wordlist_lse_mfma_kernel (jlm_gemm.hip).  One workgroup per (sentence, frame)
// group (beams above 32: one per 32 of its rows), its <= 32 hypothesis rows stationary in registers; the word list is walked in
// 32-word tiles, wave w taking tiles w, w+4, ...; a tile's split rows are GATHERED row by row into
// the wave's private LDS ring by the DMA (per-lane source = the word's row).  Per group that is
// ~600 words x 1 KB = 0.6 MB of scattered rows -- the kernel is bound by how many of those requests
// are in flight, not by arithmetic (f32 form: 44 us, 3.5 TB/s with one 4 KB chunk per wave in flight),
// hence the deep ring: WLS_RING chunks of 32 words x 64 k-values (8 KB) per wave, no workgroup
// barrier in the loop, only the wave's own s_waitcnt vmcnt(chunks still allowed in flight).
// The list's word ids and biases are staged in LDS once (the DMA addresses and the fold need them).
#define WLS_RING 3
#define WLS_MAX_WORDS 4096

template <int NS>
__global__ __launch_bounds__(256) void wordlist_lse_split_kernel(
    jlm_segment sg, float t_scale, float descale, const float *__restrict__ b2, const float *__restrict__ T, int ldt,
    const int *__restrict__ g0v, const int *__restrict__ cnt, const int *__restrict__ cnt_idx,
    const int *__restrict__ wl, const int *__restrict__ wl_off, const int *__restrict__ wl_idx, int wl_base,
    float *__restrict__ run_max, double *__restrict__ run_sum, double *__restrict__ lse, int merge, int beam) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using CH = SplitChunks<NS>;
    constexpr int NC = CH::NC;
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    const int j = blockIdx.x;
    // beams above 32: a group's rows are taken by ceil(beam / 32) workgroups (blockIdx.y), 32 rows each
    const int nrows = min(min(cnt[cnt_idx[j]], beam) - 32 * (int)blockIdx.y, 32);
    if (nrows <= 0) return;
    const int gbase = g0v[j] + 32 * (int)blockIdx.y;
    const int lid = wl_base + wl_idx[j];
    const int w0 = wl_off[lid], nw = wl_off[lid + 1] - w0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, li = lane & 31;
    const int K = sg.k, ldb = sg.ldb;
    const float *__restrict__ Bp = sg.B;
    const __amdgpu_buffer_rsrc_t rs_b = jlm_raw_rsrc(Bp);
    const int ntiles = (nw + 31) >> 5;
    // LDS: [4 waves][WLS_RING][32][64] rings | word ids (padded to whole tiles) | biases (base-2 units)
    float *ring = smem + wave * (WLS_RING * 32 * 64);
    int *wid_s = reinterpret_cast<int *>(smem + 4 * WLS_RING * 32 * 64);
    float *bias_l = reinterpret_cast<float *>(wid_s + WLS_MAX_WORDS);
    for (int i = tid; i < ntiles * 32; i += 256) {
        const int w = i < nw ? wl[w0 + i] : -1;
        wid_s[i] = w < 0 ? 0 : w - sg.v_start;                         // padded entries re-read row 0 ...
        bias_l[i] = w < 0 ? JLM_NEG_BIG : b2[w] * LOG2E;               // ... and are switched off by their bias
    }
    __syncthreads();
    float m = JLM_NEG_BIG, s = 0.0f;
    const int my_tiles = wave < ntiles ? (ntiles - wave + 3) / 4 : 0;  // tiles wave, wave + 4, ...
    if (my_tiles > 0) {
        const bool row_ok = li < nrows;
        const float *trow = T + (size_t)(gbase + (row_ok ? li : 0)) * ldt + sg.t_off;
        f16x8 thi[NS], tlo[NS];
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            float x[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = 16 * st + 8 * h + 4 * q;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(trow + (k < K ? k : 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) x[4 * q + e] = (row_ok && k < K) ? v[e] : 0.0f;
            }
            jlm_split8(x, t_scale * LOG2E, thi[st], tlo[st]);
        }
        // DMA piece i of a chunk fills ring rows 4 i .. 4 i + 3 (lane: row 4 i + (lane >> 4), slot lane & 15)
        const int lrow = lane >> 4, pslot = lane & 15;
        const int total = my_tiles * NC;                               // this wave's chunks, in order
        auto issue = [&](int t, int c_start, int slot) {               // chunk (tile t, steps c_start ..) -> ring slot
            float *dst = ring + slot * 32 * 64;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = 4 * i + lrow;
                const int g = pslot ^ (r & 15);
                const int word = wid_s[t * 32 + r];
                const unsigned off = ((unsigned)word * (unsigned)ldb + (unsigned)min(c_start * 16 + g * 4, ldb - 4)) * 4u;
                GLDS16(reinterpret_cast<const char *>(Bp) + off, dst + (4 * i) * 64);
            }
        };
        int goff[4][2];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int p = 0; p < 2; ++p) goff[jj][p] = li * 64 + (((4 * jj + 2 * h + p) ^ (li & 15)) * 4);
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < WLS_RING - 1; ++q)
            if (q < total) issue(wave + 4 * (q / NC), CH::start(q % NC), q);
        int q = 0, slot = 0;
        for (int ti = 0; ti < my_tiles; ++ti) {
            const int t = wave + 4 * ti;
            f32x16 acc0 = zero16, acc1 = zero16;                       // two chains: consecutive MFMAs never share one
#pragma unroll
            for (int c = 0; c < NC; ++c, ++q) {
                // the chunk WLS_RING - 1 ahead goes into the slot consumed one iteration ago (this wave's own:
                // no barrier); then everything but the WLS_RING - 1 youngest chunks has landed
                if (q + WLS_RING - 1 < total) {
                    const int cn = (c + WLS_RING - 1) % NC, dti = (c + WLS_RING - 1) / NC;
                    issue(wave + 4 * (ti + dti), CH::start(cn), slot == 0 ? WLS_RING - 1 : slot - 1);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WLS_RING - 1) * 8) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                const float *bs = ring + slot * 32 * 64;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    if (jj >= CH::size(c)) break;
                    const int st = CH::start(c) + jj;
                    const f16x8 al = *reinterpret_cast<const f16x8 *>(bs + goff[jj][1]);
                    const f16x8 ah = *reinterpret_cast<const f16x8 *>(bs + goff[jj][0]);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, thi[st], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, tlo[st], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, thi[st], acc0, 0, 0, 0);
                }
                slot = (slot + 1 == WLS_RING) ? 0 : slot + 1;
            }
            // fold the tile's 16 base-2 logits of this lane's row
            const float *bt = bias_l + t * 32 + 4 * h;
            float v[16], tmax = JLM_NEG_BIG;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bt + 8 * jj);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[4 * jj + e] = fmaf(acc0[4 * jj + e] + acc1[4 * jj + e], descale, b4[e]);
                    tmax = fmaxf(tmax, v[4 * jj + e]);
                }
            }
            const float mn = fmaxf(m, tmax);
            float add = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) add += __builtin_amdgcn_exp2f(v[r] - mn);
            s = s * __builtin_amdgcn_exp2f(m - mn) + add;
            m = mn;
        }
        const float m2 = __shfl_xor(m, 32), s2 = __shfl_xor(s, 32);
        const float mm = fmaxf(m, m2);
        s = s * __builtin_amdgcn_exp2f(m - mm) + s2 * __builtin_amdgcn_exp2f(m2 - mm);
        m = mm;
    }
    // the four waves' partial (max, sum) per row meet in LDS (base-2 units)
    __syncthreads();
    float *red = smem;                     // [4][32][2], reuses wave 0's ring
    if (h == 0) { red[(wave * 32 + li) * 2] = m; red[(wave * 32 + li) * 2 + 1] = s; }
    __syncthreads();
    if (tid < nrows) {
        float M = red[tid * 2], S = red[tid * 2 + 1];
        for (int w = 1; w < 4; ++w) {
            const float m2 = red[(w * 32 + tid) * 2], s2 = red[(w * 32 + tid) * 2 + 1];
            const float mm = fmaxf(M, m2);
            S = S * __builtin_amdgcn_exp2f(M - mm) + s2 * __builtin_amdgcn_exp2f(m2 - mm);
            M = mm;
        }
        const int g = gbase + tid;
        float Mn = M * LN2;                // natural-log units from here on
        double Sd = (double)S;
        if (merge) {
            const float pm = run_max[g];
            const double ps = run_sum[g];
            const float mm = fmaxf(pm, Mn);
            Sd = ps * exp((double)pm - (double)mm) + Sd * exp((double)Mn - (double)mm);
            Mn = mm;
        }
        run_max[g] = Mn;
        run_sum[g] = Sd;
        lse[g] = (double)Mn + log(Sd);
    }
}

// Returns 0, a HIP error, or -2 when the shape is outside this kernel (caller uses jlm_wordlist_lse).
extern "C" int jlm_wordlist_lse_split(const jlm_segment *seg_host, float t_scale, float descale, const float *b2, const float *T,
                                      int ldt, const int *g0, const int *cnt, const int *cnt_idx, const int *wl,
                                      const int *wl_off, const int *wl_idx, int wl_base, int max_words, float *run_max,
                                      double *run_sum, double *lse, int merge, int beam, int n_groups, void *stream) {
    const jlm_segment sg = *seg_host;
    const int ns = (sg.k + 15) / 16;
    if (ns < 1 || ns > 16 || sg.k % 4 || sg.ldb % 16 || sg.ldb < ns * 16 || sg.t_off % 4 || ldt % 4 || beam > 64) return -2;
    if (max_words > WLS_MAX_WORDS - 32) return -2;
    if (n_groups <= 0) return 0;
    const int lds = (4 * WLS_RING * 32 * 64 + 2 * WLS_MAX_WORDS) * 4;
    hipStream_t st = (hipStream_t)stream;
#define JLM_WLS_LAUNCH(N)                                                                                                  \
    do {                                                                                                                   \
        static JlmLdsGrant grant;                                                                                          \
        if (int rc = jlm_grant_lds(grant, reinterpret_cast<const void *>(wordlist_lse_split_kernel<N>), lds)) return rc;          \
        hipLaunchKernelGGL(wordlist_lse_split_kernel<N>, dim3(n_groups, (beam + 31) / 32), dim3(256), lds, st, sg, t_scale, descale, b2, T, ldt, \
                           g0, cnt, cnt_idx, wl, wl_off, wl_idx, wl_base, run_max, run_sum, lse, merge, beam);             \
    } while (0)
    if (ns <= 2) JLM_WLS_LAUNCH(2);
    else if (ns <= 4) JLM_WLS_LAUNCH(4);
    else if (ns <= 8) JLM_WLS_LAUNCH(8);
    else if (ns <= 12) JLM_WLS_LAUNCH(12);
    else JLM_WLS_LAUNCH(16);
#undef JLM_WLS_LAUNCH
    JLM_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Incremental vocabulary, K11 (decoder_dynamic.py:133-148): at frame f every OLDER row of a
// sentence learns the words that first appear at f -- the same short list (tens of words) for all of
// the sentence's rows.  The group-per-(frame, sentence) kernels above gather that list once per
// older frame, (f - 1) x 256 workgroups of a few words each (62 us per call at f ~ 10, bound by
// workgroup turnover).  Here: ONE workgroup per sentence gathers the list's split rows once (all
// tiles requested up front, <= 128 words = 128 KB of LDS), then its 4 waves walk the sentence's older
// rows in blocks of 32 (row operands split on the fly, 3 f16 MFMAs per k-step) and merge the list's
// (max, sum exp) into each row's running pair.  Rows: g = frame * rmax + sentence * beam + slot,
// slot < cnt[frame * n_sent + sentence] (the layout of jlm_beam_state).
#define WLM_MAX_WORDS 128

template <int NS>
__global__ __launch_bounds__(256) void wordlist_merge_split_kernel(
    jlm_segment sg, float t_scale, float descale, const float *__restrict__ b2, const float *__restrict__ T, int ldt,
    const int *__restrict__ cnt, int n_sent, int beam, int n_old_frames, const int *__restrict__ wl,
    const int *__restrict__ wl_off, int wl_base, float *__restrict__ run_max, double *__restrict__ run_sum,
    double *__restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    const int sent = blockIdx.x;
    const int w0 = wl_off[wl_base + sent], nw = wl_off[wl_base + sent + 1] - w0;
    if (nw <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, li = lane & 31;
    const int K = sg.k, ldb = sg.ldb, rmax = n_sent * beam;
    const int ntiles = (nw + 31) >> 5;
    // LDS: [ntiles][NS / 4 chunks][32 words][64 values as split rows] | biases | word rows
    constexpr int NCH = (NS + 3) / 4;
    float *Ws = smem;
    float *bias_l = smem + WLM_MAX_WORDS * NCH * 64;
    int *wid_s = reinterpret_cast<int *>(bias_l + WLM_MAX_WORDS);
    for (int i = tid; i < ntiles * 32; i += 256) {
        const int w = wl[w0 + (i < nw ? i : 0)];                       // padded entries re-read the first word ...
        wid_s[i] = w - sg.v_start;
        bias_l[i] = i < nw ? b2[w] * LOG2E : JLM_NEG_BIG;              // ... and are switched off by their bias
    }
    __syncthreads();
    {   // gather: a piece = 4 words x one 64-value chunk (16 granules), lane (lane >> 4, lane & 15); every piece
        // of the list is requested before anything is waited for
        const int lrow = lane >> 4, pslot = lane & 15;
        const int npieces = ntiles * 8 * NCH;
        for (int p = wave; p < npieces; p += 4) {
            const int cchunk = p % NCH, w4 = p / NCH;                  // 64-value chunk, group of 4 words
            const int r = w4 * 4 + lrow;                               // word position in the (padded) list
            const int g = pslot ^ (r & 15);
            const unsigned off = ((unsigned)wid_s[r] * (unsigned)ldb + (unsigned)min(cchunk * 64 + g * 4, ldb - 4)) * 4u;
            GLDS16(reinterpret_cast<const char *>(sg.B) + off, Ws + ((size_t)((w4 >> 3) * NCH + cchunk) * 32 + ((w4 & 7) * 4)) * 64);
        }
    }
    __syncthreads();                                                   // vmcnt(0): the whole list has landed
    int goff[4][2];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int p = 0; p < 2; ++p) goff[jj][p] = li * 64 + (((4 * jj + 2 * h + p) ^ (li & 15)) * 4);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int n_slots = n_old_frames * beam;                            // uncompacted (frame, slot) pairs of this sentence
    for (int blk = wave; blk * 32 < n_slots; blk += 4) {
        const int i = blk * 32 + li;
        const int fr = i / beam, slot = i - fr * beam;
        const bool row_ok = i < n_slots && slot < cnt[fr * n_sent + sent];
        if (!__builtin_amdgcn_readfirstlane(__any(row_ok))) continue;
        const int g = fr * rmax + sent * beam + slot;
        const float *trow = T + (size_t)(row_ok ? g : 0) * ldt + sg.t_off;
        f16x8 thi[NS], tlo[NS];
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            float x[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = 16 * st + 8 * h + 4 * q;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(trow + (k < K ? k : 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) x[4 * q + e] = (row_ok && k < K) ? v[e] : 0.0f;
            }
            jlm_split8(x, t_scale * LOG2E, thi[st], tlo[st]);
        }
        float m = JLM_NEG_BIG, s = 0.0f;
        for (int t = 0; t < ntiles; ++t) {
            f32x16 acc0 = zero16, acc1 = zero16;
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const float *bs = Ws + (size_t)(t * NCH + (st >> 2)) * 32 * 64;
                const f16x8 al = *reinterpret_cast<const f16x8 *>(bs + goff[st & 3][1]);
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(bs + goff[st & 3][0]);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, thi[st], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, tlo[st], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, thi[st], acc0, 0, 0, 0);
            }
            const float *bt = bias_l + t * 32 + 4 * h;
            float v[16], tmax = JLM_NEG_BIG;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bt + 8 * jj);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[4 * jj + e] = fmaf(acc0[4 * jj + e] + acc1[4 * jj + e], descale, b4[e]);
                    tmax = fmaxf(tmax, v[4 * jj + e]);
                }
            }
            const float mn = fmaxf(m, tmax);
            float add = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) add += __builtin_amdgcn_exp2f(v[r] - mn);
            s = s * __builtin_amdgcn_exp2f(m - mn) + add;
            m = mn;
        }
        const float m2 = __shfl_xor(m, 32), s2 = __shfl_xor(s, 32);
        const float mm = fmaxf(m, m2);
        s = s * __builtin_amdgcn_exp2f(m - mm) + s2 * __builtin_amdgcn_exp2f(m2 - mm);
        if (h == 0 && row_ok) {                    // merge into the row's running pair (natural-log units)
            float Mn = mm * LN2;
            double Sd = (double)s;
            const float pm = run_max[g];
            const double ps = run_sum[g];
            const float mx = fmaxf(pm, Mn);
            Sd = ps * exp((double)pm - (double)mx) + Sd * exp((double)Mn - (double)mx);
            run_max[g] = mx;
            run_sum[g] = Sd;
            lse[g] = (double)mx + log(Sd);
        }
    }
}

// Returns 0, a HIP error, or -2 when the shape is outside this kernel (use jlm_wordlist_lse with merge = 1).
extern "C" int jlm_wordlist_merge_split(const jlm_segment *seg_host, float t_scale, float descale, const float *b2,
                                        const float *T, int ldt, const int *cnt, int n_sent, int beam, int n_old_frames,
                                        const int *wl, const int *wl_off, int wl_base, int max_words, float *run_max,
                                        double *run_sum, double *lse, void *stream) {
    const jlm_segment sg = *seg_host;
    const int ns = (sg.k + 15) / 16;
    if (ns < 1 || ns > 16 || sg.k % 4 || sg.ldb % 16 || sg.ldb < ns * 16 || sg.t_off % 4 || ldt % 4 || beam > 64) return -2;
    if (max_words > WLM_MAX_WORDS) return -2;
    if (n_sent <= 0 || n_old_frames <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
#define JLM_WLM_LAUNCH(N)                                                                                                  \
    do {                                                                                                                   \
        const int lds = (WLM_MAX_WORDS * ((N + 3) / 4) * 64 + 2 * WLM_MAX_WORDS) * 4;                                                      \
        static JlmLdsGrant grant;                                                                                          \
        if (int rc = jlm_grant_lds(grant, reinterpret_cast<const void *>(wordlist_merge_split_kernel<N>), lds)) return rc;          \
        hipLaunchKernelGGL(wordlist_merge_split_kernel<N>, dim3(n_sent), dim3(256), lds, st, sg, t_scale, descale, b2, T,  \
                           ldt, cnt, n_sent, beam, n_old_frames, wl, wl_off, wl_base, run_max, run_sum, lse);              \
    } while (0)
    if (ns <= 4) JLM_WLM_LAUNCH(4);
    else if (ns <= 8) JLM_WLM_LAUNCH(8);
    else if (ns <= 12) JLM_WLM_LAUNCH(12);
    else JLM_WLM_LAUNCH(16);
#undef JLM_WLM_LAUNCH
    JLM_LAUNCH_CHECK();
    return 0;
}
