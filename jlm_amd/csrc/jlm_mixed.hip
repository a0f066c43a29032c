// jlm_mixed.hip -- the vocabulary projection + log-sum-exp with the cross terms of the split product on the INT8 matrix pipe
// (jlm_vocab_lse_mixed, jlm_pack_mixed; include/jlm_hip.h).  Reference: project + softmax, decoder/model.py:141-193, 15-20.
//
// The split-f16 scheme (jlm_split.hip) writes x 2^e = hi + lo (both f16) and takes three f16 products per pair,
//      t.b  ~  t_hi.b_hi + t_hi.b_lo + t_lo.b_hi ,
// each a v_mfma_f32_32x32x16_f16.  The two cross terms are 2^-11 of the first: they need 8 good bits, not 11.  Here
//      t_hi.b_hi               f16 x f16, exact products, f32 accumulate        (v_mfma_f32_32x32x16_f16, 16 k per instruction)
//      t_hi.b_lo + t_lo.b_hi   int8 x int8 into ONE i32 accumulator             (v_mfma_i32_32x32x32_i8, 32 k per instruction)
// with  hi8 = rint(hi / s),  lo8 = rint(lo / (s 2^-11))  and s a power of two per T row (from the row's largest |hi|) and per
// vocabulary segment: both cross terms then carry the same scale  s_t s_b 2^-11  and share the accumulator.  Per 32 k-values and
// 32 x 32 block that is 2 + 2 matrix instructions instead of 6, and on this chip -- where these loops are bound by the clock the
// power governor grants under matrix load, not by issue slots -- a pure MFMA stream of that mix runs in 0.64 of
// the three-pass stream's time (tools/probes/mfma_mix_rate.hip: 2.50 vs 3.78 ms for equal work on random operands).
// Error: the quantisation steps are 2^-8 of the row / segment maximum on terms that are 2^-11 of the product -- products good to
// ~2^-20 of |t||b| where three f16 passes give 2^-23.  Measured on the BASELINE-shaped fixtures (Gaussian-like blocks): logits
// within 8e-6 of the row's logit scale (bar: 1e-4), the log-normaliser -- what the decode consumes -- within 3e-7, path scores
// within 3e-7 of the oracle like the split form; on peaked distributions (logits of +-10 .. +-20) the scores move by 5e-5 .. 3e-4
// where the split form's move by 5e-6 .. 3e-5, and heavy-tailed blocks lose more (one int8 scale per segment): DeviceModel keeps
// those on split rows (JLM_MIXED_MAX_SPREAD).  DESIGN.md 6c.1.
//
// Row format ("mixed rows"), per 32 k-values one 128-byte block:
//      [ 32 x f16 hi | 32 x int8 hi8 | 32 x int8 lo8 ]      = 8 granules of 16 bytes: 0-3 hi (8 k each), 4-5 hi8, 6-7 lo8
// The bias rides in the f16 part as TWO extra columns (k: hi of b2 2^eB log2e, against the constant 2^eT on the T side; k + 1:
// its f16 residual x 2^11 against 2^(eT-11)): both T constants are powers of two, so the bias is exact in the f16 product and
// stays out of the int8 planes (and out of their scales).
//
// Kernel: rows-stationary like vocab_lse_split8_kernel -- a workgroup of 8 waves keeps 256 hypothesis rows' operands in
// registers (32 rows per wave: f16 hi, int8 hi8, int8 lo8) and streams its vocabulary column through LDS -- but in tiles of
// 64 .. 256 words with the WHOLE contraction of a tile resident (two buffers of up to 64 KB): one barrier per tile, no chunks; LDS-DMA
// by buffer loads (counted lgkmcnt waits, jlm_gate.hip), fragment registers refilled in place behind the MFMA that read them.
#include "jlm_common.h"
#include <stdlib.h>
#include <type_traits>

#include "jlm_mixed_body.h"
#include "jlm_mx6_body.h"
using namespace jlm_mx;

#ifndef JLM_MX_WIDE_DEFAULT
#define JLM_MX_WIDE_DEFAULT -1
#endif

namespace {

// ------------------------------------------------------------------------------------------------ packing (load time)
// one thread per (row, 32-k block)
__global__ __launch_bounds__(256) void pack_mixed_kernel(const float *__restrict__ src, int rows, int k, int ld, const float *__restrict__ bias,
                                                         float scale, float bias_scale, float inv_s8, unsigned char *__restrict__ dst,
                                                         int nb) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * nb) return;
    const int r = idx / nb, j = idx - r * nb;
    _Float16 hi[32];
    signed char h8[32], l8[32];
    const float inv_lo = inv_s8 * 2048.0f;
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        const int kk = 32 * j + e;
        float x = 0.0f;
        bool real = kk < k;
        if (real) x = src[(size_t)r * ld + kk] * scale;
        _Float16 h = (_Float16)x;
        float lo = x - (float)h;
        if (!real) {
            // bias columns: k = hi of the scaled bias, k + 1 = its f16 residual x 2^11 (exactly representable); zero int8 planes
            const float xb = (bias && k + 2 <= 32 * nb) ? bias[r] * bias_scale : 0.0f;
            const _Float16 bh = (_Float16)xb;
            if (kk == k) h = bh;
            else if (kk == k + 1) h = (_Float16)((xb - (float)bh) * 2048.0f);
            else h = (_Float16)0.0f;
            lo = 0.0f;
        }
        hi[e] = h;
        const float qh = real ? rintf((float)h * inv_s8) : 0.0f, ql = rintf(lo * inv_lo);
        h8[e] = (signed char)fminf(fmaxf(qh, -127.0f), 127.0f);
        l8[e] = (signed char)fminf(fmaxf(ql, -127.0f), 127.0f);
    }
    unsigned char *o = dst + ((size_t)r * nb + j) * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4 *>(o + 16 * q) = *reinterpret_cast<const f32x4 *>(hi + 8 * q);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        *reinterpret_cast<f32x4 *>(o + 64 + 16 * q) = *reinterpret_cast<const f32x4 *>(h8 + 16 * q);
        *reinterpret_cast<f32x4 *>(o + 96 + 16 * q) = *reinterpret_cast<const f32x4 *>(l8 + 16 * q);
    }
}

// Hypothesis rows (T, f32) -> packed rows: per segment nb blocks [32 f16 hi | 32 int8 hi8 | 32 int8 lo8] of x = T t_scale log2 e,
// the bias constants 2^eT / 2^(eT-11) at columns k, k + 1 of the f16 part, and at the END of the row JLM_MAX_SEGMENTS floats: the
// row's int8 scale per segment (the power of two at or above max |hi| / 127).  One wave per row; done ONCE per row and frame --
// the vocabulary kernel's workgroups (24 columns per row tile) only load the result.
struct MxTSeg { int k, t_off, nb, tm_off; float t_scale, tc; };      // tc = 0: no bias columns (k = 32 nb)
struct MxTArgs { int n_segs; MxTSeg seg[JLM_MAX_SEGMENTS]; };

__global__ __launch_bounds__(256) void pack_t_mixed_kernel(MxTArgs a, const float *__restrict__ T, int ldt, const int *__restrict__ rows,
                                                           int n_rows_max, const int *__restrict__ n_dev, unsigned char *__restrict__ Tm,
                                                           int ld_tm) {
    // four rows per workgroup, one wave each; lane l owns the 16-value group l of the row's groups (two per 32-k block, all segments
    // concatenated: 26 groups for the D-softmax* 200 / 100 / 50 rows): its 16 values are loaded once and stay in registers
    const int n = n_dev ? min(*n_dev, n_rows_max) : n_rows_max;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int g = rows ? rows[r] : r;
    const int lane = threadIdx.x & 63;
    const float *trow = T + (size_t)g * ldt;
    unsigned char *oblk = Tm + mx_tm_block(r, ld_tm);          // COMPACT: packed row r = hypothesis row rows[r]; granule-major (jlm_mixed_body.h)
    // (the launcher admits at most 64 groups -- 32 blocks -- per row: ONE pass, so every segment's maximum is reduced over all of
    //  the lanes that hold it before anything is quantised)
    int total = 0;
    for (int si = 0; si < a.n_segs; ++si) total += 2 * a.seg[si].nb;
    {
        const int grp0 = 0;
        const int grp = grp0 + lane;
        int si = 0, base = 0;                         // the group's segment
        while (si + 1 < a.n_segs && grp >= base + 2 * a.seg[si].nb) { base += 2 * a.seg[si].nb; ++si; }
        const bool act = grp < total;
        const MxTSeg sg = a.seg[si];
        const int gs = grp - base;                    // group inside the segment: block gs >> 1, half gs & 1
        f32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k0 = 16 * gs + 4 * q;
            v[q] = *reinterpret_cast<const f32x4 *>(trow + sg.t_off + ((act && k0 < sg.k) ? k0 : 0));
        }
        float amax = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) amax = fmaxf(amax, (act && 16 * gs + 4 * q + e < sg.k) ? fabsf((float)(_Float16)(v[q][e] * sg.t_scale)) : 0.0f);
        // per segment: the maximum over the lanes that hold it
        float smax = 0.0f;
        for (int sj = 0; sj < a.n_segs; ++sj) {
            float x = (act && si == sj) ? amax : 0.0f;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) x = fmaxf(x, __shfl_xor(x, off));
            if (si == sj) smax = x;
            if (lane == 0 && grp0 == 0) {
                const float w = x * (1.0f / 127.0f);
                const int bt = (__float_as_int(w) + 0x007fffff) & 0x7f800000;
                *reinterpret_cast<float *>(oblk + mx_tm_scale(ld_tm, r, sj)) = x > 0.0f ? __int_as_float(bt) : 1.0f;
            }
        }
        if (!act) return;
        const float want = smax * (1.0f / 127.0f);
        const int bits = (__float_as_int(want) + 0x007fffff) & 0x7f800000;
        const float s_t = smax > 0.0f ? __int_as_float(bits) : 1.0f;
        const float inv_h = 1.0f / s_t, inv_l = inv_h * 2048.0f;
        const int j = gs >> 1, half = gs & 1;
        _Float16 hi[16];
        int ph[4], pl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int wh = 0, wl = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 16 * gs + 4 * q + e;
                const bool real = k < sg.k;
                const float x = real ? v[q][e] * sg.t_scale : (k == sg.k ? sg.tc : (k == sg.k + 1 ? sg.tc * (1.0f / 2048.0f) : 0.0f));
                const _Float16 h = (_Float16)x;
                hi[4 * q + e] = h;
                const float hf_ = (float)h;
                const int qh = real ? (int)rintf(hf_ * inv_h) : 0;
                const int ql = real ? (int)rintf((x - hf_) * inv_l) : 0;
                wh |= (max(min(qh, 127), -127) & 0xff) << (8 * e);
                wl |= (max(min(ql, 127), -127) & 0xff) << (8 * e);
            }
            ph[q] = wh; pl[q] = wl;
        }
        // granules of the row's block j: 2 half, 2 half + 1 (f16 hi), 4 + half (hi8), 6 + half (lo8)
        *reinterpret_cast<f32x4 *>(oblk + mx_tm_granule(sg.tm_off, j * 8 + 2 * half, r)) = *reinterpret_cast<const f32x4 *>(hi);
        *reinterpret_cast<f32x4 *>(oblk + mx_tm_granule(sg.tm_off, j * 8 + 2 * half + 1, r)) = *reinterpret_cast<const f32x4 *>(hi + 8);
        *reinterpret_cast<i32x4 *>(oblk + mx_tm_granule(sg.tm_off, j * 8 + 4 + half, r)) = i32x4{ph[0], ph[1], ph[2], ph[3]};
        *reinterpret_cast<i32x4 *>(oblk + mx_tm_granule(sg.tm_off, j * 8 + 6 + half, r)) = i32x4{pl[0], pl[1], pl[2], pl[3]};
    }
}


// ------------------------------------------------------------------------------------------------ mx6 rows (round 6, jlm_mx6_body.h)
// vocabulary rows: one thread per (row, 32-k block).  hi as in pack_mixed_kernel (bias columns included); the FP6 planes -- hi6 = FP6 of
// the f16 hi, lo6 = FP6 of the f32 residual, each with the block's own E8M0 scale -- cover the REAL k-values only (the bias columns are
// exact in the f16 product).  Block 0's granule 7 collects the row's scale bytes: thread (r, j) writes bytes j and 8 + j of it.
__global__ __launch_bounds__(256) void pack_mx6_kernel(const float *__restrict__ src, int rows, int k, int ld, const float *__restrict__ bias,
                                                       float scale, float bias_scale, unsigned char *__restrict__ dst, int nb) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * nb) return;
    const int r = idx / nb, j = idx - r * nb;
    _Float16 hi[32];
    float hf32[32], lo[32];
    float amax_h = 0.0f, amax_l = 0.0f;
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        const int kk = 32 * j + e;
        const bool real = kk < k;
        float x = 0.0f;
        if (real) x = src[(size_t)r * ld + kk] * scale;
        asm volatile("" : "+v"(x));                  // (the scaled value is used twice: jlm_common.h jlm_split2)
        _Float16 h = (_Float16)x;
        asm volatile("" : "+v"(h));
        float l = x - (float)h;
        if (!real) {
            const float xb = (bias && k + 2 <= 32 * nb) ? bias[r] * bias_scale : 0.0f;
            const _Float16 bh = (_Float16)xb;
            if (kk == k) h = bh;
            else if (kk == k + 1) h = (_Float16)((xb - (float)bh) * 2048.0f);
            else h = (_Float16)0.0f;
            l = 0.0f;
        }
        hi[e] = h;
        hf32[e] = real ? (float)h : 0.0f;
        lo[e] = l;
        amax_h = fmaxf(amax_h, fabsf(hf32[e]));
        amax_l = fmaxf(amax_l, fabsf(l));
    }
    const int bh_ = mx6_block_byte(amax_h), bl_ = mx6_block_byte(amax_l);
    unsigned ph[6], pl[6];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        unsigned ch[16], cl[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { ch[e] = mx6_code(hf32[16 * half + e], bh_); cl[e] = mx6_code(lo[16 * half + e], bl_); }
        mx6_pack16(ch, ph + 3 * half);
        mx6_pack16(cl, pl + 3 * half);
    }
    unsigned char *o = dst + ((size_t)r * nb + j) * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4 *>(o + 16 * q) = *reinterpret_cast<const f32x4 *>(hi + 8 * q);
    // granule 4: half 0 (hi6) dwords 0-3; granule 5: [hi6 dwords 4-5 | lo6 dwords 4-5]; granule 6: half 1 (lo6) dwords 0-3
    *reinterpret_cast<i32x4 *>(o + 64) = i32x4{(int)ph[0], (int)ph[1], (int)ph[2], (int)ph[3]};
    *reinterpret_cast<i32x4 *>(o + 80) = i32x4{(int)ph[4], (int)ph[5], (int)pl[4], (int)pl[5]};
    *reinterpret_cast<i32x4 *>(o + 96) = i32x4{(int)pl[0], (int)pl[1], (int)pl[2], (int)pl[3]};
    unsigned char *g7 = dst + (size_t)r * nb * 128 + 112;
    g7[j] = (unsigned char)bh_;
    g7[8 + j] = (unsigned char)bl_;
    if (j == 0) { for (int jj = nb; jj < 8; ++jj) { g7[jj] = 0; g7[8 + jj] = 0; } }
    else *reinterpret_cast<i32x4 *>(o + 112) = i32x4{0, 0, 0, 0};
}

// hypothesis rows: one wave per row; a lane owns ONE PLANE of one 16-value group of the row's segments (lanes 0-31: the lo6 plane of groups
// g0 .. g0 + 31, lanes 32-63: the hi6 plane of the same groups) -- the quantiser is ~20 VALU instructions per value, and with a lane
// per group and BOTH planes (the int8 packer's assignment) the launch took 12-14 us where the int8 one takes 9.
// The halves are SWAPPED against the vocabulary rows (half 0 = lo6, half 1 = hi6): the instruction pairs k-slot with k-slot.
__global__ __launch_bounds__(256) void pack_t_mx6_kernel(MxTArgs a, const float *__restrict__ T, int ldt, const int *__restrict__ rows,
                                                         int n_rows_max, const int *__restrict__ n_dev, unsigned char *__restrict__ Tm, int ld_tm) {
    const int n = n_dev ? min(*n_dev, n_rows_max) : n_rows_max;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int g = rows ? rows[r] : r;
    const int lane = threadIdx.x & 63;
    const int sl = lane >> 5;                         // plane: 0 = lo6 (half 0 of the row format), 1 = hi6
    const float *trow = T + (size_t)g * ldt;
    unsigned char *oblk = Tm + mx_tm_block(r, ld_tm);
    int total = 0;
    for (int si = 0; si < a.n_segs; ++si) total += 2 * a.seg[si].nb;
    for (int g0 = 0; g0 < total; g0 += 32) {
        const int grp = g0 + (lane & 31);
        int si = 0, base = 0;
        while (si + 1 < a.n_segs && grp >= base + 2 * a.seg[si].nb) { base += 2 * a.seg[si].nb; ++si; }
        const bool act = grp < total;
        const MxTSeg sg = a.seg[si];
        const int gs = grp - base;                    // group inside the segment: block gs >> 1, half gs & 1 (partner lane ^ 1: base is even)
        f32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k0 = 16 * gs + 4 * q;
            v[q] = *reinterpret_cast<const f32x4 *>(trow + sg.t_off + ((act && k0 < sg.k) ? k0 : 0));
        }
        _Float16 hi[16];
        float val[16];
        float amax = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = 16 * gs + e;
            const bool real = act && k < sg.k;
            float x = real ? v[e >> 2][e & 3] * sg.t_scale : (k == sg.k ? sg.tc : (k == sg.k + 1 ? sg.tc * (1.0f / 2048.0f) : 0.0f));
            asm volatile("" : "+v"(x));               // (the scaled value is used twice: jlm_common.h jlm_split2)
            _Float16 h = (_Float16)x;
            asm volatile("" : "+v"(h));
            hi[e] = h;
            val[e] = real ? (sl ? (float)h : x - (float)h) : 0.0f;
            amax = fmaxf(amax, fabsf(val[e]));
        }
        amax = fmaxf(amax, __shfl_xor(amax, 1));
        if (act) {
            const int byte = mx6_block_byte(amax);
            unsigned c[16], pw[3];
#pragma unroll
            for (int e = 0; e < 16; ++e) c[e] = mx6_code(val[e], byte);
            mx6_pack16(c, pw);
            const int j = gs >> 1, half = gs & 1;
            // the f16 part: granules 2 half, 2 half + 1 of the block -- one each from the group's two lanes
            *reinterpret_cast<f32x4 *>(oblk + mx_tm_granule(sg.tm_off, j * 8 + 2 * half + sl, r)) = *reinterpret_cast<const f32x4 *>(hi + 8 * sl);
            // plane sl: dwords 0-3 in granule 4 + 2 sl, dwords 4-5 in granule 5 at byte 8 sl; this lane holds dwords 3 half .. 3 half + 2
            unsigned char *ga = oblk + mx_tm_granule(sg.tm_off, j * 8 + 4 + 2 * sl, r), *g5 = oblk + mx_tm_granule(sg.tm_off, j * 8 + 5, r) + 8 * sl;
            if (half == 0) {
                *reinterpret_cast<unsigned *>(ga + 0) = pw[0]; *reinterpret_cast<unsigned *>(ga + 4) = pw[1]; *reinterpret_cast<unsigned *>(ga + 8) = pw[2];
                (oblk + mx_tm_granule(sg.tm_off, 7, r))[8 * sl + j] = (unsigned char)byte;
            } else {
                *reinterpret_cast<unsigned *>(ga + 12) = pw[0]; *reinterpret_cast<unsigned *>(g5 + 0) = pw[1]; *reinterpret_cast<unsigned *>(g5 + 4) = pw[2];
            }
        }
    }
}

#ifdef JLM_WGTIME
// -DJLM_WGTIME: per workgroup [start, end] on the constant 100 MHz clock, the segment of its last sub-range, shader-clock cycles
// in between (tools/probes/mixed_clock.py: workgroup durations and the shader clock the kernel actually ran at)
static __device__ unsigned long long jlm_prof_wg_mx[1024][4];
extern "C" int jlm_prof_read_wg_mx(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(jlm_prof_wg_mx), sizeof(jlm_prof_wg_mx)) == hipSuccess ? 0 : -1;
}
#define MX_WG_T0 const unsigned long long wg_t0 = wall_clock64(), wg_c0 = clock64(); int si_last = 0;
#define MX_WG_T1 if (threadIdx.x == 0 && b < 1024) { jlm_prof_wg_mx[b][0] = wg_t0; jlm_prof_wg_mx[b][1] = wall_clock64(); jlm_prof_wg_mx[b][2] = si_last; jlm_prof_wg_mx[b][3] = clock64() - wg_c0; }
#else
#define MX_WG_T0
#define MX_WG_T1
#endif

// The kernel hosts the bodies of a LIST of (NB, NS16) shapes -- the segments of one model -- and picks per sub-range.  Hosting
// every shape at once (16 bodies) costs hundreds of spilled registers in all of them; the launcher instantiates the lists it
// knows (the BASELINE D-softmax* model: k = 200, 100, 50) and a generic kernel of out-of-line bodies for the rest.
template <bool INLINE, bool XB, int NB, int NS16>
struct MxCall {
    static __device__ __forceinline__ void run(const MxSeg &sg, int vt0, int vt1, int pt, int n_paths, const float *T, int ldt, const int *rows,
                                               float2 *prow, unsigned char *smem) {
        mx_body<NB, NS16, mx_blocks_per_tile(NB), XB>(sg, vt0, vt1, pt, n_paths, T, ldt, rows, prow, smem);
    }
};
template <bool XB, int NB, int NS16>
__device__ __noinline__ void mx_body_outline(const MxSeg &sg, int vt0, int vt1, int pt, int n_paths, const float *T, int ldt, const int *rows,
                                             float2 *prow, unsigned char *smem) {
    mx_body<NB, NS16, mx_blocks_per_tile(NB), XB>(sg, vt0, vt1, pt, n_paths, T, ldt, rows, prow, smem);
}
template <bool XB, int NB, int NS16>
struct MxCall<false, XB, NB, NS16> {
    static __device__ __forceinline__ void run(const MxSeg &sg, int vt0, int vt1, int pt, int n_paths, const float *T, int ldt, const int *rows,
                                               float2 *prow, unsigned char *smem) {
        mx_body_outline<XB, NB, NS16>(sg, vt0, vt1, pt, n_paths, T, ldt, rows, prow, smem);
    }
};

template <bool INLINE, bool XB, int... SH>      // SH = NB0, NS0, NB1, NS1, ...
struct MxDispatch;
template <bool INLINE, bool XB>
struct MxDispatch<INLINE, XB> {
    static __device__ __forceinline__ void run(const MxSeg &, int, int, int, int, int, const float *, int, const int *, float2 *, unsigned char *) {}
};
template <bool INLINE, bool XB, int NB, int NS16, int... REST>
struct MxDispatch<INLINE, XB, NB, NS16, REST...> {
    static __device__ __forceinline__ void run(const MxSeg &sg, int ns16, int vt0, int vt1, int pt, int n_paths, const float *T, int ldt,
                                               const int *rows, float2 *prow, unsigned char *smem) {
        if (sg.nb == NB && ns16 == NS16) MxCall<INLINE, XB, NB, NS16>::run(sg, vt0, vt1, pt, n_paths, T, ldt, rows, prow, smem);
        else MxDispatch<INLINE, XB, REST...>::run(sg, ns16, vt0, vt1, pt, n_paths, T, ldt, rows, prow, smem);
    }
};

// XB: every segment of the launch is in the external-bias form (mx_body XBIAS); f16 steps = min(2 nb, ceil((k + 2) / 16))
template <bool INLINE, bool XB, int... SH>
__global__ __launch_bounds__(512, 1) void vocab_lse_mixed_kernel(MxArgs a, const float *__restrict__ T, int ldt, const int *__restrict__ rows,
                                                                 float2 *__restrict__ part, int ld_part, int n_rows_max,
                                                                 const int *n_dev, int n_ptiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mx_smem[];
    const int n_paths = n_dev ? min(*n_dev, n_rows_max) : n_rows_max;
    // XCD-aware order as in vocab_lse_split_main: column p on XCD p % 8 with all of its row tiles
    const int b = blockIdx.x;
    int p, pt;
    const int nb8 = (a.n_cols & ~7) * n_ptiles;
    if (b < nb8) { const int x = b & 7, jb = b >> 3; p = (jb / n_ptiles) * 8 + x; pt = jb % n_ptiles; }
    else { const int bb = b - nb8; p = (a.n_cols & ~7) + bb / n_ptiles; pt = bb % n_ptiles; }
    if (p >= a.n_cols || pt * 256 >= n_paths) return;
    MX_WG_T0
    for (int r = a.col_first[p]; r < a.col_first[p + 1]; ++r) {
        const MxSeg sg = a.seg[a.sub_seg[r]];
        const int vt0 = a.sub_t0[r], vt1 = a.sub_t1[r];
        float2 *prow = part + (size_t)r * ld_part;
        if (r != a.col_first[p]) __syncthreads();
        const int ns16 = XB ? 2 * sg.nb : (sg.k + 2 + 15) >> 4;
#ifdef JLM_WGTIME
        si_last = a.sub_seg[r];
#endif
        MxDispatch<INLINE, XB, SH...>::run(sg, ns16, vt0, vt1, pt, n_paths, T, ldt, rows, prow, mx_smem);
    }
    MX_WG_T1
}
// the shapes of BASELINE configs[1] (D-softmax* 200 / 100 / 50: k + 2 = 202, 102, 52), bodies inlined
#define MX_KERNEL_DSOFTMAX vocab_lse_mixed_kernel<true, false, 7, 13, 4, 7, 2, 4>
// every other shape, out-of-line bodies
#define MX_KERNEL_GENERIC vocab_lse_mixed_kernel<false, false, 1, 1, 1, 2, 2, 3, 2, 4, 3, 5, 3, 6, 4, 7, 4, 8, 5, 9, 5, 10, 6, 11, 6, 12, 7, 13, 7, 14, 8, 15, 8, 16>
// the tied k = 256 models (BASELINE configs[2..4]: one segment, no spare column for the bias), body inlined
#define MX_KERNEL_TIED vocab_lse_mixed_kernel<true, true, 8, 16>
// external-bias form of the other contractions that fill their last block (k = 64, 128, 192), out-of-line
#define MX_KERNEL_GENERIC_XB vocab_lse_mixed_kernel<false, true, 2, 4, 4, 8, 6, 12, 8, 16>


#ifdef JLM_MX_RESOURCES
// one kernel per instantiation: hipcc -S -DJLM_MX_RESOURCES shows each form's own register count (the shipped kernel hosts all)
#define MX_RES(NB_, NS_) __global__ __launch_bounds__(512, 1) void mx_res_##NB_##_##NS_(MxSeg sg, const float *T, int ldt, const int *rows, float2 *part) { \
        extern __shared__ __attribute__((aligned(16))) unsigned char sm_[]; mx_body<NB_, NS_, mx_blocks_per_tile(NB_), false>(sg, 0, 100, blockIdx.x, 2560, T, ldt, rows, part, sm_); }
MX_RES(2, 4) MX_RES(4, 7) MX_RES(7, 13) MX_RES(8, 16)
#endif

}  // namespace

// dst rows: nb = ceil((k + 2) / 32) blocks of 128 bytes (ld_dst = 32 nb in 4-byte units).  scale = 2^eB, bias_scale = 2^eB log2(e),
// s8 = the segment's int8 scale (a power of two >= max |f16(src scale)| / 127).
extern "C" int jlm_pack_mixed(const float *src, int rows, int k, int ld, const float *bias, float scale, float bias_scale, float s8,
                              void *dst, int ld_dst, void *stream) {
    if (rows <= 0) return 0;
    // ld_dst = 32 ceil((k + 2) / 32): the bias rides in columns k, k + 1; = 32 ceil(k / 32) < that (k a multiple of 32): no bias columns
    // (bias ignored; jlm_vocab_lse_mixed takes the biases separately)
    if (k <= 0 || k % 4 || ld < k || ld_dst % 32 || (ld_dst / 32 != (k + 2 + 31) / 32 && ld_dst / 32 != (k + 31) / 32) || !(s8 >= 0.0f)) return -1;
    const int nb = ld_dst / 32;
    const long n = (long)rows * nb;
    if (s8 == 0.0f) {                                    // ABI 11: mx6 rows (FP6 cross-term planes with block scales; at most 8 blocks per row)
        if (nb > 8) return -1;
        hipLaunchKernelGGL(pack_mx6_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, rows, k, ld, bias,
                           scale, bias_scale, reinterpret_cast<unsigned char *>(dst), nb);
        JLM_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(pack_mixed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, rows, k, ld, bias,
                       scale, bias_scale, 1.0f / s8, reinterpret_cast<unsigned char *>(dst), nb);
    JLM_LAUNCH_CHECK();
    return 0;
}

// segs[i].B = mixed rows of the segment (jlm_pack_mixed), segs[i].ldb = 32 nb, segs[i].k the true contraction length;
// t_scale[i] = 2^eT_i (a power of two: the kernel multiplies T by t_scale log2 e and uses t_scale itself as the bias constant),
// descale[i] = 2^-(eT_i + eB_i), s8[i] = the segment's int8 scale.  Same partial-slice contract and return value as
// jlm_vocab_lse_split (one slice per column and segment it touches); -2: a shape this form does not take.
// blocks per row of a segment as its ldb says: ceil((k + 2) / 32) (bias columns) or, for k a multiple of 32, k / 32 (none); -1: neither
static int mx_seg_blocks(const jlm_segment &sg) {
    if (sg.k <= 0 || sg.ldb % 32) return -1;
    const int nb = sg.ldb / 32;
    return (nb == (sg.k + 2 + 31) / 32 || nb == (sg.k + 31) / 32) ? nb : -1;
}

// Row stride (4-byte units) of the packed rows for these segments: the segments' blocks + JLM_MAX_SEGMENTS scale floats.
// -2: more than MX_MAX_ROW_BLOCKS blocks per row in total (the packer holds a row's 16-value groups in the 64 lanes of one wave)
#define MX_MAX_ROW_BLOCKS 32
extern "C" int jlm_mixed_t_stride(const jlm_segment *segs_host, int n_segs) {
    if (n_segs < 1 || n_segs > JLM_MAX_SEGMENTS) return -1;
    int b = 0, blocks = 0;
    for (int i = 0; i < n_segs; ++i) {
        const int nb = mx_seg_blocks(segs_host[i]);
        if (nb < 0) return -1;
        b += nb * 128;
        blocks += nb;
    }
    if (blocks > MX_MAX_ROW_BLOCKS) return -2;
    return (b + 4 * JLM_MAX_SEGMENTS + 15) / 16 * 4;
}

// T [G, ldt] f32 -> Tm [G, ld_tm] packed rows, for the rows listed (rows[0 .. min(*n_dev, n_rows_max)), or 0 .. n_rows_max).
// t_scale[i] = 2^eT_i (a power of two: x = T 2^eT log2 e).
static int pack_t_mixed_impl(const jlm_segment *segs_host, const float *t_scale, int n_segs, const float *T, int ldt, const int *rows,
                             int n_rows_max, const int *n_dev, void *Tm, int ld_tm, void *stream, int fmt6) {
    if (n_segs < 1 || n_segs > JLM_MAX_SEGMENTS || ldt % 4) return -1;
    {
        const int want = jlm_mixed_t_stride(segs_host, n_segs);
        if (want == -2) return -2;
        if (ld_tm != want) return -1;
    }
    if (n_rows_max <= 0) return 0;
    MxTArgs a;
    a.n_segs = n_segs;
    int off = 0;
    for (int i = 0; i < n_segs; ++i) {
        const jlm_segment &sg = segs_host[i];
        const int nb = mx_seg_blocks(sg);
        if (nb < 1 || nb > MXW_MAX_NB || sg.k % 4 || sg.t_off % 4) return -2;
        a.seg[i] = MxTSeg{sg.k, sg.t_off, nb, off, t_scale[i] * 1.4426950408889634f, sg.k + 2 <= 32 * nb ? t_scale[i] : 0.0f};
        off += nb * 128;
    }
    if (fmt6) {
        for (int i = 0; i < n_segs; ++i) if (a.seg[i].nb > 8) return -2;      // (a row's scale bytes: eight per half)
        hipLaunchKernelGGL(pack_t_mx6_kernel, dim3((n_rows_max + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, T, ldt, rows, n_rows_max, n_dev,
                           reinterpret_cast<unsigned char *>(Tm), ld_tm);
    } else {
        hipLaunchKernelGGL(pack_t_mixed_kernel, dim3((n_rows_max + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, T, ldt, rows, n_rows_max, n_dev,
                           reinterpret_cast<unsigned char *>(Tm), ld_tm);
    }
    JLM_LAUNCH_CHECK();
    return 0;
}

extern "C" int jlm_pack_t_mixed(const jlm_segment *segs_host, const float *t_scale, int n_segs, const float *T, int ldt, const int *rows,
                                int n_rows_max, const int *n_dev, void *Tm, int ld_tm, void *stream) {
    return pack_t_mixed_impl(segs_host, t_scale, n_segs, T, ldt, rows, n_rows_max, n_dev, Tm, ld_tm, stream, 0);
}

// ABI 11: the same rows in the mx6 form (FP6 planes, halves swapped against the vocabulary rows: jlm_mx6_body.h); same stride and buffer
extern "C" int jlm_pack_t_mixed6(const jlm_segment *segs_host, const float *t_scale, int n_segs, const float *T, int ldt, const int *rows,
                                 int n_rows_max, const int *n_dev, void *Tm, int ld_tm, void *stream) {
    return pack_t_mixed_impl(segs_host, t_scale, n_segs, T, ldt, rows, n_rows_max, n_dev, Tm, ld_tm, stream, 1);
}

// the mx6 form (jlm_mx6.hip: FP6 cross terms on the block-scaled matrix instruction)
int jlm_mx6_launch(const MxArgs &a, bool xbias, int fixed_ref, const void *Tm, int ld_tm, float2 *part, int ld_part, int n_rows_max, const int *n_dev, int n_ptiles,
                   int lds, hipStream_t st);
// the wide form of the D-softmax* kernel (jlm_mixed_w.hip: four waves of 64 rows, row operands in accumulation registers)
int jlm_mx_wide_launch(int which, const MxArgs &a, const void *Tm, int ld_tm, float2 *part, int ld_part, int n_rows_max, const int *n_dev,
                       int n_ptiles, int lds, hipStream_t st);

static int vocab_lse_mixed_impl(const jlm_segment *segs_host, const float *descale, const float *s8, const float *bias2, int n_segs,
                                const void *Tm, int ld_tm, float *part, int ld_part, int max_parts, int n_rows_max,
                                const int *n_dev, void *stream, int fixed_ref) {
    const int *rows = nullptr;                       // the packed rows are compact (jlm_pack_t_mixed)
    if (n_segs < 1 || n_segs > JLM_MAX_SEGMENTS || n_rows_max <= 0 || ld_tm != jlm_mixed_t_stride(segs_host, n_segs)) return -1;
    const float *T = reinterpret_cast<const float *>(Tm);
    const int ldt = ld_tm;
    MxArgs a;
    a.n_segs = n_segs;
    static int c0x2 = -1, np8 = -1;
    // per-block constant of the cost model below; swept on the three-segment launch (kbench, C0 x prologue cost): 0.5: 78-80 us,
    // 1: 72.7-73.4, 1.5: 72.2-72.6, 2: 69.8-72.3, 3: 72.3-73.2
    if (c0x2 < 0) { const char *e = getenv("JLM_MX_C0"); c0x2 = e ? (int)(2.0 * atof(e) + 0.5) : 4; }
    // the column count of a launch that has the chip to itself may use every CU (25 columns x 10 row tiles instead of 24: 67.3 vs
    // 68.9 us, the decode 2.04 vs 2.06 ms per step); pipelined launches are capped by the CU share below that anyway.  1: multiples of 8
    if (np8 < 0) { const char *e = getenv("JLM_MX_NP8"); np8 = e ? atoi(e) : 0; }
    int ntiles[JLM_MAX_SEGMENTS];
    int rows_wg = 256;                                   // hypothesis rows per workgroup (128: the wide kernel's k = 512 form)
    // ABI 11: s8[i] == 0 for every segment = mx6 rows (jlm_mx6_body.h); one format per launch
    int n6 = 0;
    for (int i = 0; i < n_segs; ++i) n6 += s8[i] == 0.0f;
    if (n6 && n6 != n_segs) return -2;
    // (Round 6 also built 512-row workgroups -- two row sets per wave for the segments of up to four 32-k blocks, the 200-wide one walking
    //  its two 256-row halves in turn: half the fragment reads and LDS-DMA per row for 60 % of the launch.  Correct, and no faster:
    //  61-62 us against 59.8; LDS instructions -29 %, wave cycles +6 % (profiles/r06_k_pair512.txt, r06_l_pmc_mx6_forms.txt).  Removed.)
    double ctile[JLM_MAX_SEGMENTS], total = 0.0;
    long n_tiles_all = 0;
    int lds_max = 0, tm_off = 0;
    bool xbias = false;
    for (int i = 0; i < n_segs; ++i) {
        const jlm_segment &sg = segs_host[i];
        const int nb = mx_seg_blocks(sg);
        // (a single segment of sixteen blocks in the external-bias form -- k = 512: an untied model's vocabulary matrix -- runs on the
        //  wide kernel's one-row-set form, jlm_mixed_w.hip; every other shape has at most eight blocks)
        const bool k512 = n_segs == 1 && nb == 16 && sg.k == 512;
        if (nb < 1 || (nb > MX_MAX_NB && !k512) || sg.k % 4 || sg.t_off % 4) return -2;
        if ((long)(sg.v_end - sg.v_start) * nb * 128 >= (1l << 31)) return -2;        // 32-bit buffer offsets
        const bool xb = sg.k + 2 > 32 * nb;                  // no bias columns: the biases come from bias2 (base-2 units)
        if (k512) rows_wg = 128;
        if (i == 0) xbias = xb;
        if (xb != xbias || (xb && (!bias2 || nb % 2))) return -2;      // one form per launch; external-bias bodies exist for even nb
        MxSeg &m = a.seg[i];
        m.bias2 = xb ? bias2 + sg.v_start : nullptr;
        m.B = reinterpret_cast<const unsigned char *>(sg.B);
        m.n_vocab = sg.v_end - sg.v_start; m.k = sg.k; m.t_off = sg.t_off; m.nb = nb;
        m.tm_off = tm_off; m.seg = i;
        tm_off += nb * 128;
        m.descale = descale[i];
        m.cs = s8[i] * (1.0f / 2048.0f);
        const int mtt = mx_blocks_per_tile(nb);
        ntiles[i] = (m.n_vocab + 32 * mtt - 1) / (32 * mtt);
        if (ntiles[i] > 65535) return -2;
        const int ns16 = xb ? 2 * nb : (sg.k + 2 + 15) / 16;
        // cost of a tile ~ its matrix instructions (f16 steps + two int8 per 32-k block, per 32-word block) + a per-tile constant,
        // in the units of the split kernel's model (half k-steps of a 128-word tile)
        // measured (kbench, single-segment launches): a 32-word block costs 0.055 us x (its matrix instructions + ~6: combine,
        // fold, block start); here in the split kernel's units (a k-step of a 128-word tile = 12 instructions ~ 2 units)
        ctile[i] = mtt * (ns16 + 2 * nb + 3.0 * c0x2) / 6.0;
        // (per-shape block costs fitted to single-segment launches -- 2.15 : 1.28 : 1 for k = 200 / 100 / 50 -- cut the three-segment
        //  launch WORSE than this formula's 1.95 : 1.35 : 1 (72.6 vs 70.5 us); what is left between the columns is the XCDs' clocks:
        //  equal cycles per workgroup within 2 %, 1.81-1.92 GHz from XCD to XCD on one chip -- tools/probes/mixed_wg_timeline.py)
        const int lds_i = 2 * 32 * mtt * nb * 128 + (xb ? 3 * 32 * mtt * 4 : 0);
        if (lds_i > lds_max) lds_max = lds_i;
        total += ctile[i] * ntiles[i];
        n_tiles_all += ntiles[i];
    }
    const int n_ptiles = (n_rows_max + rows_wg - 1) / rows_wg;
    int cap = max_parts < MX_MAX_SUB ? max_parts : MX_MAX_SUB;
    cap -= n_segs - 1;
    if (cap > MX_MAX_PARTS) cap = MX_MAX_PARTS;
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
            if (emit) { if (cols >= MX_MAX_PARTS) return -1; a.col_first[cols] = (unsigned char)n_sub; }
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
                    if (n_sub >= MX_MAX_SUB) return -1;
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
    const int lds = lds_max;
    if (n6) {
        if (rows_wg == 128) return -2;
        // (the mx6 forms without a running maximum take the accumulators as base-2 logits: only for launches whose descale is 1)
        int fr6 = fixed_ref;
        for (int i = 0; i < n_segs; ++i) fr6 &= descale[i] == 1.0f;
        if (int rc = jlm_mx6_launch(a, xbias, fr6, Tm, ld_tm, reinterpret_cast<float2 *>(part), ld_part, n_rows_max, n_dev, n_ptiles, lds,
                                    (hipStream_t)stream)) return rc;
        return n_sub;
    }
    // which kernel: 0 the D-softmax* shapes (inlined), 1 any other bias-column shape, 2 tied k = 256 (inlined), 3 other external-bias shapes
    int which = xbias ? 2 : 0;
    for (int i = 0; i < n_segs; ++i) {
        const int nb = a.seg[i].nb, ns16 = (a.seg[i].k + 2 + 15) / 16;
        if (xbias) { if (nb != 8) which = 3; }
        else if (!((nb == 7 && ns16 == 13) || (nb == 4 && ns16 == 7) || (nb == 2 && ns16 == 4))) which = 1;
    }
    float2 *part2 = reinterpret_cast<float2 *>(part);
    hipStream_t st = (hipStream_t)stream;
    static int wide = -1;
    if (wide < 0) { const char *e = getenv("JLM_MX_WIDE"); wide = e ? atoi(e) : JLM_MX_WIDE_DEFAULT; }
    // fixed_ref (jlm_vocab_lse_mixed_fr): the wide kernel's forms WITHOUT a running maximum -- s = sum 2^y against the reference 0, slices
    // (0, s) -- where a form exists (tied k = 256, k = 512); any other shape runs as usual.  -3 % on those launches (113.5 vs 117 us,
    // 1 589 vs 1 634 us at configs[2]'s shape: profiles/r05_u_fixed_ref.txt).  Valid while a row's largest base-2 logit stays within
    // (The eight-wave kernel's D-softmax* bodies were built in this form too and measured SLOWER -- 68.9-73.8 vs 67.0 us per launch,
    //  profiles/r05_v_fixed_ref_dsoftmax.txt -- and left as they were: that launch ignores the flag.)
    // +-100 or so (f32 range, 2^16 words): the caller's decision (DeviceModel measures its model at load); a row outside it yields s = 0 or
    // inf, never a plausible number.
    const int fixref = fixed_ref;
    if (rows_wg == 128) {
        if (int rc = jlm_mx_wide_launch(fixref ? 3 : 1, a, Tm, ld_tm, part2, ld_part, n_rows_max, n_dev, n_ptiles, lds, st)) return rc;
        return n_sub;
    }
    // JLM_MX_WIDE: 1 the wide kernel (jlm_mixed_w.hip) for every shape it hosts, 0 never, -1 (default) where it measures faster: the tied
    // k = 256 shapes -- 116.6-118.0 vs 122.7-124.5 us at V = 50 k / 2 560 rows, 1 691 vs 1 813 us at V = 100 k / 20 480 rows; the
    // D-softmax* launch measures the same on both (70.5 vs 70.0 us) and stays on the eight-wave kernel (profiles/r05_r_wide_tied.txt)
    if ((which == 0 && wide > 0) || (which == 2 && wide != 0)) {
        if (int rc = jlm_mx_wide_launch(which == 2 && fixref ? 4 : which, a, Tm, ld_tm, part2, ld_part, n_rows_max, n_dev, n_ptiles, lds, st)) return rc;
        return n_sub;
    }
    static JlmLdsGrant grant[4];
    const void *fns[4] = {reinterpret_cast<const void *>(MX_KERNEL_DSOFTMAX), reinterpret_cast<const void *>(MX_KERNEL_GENERIC),
                          reinterpret_cast<const void *>(MX_KERNEL_TIED), reinterpret_cast<const void *>(MX_KERNEL_GENERIC_XB)};
    if (int rc = jlm_grant_lds(grant[which], fns[which], lds)) return rc;
    const dim3 grid(n_cols * n_ptiles), block(512);
    switch (which) {
    case 0: hipLaunchKernelGGL(MX_KERNEL_DSOFTMAX, grid, block, lds, st, a, T, ldt, rows, part2, ld_part, n_rows_max, n_dev, n_ptiles); break;
    case 1: hipLaunchKernelGGL(MX_KERNEL_GENERIC, grid, block, lds, st, a, T, ldt, rows, part2, ld_part, n_rows_max, n_dev, n_ptiles); break;
    case 2: hipLaunchKernelGGL(MX_KERNEL_TIED, grid, block, lds, st, a, T, ldt, rows, part2, ld_part, n_rows_max, n_dev, n_ptiles); break;
    default: hipLaunchKernelGGL(MX_KERNEL_GENERIC_XB, grid, block, lds, st, a, T, ldt, rows, part2, ld_part, n_rows_max, n_dev, n_ptiles); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(int)e - 100;
    return n_sub;
}

extern "C" int jlm_vocab_lse_mixed(const jlm_segment *segs_host, const float *descale, const float *s8, const float *bias2, int n_segs,
                                   const void *Tm, int ld_tm, float *part, int ld_part, int max_parts, int n_rows_max,
                                   const int *n_dev, void *stream) {
    return vocab_lse_mixed_impl(segs_host, descale, s8, bias2, n_segs, Tm, ld_tm, part, ld_part, max_parts, n_rows_max, n_dev, stream, 0);
}

extern "C" int jlm_vocab_lse_mixed_fr(const jlm_segment *segs_host, const float *descale, const float *s8, const float *bias2, int n_segs,
                                      const void *Tm, int ld_tm, float *part, int ld_part, int max_parts, int n_rows_max,
                                      const int *n_dev, void *stream) {
    return vocab_lse_mixed_impl(segs_host, descale, s8, bias2, n_segs, Tm, ld_tm, part, ld_part, max_parts, n_rows_max, n_dev, stream, 1);
}
