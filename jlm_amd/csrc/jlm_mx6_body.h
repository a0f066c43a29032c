// jlm_mx6_body.h -- round 6: the mixed-row vocabulary kernel with the split product's cross terms on the BLOCK-SCALED matrix
// instruction (v_mfma_scale_f32_32x32x64_f8f6f4, FP6 e2m3 operands, one E8M0 scale per 32 k-values of every row), accumulated
// into the SAME f32 accumulator as the f16 hi.hi pass.  Reference: project + softmax, decoder/model.py:141-193, 15-20.
//
//      t.b  ~  t_hi.b_hi                       f16 x f16, exact products, f32 accumulate   (v_mfma_f32_32x32x16_f16, 16 k per instruction)
//            + t_lo6.b_hi6 + t_hi6.b_lo6       FP6 x FP6, K = 64 = the 32 k-values of BOTH cross terms in ONE instruction:
//                                              A (words)  k-slots 0-31 = hi6, 32-63 = lo6;  B (rows)  k-slots 0-31 = lo6, 32-63 = hi6
// Against the int8 form (jlm_mixed.hip): 3 matrix instructions per 32 k-values instead of 4 (96 instead of 128 cycles of the matrix
// pipe: 0.82 of the time on random operands with the clock the governor grants, profiles/r06_a_fp6_probe.txt), 3.5 KB of LDS fragments
// instead of 4, no integer accumulator and no per-logit convert-and-combine (2 of ~6.5 VALU instructions per logit), 14 + 16 registers
// per lane less.  Error: e2m3 has 4 significant bits where int8 has 7 of the ROW / SEGMENT maximum -- but relative to its own block of
// 32: Gaussian-like blocks lose a factor 1 .. 1.8 on the log-normaliser (3e-10 -> 5.5e-10 at BASELINE configs[1], gate 1e-6), heavy-tailed
// blocks GAIN 6 .. 10 x (tests/mx6_emu.py, tests/test_mx6_emulation.py).  The loader measures the form on the model like the int8 one.
//
// Row format ("mx6 rows"): the 128-byte block per 32 k-values of the mixed rows, granules of 16 bytes:
//      0-3   32 x f16 hi (the bias columns k, k + 1 as in jlm_mixed.hip)
//      4     half 0: FP6 codes of k-values 0 .. 20 (bytes 0-15 of its 24: value i at bits [6 i, 6 i + 6))
//      5     [ half 0 bytes 16-23 | half 1 bytes 16-23 ]
//      6     half 1 bytes 0-15
//      7     block 0 of a row only: [ E8M0 scale of half 0 for blocks 0 .. 7 | ... of half 1 ]      (other blocks: unused)
// vocabulary rows: half 0 = hi6 (FP6 of the f16 hi), half 1 = lo6 (FP6 of the residual); packed hypothesis rows: half 0 = lo6, half 1 = hi6
// -- lane (row l & 31, half l >> 5) of either operand reads granule 4 + 2 half, half of granule 5 and its 8 scale bytes.
// Same tiles, LDS image (source-swizzled LDS-DMA), column cuts and partial (max, sum) slices as the int8 kernel.
#pragma once
#include "jlm_mixed_body.h"

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

namespace jlm_mx {

// ------------------------------------------------------------------------------------------------ FP6 packing (device)
// E8M0 byte of a block: the smallest power of two s = 2^(byte - 127) with amax <= 7.5 s (0 for an all-zero block)
__device__ __forceinline__ int mx6_block_byte(float amax) {
    if (!(amax > 0.0f)) return 0;
    const int bits = __float_as_int(amax);
    const int e0 = ((bits >> 23) & 0xff) - 127;                  // amax = m 2^e0, m in [1, 2)
    const int e = (bits & 0x007fffff) <= 0x00700000 ? e0 - 2 : e0 - 1;      // m <= 1.875: 7.5 2^(e0 - 2) >= amax
    return min(max(e + 127, 0), 254);
}
// code of x / 2^(byte - 127): round to nearest even on the e2m3 grid (steps 0.125 below 2, 0.25 below 4, 0.5 up to 7.5), saturating
// (below 2 the code IS 8 x the value -- the subnormal codes 0 .. 7 and the first binade 8 .. 15 share the step 0.125; in [2, 4): 8 + 4 a; in
//  [4, 7.5]: 16 + 2 a; a value that rounds up to the next binade's first grid point gets that point's code from the same formula)
__device__ __forceinline__ unsigned mx6_code(float x, int byte) {
    const float a = fminf(fabsf(__builtin_ldexpf(x, 127 - byte)), 7.5f);
    const float mul = a < 2.0f ? 8.0f : (a < 4.0f ? 4.0f : 2.0f);
    const float off = a < 2.0f ? 0.0f : (a < 4.0f ? 8.0f : 16.0f);
    const unsigned mag = (unsigned)(rintf(a * mul) + off);
    return mag | ((__float_as_uint(x) >> 26) & 32u);
}
// 16 codes -> 96 bits
__device__ __forceinline__ void mx6_pack16(const unsigned *c, unsigned w[3]) {
    w[0] = w[1] = w[2] = 0u;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int bit = 6 * e;
        w[bit >> 5] |= c[e] << (bit & 31);
        if ((bit & 31) > 26) w[(bit >> 5) + 1] |= c[e] >> (32 - (bit & 31));
    }
}

// ------------------------------------------------------------------------------------------------ the kernel body
// one sub-range (tiles [vt0, vt1) of one segment) for this workgroup's 256 rows; NB 32-k blocks, NS16 f16 steps (2 NB or 2 NB - 1),
// MTT 32-word blocks per tile (even), XBIAS as mx_body.  TWO accumulators: block n + 1 multiplies into one while block n's logits
// are folded straight out of the other -- there is nothing left to combine (no integer accumulator), so no v[] and no copy.
// FR (fixed reference, jlm_vocab_lse_mixed_fr): no running maximum -- s = sum 2^y against the reference 0, slices (0, s) -- for launches whose
// accumulators ARE base-2 logits (descale = 1: the loader scales mx6 operands so, DeviceModel._build_mixed): the fold is exp2 + add per
// logit, 20 of the max form's 26 VALU cycles (the VALU's share of this kernel is 40 % of its time: profiles/r06_l_pmc_mx6_forms.txt).
template <int NB, int NS16, int MTT, bool XBIAS = false, bool FR = false>
__device__ __forceinline__ void mx6_body(const MxSeg &sg, int vt0, int vt1, int pt, int n_paths, const unsigned char *__restrict__ Tm, int ld_tm,
                                         float2 *__restrict__ part_row, unsigned char *smem) {
    static_assert(MTT % 2 == 0, "the accumulators alternate by 32-word block: an even number per tile");
    constexpr float LN2 = 0.6931471805599453f;
    constexpr int ROWB = NB * 128;
    constexpr int TW = 32 * MTT;
    constexpr int BUFB = TW * ROWB;
    constexpr int BIAS_OFF = 2 * BUFB;
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = lane >> 5, li = lane & 31;
    // ---- 1. this lane's row operands (packed by jlm_pack_t_mixed6: granule-major, one contiguous half-kilobyte per load and half)
    const int prow = pt * 256 + wave * 32 + li;
    const bool row_ok = prow < n_paths;
    const unsigned char *tb0 = Tm + (row_ok ? mx_tm_block(prow, ld_tm) : 0) + mx_tm_granule(sg.tm_off, 0, prow);
    f16x8 thi[NS16];
    i32x8 t6[NB];
    i32x2 tsc;
    {
        i32x4 raw[NS16], ra[NB];
        i32x2 rb[NB];
#pragma unroll
        for (int q = 0; q < NS16; ++q) raw[q] = *reinterpret_cast<const i32x4 *>(tb0 + ((q >> 1) * 8 + 2 * (q & 1) + hf) * 512);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            ra[j] = *reinterpret_cast<const i32x4 *>(tb0 + (j * 8 + 4 + 2 * hf) * 512);
            rb[j] = *reinterpret_cast<const i32x2 *>(tb0 + (j * 8 + 5) * 512 + 8 * hf);
        }
        tsc = *reinterpret_cast<const i32x2 *>(tb0 + 7 * 512 + 8 * hf);
        const i32x4 z = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < NS16; ++q) thi[q] = __builtin_bit_cast(f16x8, row_ok ? raw[q] : z);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const i32x4 a = row_ok ? ra[j] : z;
            t6[j] = i32x8{a[0], a[1], a[2], a[3], row_ok ? rb[j][0] : 0, row_ok ? rb[j][1] : 0, 0, 0};
        }
    }
    const float descale = sg.descale;

    // ---- 2. LDS-DMA of a tile (as mx_body: wave w fills row group w of every block j; the swizzle sits on the source)
    const unsigned long long bptr = reinterpret_cast<unsigned long long>(sg.B);
    const unsigned long long bptr_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bptr >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)bptr);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(bptr_u), 0,
                                                                          __builtin_amdgcn_readfirstlane(sg.n_vocab) * ROWB, 0x00020000);
    const int r8 = lane >> 3, dslot = lane & 7;
    const int drow = 8 * wave + r8;
    const int dvoff = drow * ROWB + ((dslot ^ ((drow >> 1) & 7)) * 16);
    constexpr int NRG = MTT / 2;
    __amdgpu_buffer_rsrc_t rs_bias = rs_b;
    if (XBIAS) {
        const unsigned long long p2 = reinterpret_cast<unsigned long long>(sg.bias2);
        const unsigned long long p2u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(p2 >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)p2);
        rs_bias = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(p2u), 0, __builtin_amdgcn_readfirstlane(sg.n_vocab) * 4, 0x00020000);
    }
    auto issue_bias = [&](int t) {
        if (XBIAS && wave == 0) {
#pragma unroll
            for (int i = 0; i < (TW + 63) / 64; ++i)
                if (i * 64 + 64 <= TW || lane < TW - i * 64)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_bias, (__attribute__((address_space(3))) void *)(smem + BIAS_OFF + (t % 3) * (TW * 4) + i * 256),
                                                             4, (t * TW + i * 64 + lane) * 4, 0, 0, 0);
        }
    };
    constexpr int NDMA = NRG * NB;
    constexpr int NMB = 3;                               // issue slots per 32-k block: f16, FP6, f16
    constexpr bool SPREAD = MX_DMA_SPREAD && XBIAS && NDMA <= NMB * (NB - 1) + (NS16 == 2 * NB ? 3 : 2);
    auto issue_rows = [&](int t, int buf) {
        const int voff = dvoff + t * (TW * ROWB);
#pragma unroll
        for (int i = 0; i < NRG; ++i) {
            unsigned char *dst = smem + buf * BUFB + ((wave + 8 * i) * NB) * 1024;
#pragma unroll
            for (int j = 0; j < NB; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16,
                                                         voff + i * (64 * ROWB), j * 128, 0, 0);
        }
    };
    auto issue = [&](int t, int buf) { issue_bias(t); issue_rows(t, buf); };
    auto issue_piece = [&](int t, int buf, int q) {
        const int i = q / NB, j = q % NB;
        unsigned char *dst = smem + buf * BUFB + ((wave + 8 * i) * NB) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16,
                                                 dvoff + t * (TW * ROWB) + i * (64 * ROWB), j * 128, 0, 0);
    };
    // fragment addresses: block mt, row li: piece (4 mt + li / 8, j); inside it row li % 8, source granule g at slot g ^ ((li >> 1) & 7)
    const int x = (li >> 1) & 7;
    const int fbase = (li >> 3) * (NB * 1024) + (li & 7) * 128;
    const int g_f0 = fbase + ((0 + hf) ^ x) * 16, g_f1 = fbase + ((2 + hf) ^ x) * 16;     // f16 steps 2 J, 2 J + 1
    const int g_6a = fbase + ((4 + 2 * hf) ^ x) * 16, g_6b = fbase + (5 ^ x) * 16 + 8 * hf;   // this half's FP6 codes: 16 + 8 bytes
    const int g_sc = fbase + (7 ^ x) * 16 + 8 * hf;                                     // ... and its scales (32-k block 0 of the row)

    float m = FR ? 0.0f : JLM_NEG_BIG, s = 0.0f;
    // the accumulator that is "finished" before the first block: sixteen -1e30 logits -- their fold leaves (m, s) = (very negative, 16),
    // which the first real fold scales to 0 (as mx_body's v[])
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.0f; acc[1][r] = -1.0e30f; }
    float tmax, nmn, sc_old, add0, add1;
    constexpr int NMF = NS16 + NB;                       // matrix instructions of a block
    constexpr int NPIECE = FR ? 16 + 1 : 8 + 1 + 16 + 1;   // 8 x max3, 1, 16 x exp, 1   (FR: 16 x exp, 1)
    constexpr int PP = (NPIECE + NMF - 1) / NMF;
    // (XBIAS: the finished accumulator holds base-2 logits after the bias burst; else raw accumulator units, descaled inside the fold)
    auto fold_piece = [&](const f32x16 &pf, int pc) {
        if (MX_ABL & 1) { if (pc == 0) asm volatile("" :: "v"(pf)); return; }
        if (FR) {
            if (pc < 16) {
                const float e = __builtin_amdgcn_exp2f(pf[pc]);
                if (pc == 0) { add0 = e; add1 = 0.0f; } else if (pc & 1) add1 += e; else add0 += e;
            } else if (pc == 16) {
                s += add0 + add1;
            }
            return;
        }
        if (pc < 8) {
            tmax = pc == 0 ? fmaxf(pf[0], pf[1]) : fmaxf(fmaxf(tmax, pf[2 * pc]), pf[2 * pc + 1]);      // (v_max3_f32)
        } else if (pc == 8) {
            const float mn = fmaxf(m, XBIAS ? tmax : tmax * descale);
            nmn = -mn;
            sc_old = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            add0 = 0.0f; add1 = 0.0f;
        } else if (pc < 25) {
            const int r = pc - 9;
            const float e = __builtin_amdgcn_exp2f(XBIAS ? pf[r] + nmn : fmaf(pf[r], descale, nmn));
            if (r & 1) add1 += e; else add0 += e;
        } else if (pc == 25) {
            s = fmaf(s, sc_old, add0 + add1);
        }
    };
    issue(vt0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int buf = 0;
    int lim_acc = 1 << 30, mt_acc = 0, t_acc = vt0;      // the finished block: of tile t_acc with lim_acc valid words, its block mt_acc
    // the burst in front of a block's matrix instructions: the finished block's biases (XBIAS) and, in a segment's last partial tile,
    // the mask of the words past its end
    auto finish = [&](auto masked_c, f32x16 &pf) {
        constexpr bool MASKED = decltype(masked_c)::value != 0;
        if (!XBIAS && !MASKED) return;
        f32x4 bq[4];
        if (XBIAS) {
            const unsigned char *bp = smem + BIAS_OFF + (t_acc % 3) * (TW * 4) + (mt_acc * 32 + 4 * hf) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4 *>(bp + q * 32);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y = pf[r];
            if (XBIAS) y = FR ? y + bq[r >> 2][r & 3] : fmaf(y, descale, bq[r >> 2][r & 3]);
            pf[r] = (MASKED && mt_acc * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf >= lim_acc) ? JLM_NEG_BIG : y;
        }
    };
    auto tile = [&](auto masked_c, int t) {
        if (!SPREAD) { if (!(MX_ABL & 4) && t + 1 < vt1) issue(t + 1, buf ^ 1); }
        else if (!(MX_ABL & 4) && t + 1 < vt1) issue_bias(t + 1);
        const int lim = sg.n_vocab - t * TW;
        // fragments of 32-k block j: F0 / F1 the f16 granules of steps 2 j, 2 j + 1, F6 this half's FP6 codes (6 registers), each
        // refilled in place with block j + 1's (or the next 32-word block's first) right behind the instruction that read it; fsc: the
        // scale bytes of the 32-word block's rows (all 32-k blocks), refilled behind the block's last FP6 instruction
        i32x4 F0, F1;
        i32x8 F6;
        i32x2 fsc;
        {
            const unsigned char *bs0 = smem + buf * BUFB;
            F0 = *reinterpret_cast<const i32x4 *>(bs0 + g_f0);
            const i32x4 a = *reinterpret_cast<const i32x4 *>(bs0 + g_6a);
            const i32x2 b = *reinterpret_cast<const i32x2 *>(bs0 + g_6b);
            fsc = *reinterpret_cast<const i32x2 *>(bs0 + g_sc);
            F1 = *reinterpret_cast<const i32x4 *>(bs0 + g_f1);
            F6 = i32x8{a[0], a[1], a[2], a[3], b[0], b[1], 0, 0};
        }
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) {
            const unsigned char *bs = smem + buf * BUFB + mt * (4 * NB * 1024);
            f32x16 &wf = acc[mt & 1];                    // this block's accumulator
            f32x16 &pf = acc[(mt & 1) ^ 1];              // the finished one (the block before; the tile before's last for mt = 0)
            finish(masked_c, pf);
            __builtin_amdgcn_sched_barrier(0);
            const f32x16 zf = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            i32x2 fsc_next = fsc;
            mx_for_each_ic([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                constexpr bool second = 2 * J + 1 < NS16;
                constexpr bool rdm = (J + 1 < NB);
                const bool more = rdm || mt + 1 < MTT;              // a next 32-k block to read exists in this tile
                const unsigned char *nx = rdm ? bs + (J + 1) * 1024 : bs + (4 * NB * 1024);
                auto pieces = [&](int q) {
#pragma unroll
                    for (int pc = q * PP; pc < (q + 1) * PP && pc < NPIECE; ++pc) fold_piece(pf, pc);
                };
                auto dma = [&](int i) {
                    if (SPREAD && !(MX_ABL & 4) && mt == 0 && NMB * J + i < NDMA) issue_piece(t + 1, buf ^ 1, NMB * J + i);
                };
                wf = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, F0), thi[2 * J], J == 0 ? zf : wf, 0, 0, 0);
                dma(0);
                if (more && !(MX_ABL & 32)) F0 = *reinterpret_cast<const i32x4 *>(nx + g_f0);
                pieces(NMB * J);
                wf = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(F6, t6[J], wf, 2, 2, J & 3, fsc[J >> 2], J & 3, tsc[J >> 2]);
                dma(1);
                if (more && !(MX_ABL & 32)) {
                    const i32x4 a = *reinterpret_cast<const i32x4 *>(nx + g_6a);
                    const i32x2 b = *reinterpret_cast<const i32x2 *>(nx + g_6b);
                    F6 = i32x8{a[0], a[1], a[2], a[3], b[0], b[1], 0, 0};
                    if (!rdm) fsc_next = *reinterpret_cast<const i32x2 *>(nx + g_sc);
                }
                pieces(NMB * J + 1);
                if constexpr (second) {
                    wf = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, F1), thi[second ? 2 * J + 1 : 0], wf, 0, 0, 0);
                    dma(2);
                }
                if (more && !(MX_ABL & 32)) F1 = *reinterpret_cast<const i32x4 *>(nx + g_f1);
                pieces(NMB * J + 2);
                // issue order of the block: matrix instruction, [the LDS-DMA instruction], the read(s) behind it, its share of the fold
                constexpr int NM = second ? 3 : 2;
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (SPREAD && !(MX_ABL & 4) && mt == 0 && NMB * J + i < NDMA) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    if ((rdm || mt + 1 < MTT) && !(MX_ABL & 32)) {
                        if (i != 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        else if (rdm) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    }
                    if (!(MX_ABL & 1)) __builtin_amdgcn_sched_group_barrier(0x002, 3 * PP, 0);
                }
                if constexpr (!second) { if ((rdm || mt + 1 < MTT) && !(MX_ABL & 32)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            }, std::make_integer_sequence<int, NB>{});
            __builtin_amdgcn_sched_barrier(0);
            fsc = fsc_next;
            // pieces the block's instruction count did not reach (short contractions)
#pragma unroll
            for (int pc = NMB * NB * PP; pc < NPIECE; ++pc) fold_piece(pf, pc);
            lim_acc = lim; mt_acc = mt; t_acc = t;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(MX_ABL & 8)) __builtin_amdgcn_s_barrier();
        buf ^= 1;
    };
    const int t_full = min(vt1, sg.n_vocab / TW);
    for (int t = vt0; t < t_full; ++t) tile(IC<0>{}, t);
    for (int t = max(vt0, t_full); t < vt1; ++t) tile(IC<1>{}, t);
    // the last block (MTT even: it sits in accumulator 1): finish, fold
    finish(IC<1>{}, acc[1]);
#pragma unroll
    for (int pc = 0; pc < NPIECE; ++pc) fold_piece(acc[1], pc);
    const float m2 = __shfl_xor(m, 32), s2 = __shfl_xor(s, 32);
    if (FR) {
        s += s2;
    } else {
        const float mm = fmaxf(m, m2);
        s = s * __builtin_amdgcn_exp2f(m - mm) + s2 * __builtin_amdgcn_exp2f(m2 - mm);
        m = mm * LN2;
    }
    if (hf == 0 && row_ok) part_row[prow] = make_float2(m, s);
}

}  // namespace jlm_mx
