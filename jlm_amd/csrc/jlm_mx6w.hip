// jlm_mx6w.hip -- round 6: the mx6 vocabulary kernel (jlm_mx6_body.h: f16 hi.hi + both cross terms in one block-scaled FP6 instruction
// per 32 k-values) in the WIDE form of jlm_mixed_w.hip: four waves per workgroup, one per SIMD, each keeping 64 hypothesis rows -- two
// 32-row sets -- so that every vocabulary fragment read from LDS feeds two matrix instructions.
// Reference: project + softmax, decoder/model.py:141-193, 15-20.
//
// Why (profiles/r06_f_mx6_ablate.txt, the eight-wave mx6 kernel at BASELINE configs[1], 60.3 us): the launch is the SUM of its parts --
// matrix instructions 25.3 us, fold 8.5, LDS-DMA issue 7, barriers 2.5 and 18 us of exposed LDS fragment reads: at 32 rows per wave the
// three instructions of a 32-k block consume 3.5 KB of fragments in 96 cycles and SIMD = 149 B per cycle and CU, more than the LDS
// delivers (128).  Two row sets halve the reads.  Their operands -- 2 x (52 + 42) registers per lane at k = 200 -- do not fit two waves
// per SIMD: one wave per SIMD owns the whole register file, the f16 row operands in its ACCUMULATION registers (legal MFMA B operands,
// loaded straight into them), the FP6 row operands (6-register tuples), the accumulators and the fragment ring in the architectural ones.
// Compiled with -mllvm -amdgpu-mfma-vgpr-form (accumulators stay out of the accumulation file) -fno-honor-nans -mno-amdgpu-ieee.
//
// One in-order issuer: fragments are requested a whole 32-k block (>= 192 cycles of matrix instructions) ahead into a ring of two; two
// accumulators per row set -- block n + 1 multiplies into one while block n's logits are folded out of the other, a few VALU
// instructions behind each matrix instruction; the next tile's LDS-DMA instructions ride one per 32-k block.
// Same tiles, LDS image, column cuts and partial (max, sum) slices as the eight-wave kernels: jlm_vocab_lse_mixed launches either.
#include "jlm_common.h"
#include <type_traits>
#include <utility>

#include "jlm_mx6_body.h"
using namespace jlm_mx;

#ifndef MX6W_ABL
#define MX6W_ABL 0      // measurement builds (wrong numbers): 1 no fold, 4 no LDS-DMA in the loop, 8 no barrier, 32 no fragment reads in the loop
#endif

namespace {

// NB 32-k blocks per row, NS16 f16 steps, MTT 32-word blocks per tile (even), XB: external biases (k a multiple of 32), FR: fixed
// reference -- s = sum 2^y against 0, no running maximum (jlm_vocab_lse_mixed_fr: the loader's decision), for launches with descale = 1
template <int NB, int NS16, int MTT, bool XB, bool FR>
struct Mx6Wide {
    static_assert(MTT % 2 == 0, "the accumulators alternate by 32-word block");
    static constexpr int RS = 2;
    static constexpr int ROWB = NB * 128;
    static constexpr int TW = 32 * MTT;
    static constexpr int BUFB = TW * ROWB;
    static constexpr int BIAS_OFF = 2 * BUFB;
    static constexpr int NSLOT = 3 * RS * NB;             // issue slots of a block: (f16, FP6, f16) x row sets x 32-k blocks
    static constexpr int SKIP = 2;                        // (the first slots carry nothing: the finished accumulators' last instructions are still in the pipe)

    f16x8 thi[RS][NS16];                                  // accumulation registers
    i32x8 t6[RS][NB];                                     // 6 registers each
    i32x2 tsc[RS];
    float descale;
    float m[RS], s[RS], tmax[RS], nmn[RS], sc_old[RS], add0[RS], add1[RS];
    f32x16 acc[2][RS];
    f32x4 bq[4];
    i32x4 F0[2], F1[2];
    i32x8 F6[2];
    i32x2 fsc, fsc_next;
    int g_f0, g_f1, g_6a, g_6b, g_sc;
    int hf;
    unsigned char *smem;

    // pieces of the treatment of a finished accumulator pf of row set S.  FIN: a finishing step per logit first (XB: y = acc descale + bias;
    // MASKED: words past the segment's end).  then FR: 16 x (exp2, add), 1;  else 8 x max3, 1, 16 x (scale, exp2, add), 1
    template <bool MASKED>
    static constexpr int np1() { return ((XB || MASKED) ? 16 : 0) + (FR ? 17 : 26); }

    template <bool MASKED>
    __device__ __forceinline__ void fold_piece(int S, f32x16 &pf, int mtp, int lim, int pc) {
#pragma clang fp contract(off)          // every fused multiply-add below is written as one: the row sets must round alike
        constexpr bool FIN = XB || MASKED;
        if (FIN) {
            if (pc < 16) {
                const int r = pc;
                float y = pf[r];
                if (XB) y = FR ? y + bq[r >> 2][r & 3] : fmaf(y, descale, bq[r >> 2][r & 3]);
                pf[r] = (MASKED && mtp * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf >= lim) ? JLM_NEG_BIG : y;
                return;
            }
            pc -= 16;
        }
        if (FR) {
            if (pc < 16) {
                const float e = __builtin_amdgcn_exp2f(pf[pc]);             // (FR launches have descale = 1: the accumulators are base-2 logits)
                if (pc == 0) { add0[S] = e; add1[S] = 0.0f; } else if (pc & 1) add1[S] += e; else add0[S] += e;
            } else if (pc == 16) {
                s[S] += add0[S] + add1[S];
            }
            return;
        }
        if (pc < 8) {
            tmax[S] = pc == 0 ? fmaxf(pf[0], pf[1]) : fmaxf(fmaxf(tmax[S], pf[2 * pc]), pf[2 * pc + 1]);
        } else if (pc == 8) {
            const float mn = fmaxf(m[S], XB ? tmax[S] : tmax[S] * descale);
            nmn[S] = -mn;
            sc_old[S] = __builtin_amdgcn_exp2f(m[S] - mn);
            m[S] = mn;
            add0[S] = 0.0f; add1[S] = 0.0f;
        } else if (pc < 25) {
            const int r = pc - 9;
            const float e = __builtin_amdgcn_exp2f(XB ? pf[r] + nmn[S] : fmaf(pf[r], descale, nmn[S]));
            if (r & 1) add1[S] += e; else add0[S] += e;
        } else if (pc == 25) {
            s[S] = fmaf(s[S], sc_old[S], add0[S] + add1[S]);
        }
    }
    // pieces alternate between the row sets: piece RS q + S -> set S piece q
    template <bool MASKED>
    __device__ __forceinline__ void fold2(const int pw, int mtp, int lim, int pc2) {
        f32x16 (&pf)[RS] = acc[pw];
        if (MX6W_ABL & 1) {
            if (pc2 == 0) { asm volatile("" :: "v"(pf[0])); asm volatile("" :: "v"(pf[RS - 1])); }
            return;
        }
        fold_piece<MASKED>(pc2 % RS, pf[pc2 % RS], mtp, lim, pc2 / RS);
    }
    __device__ __forceinline__ void load_bias(int boff) {
        if (!XB) return;
        const unsigned char *bp = smem + boff + (4 * hf) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4 *>(bp + q * 32);
    }

    // one 32-word block: its matrix instructions into accumulator W of every set; accumulator 1 - W (finished) treated between them; tile
    // t + 1's LDS-DMA instruction (mt NB + J) behind the first matrix instruction of 32-k block J
    template <bool MASKED, int mt, int W>
    __device__ __forceinline__ void block(int buf, int lim_p, int boff_p, const __amdgpu_buffer_rsrc_t rs_next, int voff_next, int wave) {
        constexpr int mtp = (mt + MTT - 1) % MTT;
        constexpr int NPIECE = RS * np1<MASKED>();
        constexpr int PP = (NPIECE + NSLOT - SKIP - 1) / (NSLOT - SKIP);
        f32x16 (&wf)[RS] = acc[W];
        const unsigned char *bs = smem + buf * BUFB + mt * (4 * NB * 1024);
        load_bias(boff_p);
        __builtin_amdgcn_sched_barrier(0);
        const f32x16 zf = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        mx_for_each_ic([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            constexpr bool second = 2 * J + 1 < NS16;
            constexpr int R = (mt * NB + J) & 1, Rn = R ^ 1;                      // this 32-k block's ring half, the next one's
            constexpr bool rdm = J + 1 < NB;
            constexpr bool reads = (rdm || mt + 1 < MTT) && !(MX6W_ABL & 32);
            const unsigned char *nx = rdm ? bs + (J + 1) * 1024 : bs + (4 * NB * 1024);
            auto pieces = [&](int q) {
                if (q < SKIP) return;
#pragma unroll
                for (int pc = (q - SKIP) * PP; pc < (q - SKIP + 1) * PP && pc < NPIECE; ++pc) fold2<MASKED>(1 - W, mtp, lim_p, pc);
            };
            if (!(MX6W_ABL & 4)) {
                constexpr int q = mt * NB + J, i = q / NB, j = q % NB;
                unsigned char *dst = smem + (buf ^ 1) * BUFB + ((wave + 4 * i) * NB) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_next, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16,
                                                         voff_next + i * (32 * ROWB), j * 128, 0, 0);
            }
            constexpr int Q0 = 3 * RS * J;
            mx_for_each_ic([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                wf[S] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, F0[R]), thi[S][2 * J], J == 0 ? zf : wf[S], 0, 0, 0);
                if (S == RS - 1 && reads) F0[Rn] = *reinterpret_cast<const i32x4 *>(nx + g_f0);
                pieces(Q0 + S);
            }, std::make_integer_sequence<int, RS>{});
            mx_for_each_ic([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                wf[S] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(F6[R], t6[S][J], wf[S], 2, 2, J & 3, fsc[J >> 2], J & 3, tsc[S][J >> 2]);
                if (S == RS - 1 && reads) {
                    const i32x4 a = *reinterpret_cast<const i32x4 *>(nx + g_6a);
                    const i32x2 b = *reinterpret_cast<const i32x2 *>(nx + g_6b);
                    F6[Rn] = i32x8{a[0], a[1], a[2], a[3], b[0], b[1], 0, 0};
                    if (!rdm) fsc_next = *reinterpret_cast<const i32x2 *>(nx + g_sc);
                }
                pieces(Q0 + RS + S);
            }, std::make_integer_sequence<int, RS>{});
            mx_for_each_ic([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                if constexpr (second) wf[S] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, F1[R]), thi[S][second ? 2 * J + 1 : 0], wf[S], 0, 0, 0);
                if (S == RS - 1 && reads) F1[Rn] = *reinterpret_cast<const i32x4 *>(nx + g_f1);
                pieces(Q0 + 2 * RS + S);
            }, std::make_integer_sequence<int, RS>{});
            // issue order: matrix instruction, [the LDS-DMA instruction], [fragment reads], its share of the fold
#pragma unroll
            for (int i = 0; i < 3 * RS; ++i) {
                const bool has_m = second || (i / RS != 2);
                if (has_m) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i == 0 && !(MX6W_ABL & 4)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if ((i % RS) == RS - 1 && reads) {
                    if (i / RS != 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    else if (rdm) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                }
                if (Q0 + i >= SKIP && !(MX6W_ABL & 1)) __builtin_amdgcn_sched_group_barrier(0x002, 3 * PP, 0);
            }
        }, std::make_integer_sequence<int, NB>{});
        __builtin_amdgcn_sched_barrier(0);
        if (mt + 1 < MTT) fsc = fsc_next;
    }

    __device__ __forceinline__ void run(const MxSeg &sg, int vt0, int vt1, int pt, int n_paths, const unsigned char *Tm, int ld_tm,
                                        float2 *__restrict__ part_row, unsigned char *smem_) {
        constexpr float LN2 = 0.6931471805599453f;
        smem = smem_;
        int tid_ = threadIdx.x;
        asm volatile("" : "+v"(tid_));
        const int tid = tid_, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        hf = lane >> 5;
        const int li = lane & 31;
        // ---- 1. row operands.  f16 planes straight into accumulation registers (a row past the end reads its lane's slot of block 0 --
        //         inside the buffer whatever its size; its result is not stored and no other row sees it); FP6 planes and their scales
        //         into architectural registers
        bool row_ok[RS];
        int prow[RS];
#pragma unroll
        for (int S = 0; S < RS; ++S) {
            prow[S] = pt * (128 * RS) + wave * (32 * RS) + S * 32 + li;
            row_ok[S] = prow[S] < n_paths;
            const unsigned char *tb0 = Tm + (row_ok[S] ? mx_tm_block(prow[S], ld_tm) : 0) + mx_tm_granule(sg.tm_off, 0, prow[S]);
            const unsigned char *tb = tb0 + hf * 512;
#pragma unroll
            for (int q = 0; q < NS16; ++q)
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&a"(thi[S][q]) : "v"(tb + (q >> 1) * 4096), "n"(2 * (q & 1) * 512) : "memory");
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const i32x4 a = *reinterpret_cast<const i32x4 *>(tb0 + (j * 8 + 4 + 2 * hf) * 512);
                const i32x2 b = *reinterpret_cast<const i32x2 *>(tb0 + (j * 8 + 5) * 512 + 8 * hf);
                t6[S][j] = i32x8{a[0], a[1], a[2], a[3], b[0], b[1], 0, 0};
            }
            tsc[S] = *reinterpret_cast<const i32x2 *>(tb0 + 7 * 512 + 8 * hf);
        }
        descale = sg.descale;
        // ---- 2. LDS-DMA: wave w fills row groups w, w + 4, ... (8 rows each) of every 32-k block
        const unsigned long long bptr = reinterpret_cast<unsigned long long>(sg.B);
        const unsigned long long bptr_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bptr >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)bptr);
        const int nrec = __builtin_amdgcn_readfirstlane(sg.n_vocab) * ROWB;
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(bptr_u), 0, nrec, 0x00020000);
        __amdgpu_buffer_rsrc_t rs_bias = rs_b;
        if (XB) {
            const unsigned long long p2 = reinterpret_cast<unsigned long long>(sg.bias2);
            const unsigned long long p2u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(p2 >> 32)) << 32) |
                                           (unsigned)__builtin_amdgcn_readfirstlane((int)p2);
            rs_bias = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(p2u), 0, __builtin_amdgcn_readfirstlane(sg.n_vocab) * 4, 0x00020000);
        }
        auto issue_bias = [&](int t, int slot) {
            if (!XB) return;
#pragma unroll
            for (int i = 0; i < (TW + 63) / 64; ++i)
                if (i * 64 + 64 <= TW || lane < TW - i * 64)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_bias, (__attribute__((address_space(3))) void *)(smem + BIAS_OFF + slot * (TW * 4) + i * 256),
                                                             4, (t * TW + i * 64 + lane) * 4, 0, 0, 0);
        };
        const int r8 = lane >> 3, dslot = lane & 7;
        const int drow = 8 * wave + r8;
        const int dvoff = drow * ROWB + ((dslot ^ ((drow >> 1) & 7)) * 16);
        constexpr int NDMA = MTT * NB;
        const int x = (li >> 1) & 7;
        const int fbase = (li >> 3) * (NB * 1024) + (li & 7) * 128;
        g_f0 = fbase + ((0 + hf) ^ x) * 16; g_f1 = fbase + ((2 + hf) ^ x) * 16;
        g_6a = fbase + ((4 + 2 * hf) ^ x) * 16; g_6b = fbase + (5 ^ x) * 16 + 8 * hf;
        g_sc = fbase + (7 ^ x) * 16 + 8 * hf;
        // accumulator 1 starts as a finished block of sixteen -1e30 logits (FR: 2^-1e30 = 0): see mx6_body
#pragma unroll
        for (int S = 0; S < RS; ++S) {
            m[S] = FR ? 0.0f : JLM_NEG_BIG; s[S] = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[1][S][r] = -1.0e30f; acc[0][S][r] = 0.0f; }
        }
        {
            const int voff = dvoff + vt0 * (TW * ROWB);
#pragma unroll
            for (int q = 0; q < NDMA; ++q) {
                const int i = q / NB, j = q % NB;
                unsigned char *dst = smem + ((wave + 4 * i) * NB) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, voff + i * (32 * ROWB),
                                                         j * 128, 0, 0);
            }
            issue_bias(vt0, 0);
            if (XB && tid < TW) *reinterpret_cast<float *>(smem + BIAS_OFF + 2 * (TW * 4) + tid * 4) = 0.0f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int S = 0; S < RS; ++S) {
#pragma unroll
            for (int q = 0; q < NS16; ++q) asm volatile("" : "+a"(thi[S][q]));
        }
        __syncthreads();
        int buf = 0;
        int bs_prev = 2, bs_cur = 0, bs_next = 1;
        auto tile = [&](auto masked_c, int t) {
            constexpr bool MASKED = decltype(masked_c)::value != 0;
            const bool more = t + 1 < vt1;
            const __amdgpu_buffer_rsrc_t rs_next = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(bptr_u), 0, more ? nrec : 0, 0x00020000);
            const int voff_next = dvoff + (t + 1) * (TW * ROWB);
            const int lim = sg.n_vocab - t * TW;
            if (more) issue_bias(t + 1, bs_next);
            {
                const unsigned char *bs0 = smem + buf * BUFB;
                F0[0] = *reinterpret_cast<const i32x4 *>(bs0 + g_f0);
                const i32x4 a = *reinterpret_cast<const i32x4 *>(bs0 + g_6a);
                const i32x2 b = *reinterpret_cast<const i32x2 *>(bs0 + g_6b);
                fsc = *reinterpret_cast<const i32x2 *>(bs0 + g_sc);
                F1[0] = *reinterpret_cast<const i32x4 *>(bs0 + g_f1);
                F6[0] = i32x8{a[0], a[1], a[2], a[3], b[0], b[1], 0, 0};
            }
            mx_for_each_ic([&](auto mc) {
                constexpr int mt = decltype(mc)::value;
                constexpr int W = mt & 1;
                // (the accumulator treated in a tile's first block belongs to the tile before -- whole -- or is the dummy: never masked)
                if constexpr (mt == 0) block<false, mt, W>(buf, lim, BIAS_OFF + bs_prev * (TW * 4) + (MTT - 1) * 128, rs_next, voff_next, wave);
                else block<MASKED, mt, W>(buf, lim, BIAS_OFF + bs_cur * (TW * 4) + (mt - 1) * 128, rs_next, voff_next, wave);
            }, std::make_integer_sequence<int, MTT>{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(MX6W_ABL & 8)) __builtin_amdgcn_s_barrier();
            buf ^= 1;
            { const int o = bs_prev; bs_prev = bs_cur; bs_cur = bs_next; bs_next = o; }
        };
        static_assert((MTT * NB) % 2 == 0, "the fragment ring alternates by 32-k block: an even number per tile");
        const int t_full = min(vt1, sg.n_vocab / TW);
        for (int t = vt0; t < t_full; ++t) tile(IC<0>{}, t);
        for (int t = max(vt0, t_full); t < vt1; ++t) tile(IC<1>{}, t);
        {
            const int lim_last = sg.n_vocab - (vt1 - 1) * TW;
            load_bias(BIAS_OFF + bs_prev * (TW * 4) + (MTT - 1) * 128);
            constexpr int NPL = RS * np1<true>();
#pragma unroll
            for (int pc = 0; pc < NPL; ++pc) fold2<true>(1, MTT - 1, lim_last, pc);
        }
#pragma unroll
        for (int S = 0; S < RS; ++S) {
            const float m2 = __shfl_xor(m[S], 32), s2 = __shfl_xor(s[S], 32);
            const float mm = fmaxf(m[S], m2);
            const float ss = fmaf(s[S], __builtin_amdgcn_exp2f(m[S] - mm), s2 * __builtin_amdgcn_exp2f(m2 - mm));
            if (hf == 0 && row_ok[S]) part_row[prow[S]] = make_float2(mm * LN2, ss);
        }
    }
};

// (out of line, arguments made provably wave-uniform again: as mxw_body in jlm_mixed_w.hip)
template <int NB, int NS16, bool XB, bool FR>
__device__ __noinline__ void mx6w_body(const MxSeg &sg, int vt0, int vt1, int pt, int n_paths, const unsigned char *Tm, int ld_tm, float2 *prow,
                                       unsigned char *smem) {
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto unip = [&](const void *q) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(q);
        return reinterpret_cast<const void *>(((unsigned long long)(unsigned)uni((int)(u >> 32)) << 32) | (unsigned)uni((int)u));
    };
    MxSeg u;
    u.B = static_cast<const unsigned char *>(unip(sg.B));
    u.n_vocab = uni(sg.n_vocab); u.k = uni(sg.k); u.t_off = uni(sg.t_off); u.nb = uni(sg.nb); u.tm_off = uni(sg.tm_off); u.seg = uni(sg.seg);
    u.descale = __int_as_float(uni(__float_as_int(sg.descale)));
    u.cs = 0.0f;
    u.bias2 = static_cast<const float *>(unip(sg.bias2));
    Mx6Wide<NB, NS16, mx_blocks_per_tile(NB), XB, FR> w;
    w.run(u, uni(vt0), uni(vt1), uni(pt), uni(n_paths), static_cast<const unsigned char *>(unip(Tm)), uni(ld_tm),
          static_cast<float2 *>(const_cast<void *>(unip(prow))), static_cast<unsigned char *>(const_cast<void *>(unip(smem))));
}

template <bool XB, bool FR, int... SH>
struct Mx6wDispatch;
template <bool XB, bool FR>
struct Mx6wDispatch<XB, FR> {
    static __device__ __forceinline__ void run(const MxSeg &, int, int, int, int, int, const unsigned char *, int, float2 *, unsigned char *) {}
};
template <bool XB, bool FR, int NB, int NS16, int... REST>
struct Mx6wDispatch<XB, FR, NB, NS16, REST...> {
    static __device__ __forceinline__ void run(const MxSeg &sg, int ns16, int vt0, int vt1, int pt, int n_paths, const unsigned char *Tm, int ld_tm,
                                               float2 *prow, unsigned char *smem) {
        if (sg.nb == NB && ns16 == NS16) mx6w_body<NB, NS16, XB, FR>(sg, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, smem);
        else Mx6wDispatch<XB, FR, REST...>::run(sg, ns16, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, smem);
    }
};

template <bool XB, bool FR, int... SH>
__global__ __launch_bounds__(256, 1) void vocab_lse_mx6w_kernel(MxArgs a, const unsigned char *__restrict__ Tm, int ld_tm, float2 *__restrict__ part,
                                                                int ld_part, int n_rows_max, const int *n_dev, int n_ptiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mx6w_smem[];
    const int n_paths = n_dev ? min(*n_dev, n_rows_max) : n_rows_max;
    const int b = blockIdx.x;
    int p, pt;
    const int nb8 = (a.n_cols & ~7) * n_ptiles;
    if (b < nb8) { const int x = b & 7, jb = b >> 3; p = (jb / n_ptiles) * 8 + x; pt = jb % n_ptiles; }
    else { const int bb = b - nb8; p = (a.n_cols & ~7) + bb / n_ptiles; pt = bb % n_ptiles; }
    if (p >= a.n_cols || pt * 256 >= n_paths) return;
    for (int r = a.col_first[p]; r < a.col_first[p + 1]; ++r) {
        const MxSeg sg = a.seg[a.sub_seg[r]];
        const int vt0 = a.sub_t0[r], vt1 = a.sub_t1[r];
        float2 *prow = part + (size_t)r * ld_part;
        if (r != a.col_first[p]) __syncthreads();
        const int ns16 = XB ? 2 * sg.nb : (sg.k + 2 + 15) >> 4;
        Mx6wDispatch<XB, FR, SH...>::run(sg, ns16, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, mx6w_smem);
    }
}
#define MX6W_KERNEL_DSOFTMAX vocab_lse_mx6w_kernel<false, false, 7, 13, 4, 7, 2, 4>
#define MX6W_KERNEL_DSOFTMAX_FR vocab_lse_mx6w_kernel<false, true, 7, 13, 4, 7, 2, 4>
#define MX6W_KERNEL_TIED vocab_lse_mx6w_kernel<true, false, 8, 16>
#define MX6W_KERNEL_TIED_FR vocab_lse_mx6w_kernel<true, true, 8, 16>

}  // namespace

// the shapes the wide mx6 kernel hosts: the D-softmax* 200 / 100 / 50 model (bias columns) and segments of k = 256 with external biases
bool jlm_mx6w_hosts(const MxArgs &a, bool xbias) {
    for (int i = 0; i < a.n_segs; ++i) {
        const int nb = a.seg[i].nb, ns16 = (a.seg[i].k + 2 + 15) / 16;
        if (xbias) { if (nb != 8) return false; }
        else if (!((nb == 7 && ns16 == 13) || (nb == 4 && ns16 == 7) || (nb == 2 && ns16 == 4))) return false;
    }
    return true;
}

// Returns 0, -3 (LDS grant) or a negative HIP error like its caller (jlm_mx6_launch, jlm_mx6.hip).
int jlm_mx6w_launch(const MxArgs &a, bool xbias, int fixed_ref, const void *Tm, int ld_tm, float2 *part, int ld_part, int n_rows_max, const int *n_dev,
                    int n_ptiles, int lds, hipStream_t st) {
    const int which = (xbias ? 2 : 0) + (fixed_ref ? 1 : 0);
    static JlmLdsGrant grant[4];
    const void *fns[4] = {reinterpret_cast<const void *>(MX6W_KERNEL_DSOFTMAX), reinterpret_cast<const void *>(MX6W_KERNEL_DSOFTMAX_FR),
                          reinterpret_cast<const void *>(MX6W_KERNEL_TIED), reinterpret_cast<const void *>(MX6W_KERNEL_TIED_FR)};
    if (int rc = jlm_grant_lds(grant[which], fns[which], lds)) return rc;
    const dim3 grid(a.n_cols * n_ptiles), block(256);
    const unsigned char *tm = reinterpret_cast<const unsigned char *>(Tm);
    switch (which) {
    case 0: hipLaunchKernelGGL(MX6W_KERNEL_DSOFTMAX, grid, block, lds, st, a, tm, ld_tm, part, ld_part, n_rows_max, n_dev, n_ptiles); break;
    case 1: hipLaunchKernelGGL(MX6W_KERNEL_DSOFTMAX_FR, grid, block, lds, st, a, tm, ld_tm, part, ld_part, n_rows_max, n_dev, n_ptiles); break;
    case 2: hipLaunchKernelGGL(MX6W_KERNEL_TIED, grid, block, lds, st, a, tm, ld_tm, part, ld_part, n_rows_max, n_dev, n_ptiles); break;
    default: hipLaunchKernelGGL(MX6W_KERNEL_TIED_FR, grid, block, lds, st, a, tm, ld_tm, part, ld_part, n_rows_max, n_dev, n_ptiles); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(int)e - 100;
    return 0;
}
