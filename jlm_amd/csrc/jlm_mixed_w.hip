// jlm_mixed_w.hip -- round 5: the mixed-row vocabulary kernel (jlm_mixed.hip) in its WIDE form: four waves per workgroup, one per
// SIMD, each keeping 64 hypothesis rows -- two 32-row sets -- so that EVERY vocabulary fragment read from LDS feeds two matrix
// instructions.  Reference: project + softmax, decoder/model.py:141-193, 15-20.
//
// Why (profiles/r05_b_valu_ablate.txt): with every VALU instruction taken out, the eight-wave kernel still needs 61 of its 70 us --
// its matrix instructions alone would need 31.  A 32 x 32 block instruction consumes a 1-KB fragment per 32 cycles and SIMD, i.e.
// 128 B per cycle and CU: all the LDS delivers.  At 32 rows per wave the LDS, not the matrix pipe and not the VALU, is the
// kernel's bound.  Two row sets halve the LDS bytes per matrix instruction.  They need 2 x (k / 2 + 8) operand registers per lane
// (216 at k = 200), which two waves per SIMD do not have: hence one wave per SIMD with the row operands in its 256 ACCUMULATION
// registers (legal MFMA B operands on gfx950; loaded straight into them) and everything the VALU touches -- four accumulator
// pairs, the fragment ring -- in the architectural ones.  Compiled with -mllvm -amdgpu-mfma-vgpr-form (the accumulators of the
// matrix instructions must stay out of the accumulation registers: jlm_gate_ws.hip has the same constraint).
//
// With one wave per SIMD nothing else covers a wave's waits, so the stream is laid out for one in-order issuer:
//   * a fragment ring of eight (two 32-k blocks): every ds_read_b128 is issued eight matrix instructions (>= 256 cycles) ahead;
//   * two accumulator pairs per row set: block n + 1 multiplies into one while block n's logits are combined, max-ed and
//     exponentiated IN PLACE in the other, a few VALU instructions behind each matrix instruction (as mx_body_a2);
//   * the next tile's LDS-DMA instructions ride one per 32-k block (their count per tile and wave equals the tile's 32-k blocks),
//     the one behind a sub-range's last tile through a zero-sized buffer descriptor: no branch in the stream, no memory traffic.
// Same tiles, LDS image, column cuts and partial (max, sum) slices as the eight-wave kernel: jlm_vocab_lse_mixed launches either.
#include "jlm_common.h"
#include <type_traits>
#include <utility>

#include "jlm_mixed_body.h"
using namespace jlm_mx;

#ifndef MXW_ABL
#define MXW_ABL 0      // measurement builds (wrong numbers): 1 no fold, 2 no exp2 in the fold, 4 no LDS-DMA in the loop, 8 no barrier, 16 combine only, 32 no fragment reads in the loop, 64 row set 1 multiplies with set 0's operands
#endif

#ifdef MXW_TRACE
// -DMXW_TRACE: wave 0 of the first 256 workgroups stamps the shader clock at kernel-body start [0], when the row operands and the
// first tile have landed [1], and at the end of every tile [2 + i] (tools/probes/mixed_w_trace.py)
static __device__ unsigned long long jlm_mxw_trace[256][64];
extern "C" int jlm_prof_read_mxw_trace(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(jlm_mxw_trace), sizeof(jlm_mxw_trace)) == hipSuccess ? 0 : -1;
}
#define MXW_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 256 && (i) < 64) jlm_mxw_trace[blockIdx.x][i] = clock64(); } while (0)
#else
#define MXW_STAMP(i) (void)0
#endif

namespace {

// NB 32-k blocks per row, NS16 f16 steps, MTT 32-word blocks per tile, RS row sets of 32 per wave (2: every fragment feeds two matrix
// instructions; 1: contractions whose operands fill the accumulation registers by themselves -- k = 512: 256 of them), XB: the rows
// carry no bias columns (k a multiple of 32): the tile's biases (base-2 units) come into LDS beside it, three slots, and join the
// logits in the combine (as mx_body's XBIAS form).
template <int NB, int NS16, int MTT, int RS, bool XB, bool FR = false>
struct MxWide {
    static_assert(RS == 1 || RS == 2, "one or two row sets per wave");
    static constexpr int ROWB = NB * 128;
    static constexpr int TW = 32 * MTT;
    static constexpr int BUFB = TW * ROWB;
    static constexpr int BIAS_OFF = 2 * BUFB;             // XB: three slots of TW floats behind the two buffers
    static constexpr int RW = 128 * RS;                   // hypothesis rows per workgroup
    static constexpr int NMF = NS16 + 2 * NB;             // matrix instructions of a block and row set
    static constexpr int NSLOT = 4 * RS * NB;             // issue slots of a block (RS per fragment; NS16 odd: RS of the last 32-k block's are empty)
    // per row set: 16 x combine, 8 x max3, 1, 16 x (scale, exp2, add), 1 -- FR (fixed reference): 16 x (combine, exp2, add), 1: no running
    // maximum, s = sum 2^y against the reference 0 (valid while the row's largest base-2 logit stays inside +-100: jlm_vocab_lse_mixed's
    // `fixed_ref` argument, decided by the load-time calibration; the slices then carry m = 0)
    static constexpr int NP1 = FR ? 17 : 42;
    static constexpr int NPIECE = RS * NP1;
    static constexpr int SKIP = 2;                        // (the first slots carry nothing: the finished pairs' last instructions are still in the pipe)
    static constexpr int PP = (NPIECE + NSLOT - SKIP - 1) / (NSLOT - SKIP);

    // row operands: MFMA B operands, in accumulation registers for the whole sub-range
    f16x8 thi[RS][NS16];
    i32x4 thi8[RS][NB], tlo8[RS][NB];
    float csr[RS], descale;
    float m[RS], s[RS];
    float tmax[RS], nmn[RS], sc_old[RS], add0[RS], add1[RS];
    f32x16 accf[2][RS];                                   // two accumulator pairs per row set: blocks alternate
    i32x16 acci[2][RS];
    f32x4 bq[4];                                          // XB: the finished block's biases of this lane's 16 words
    i32x4 F[8];                                           // fragment ring: 32-k block number n lives in F[4 (n & 1) ..]
    int goff[4];
    int hf;
    unsigned char *smem;

    // one piece of the treatment of the finished pair (pf, pi) of row set S
    __device__ __forceinline__ void fold_piece(const bool MASKED, int S, f32x16 &pf, const i32x16 &pi, int mtp, int lim, int pc) {
#pragma clang fp contract(off)          // every fused multiply-add below is written as one: the row sets must round alike
        if (FR) {
            if (pc < 16) {
                const int r = pc;
                float y = fmaf((float)pi[r], csr[S], pf[r]);
                y = XB ? fmaf(y, descale, bq[r >> 2][r & 3]) : y * descale;
                if (MASKED && mtp * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf >= lim) y = JLM_NEG_BIG;
                const float e = __builtin_amdgcn_exp2f(y);
                if (r == 0) { add0[S] = e; add1[S] = 0.0f; } else if (r & 1) add1[S] += e; else add0[S] += e;
            } else if (pc == 16) {
                s[S] += add0[S] + add1[S];
            }
            return;
        }
        if (pc < 16) {
            const int r = pc;
            float y = fmaf((float)pi[r], csr[S], pf[r]);
            if (XB) y = fmaf(y, descale, bq[r >> 2][r & 3]);            // (base-2 logit units from here on)
            pf[r] = (MASKED && mtp * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf >= lim) ? JLM_NEG_BIG : y;
        } else if (pc < 24) {
            const int q = pc - 16;
            const float t2 = fmaxf(pf[2 * q], pf[2 * q + 1]);
            tmax[S] = q == 0 ? t2 : fmaxf(tmax[S], t2);
        } else if (pc == 24) {
            const float mn = fmaxf(m[S], XB ? tmax[S] : tmax[S] * descale);
            nmn[S] = -mn;
            sc_old[S] = __builtin_amdgcn_exp2f(m[S] - mn);
            m[S] = mn;
            add0[S] = 0.0f; add1[S] = 0.0f;
        } else if (pc < 41) {
            const int r = pc - 25;
            const float x = XB ? pf[r] + nmn[S] : fmaf(pf[r], descale, nmn[S]);
            const float e = (MXW_ABL & 2) ? x : __builtin_amdgcn_exp2f(x);
            if (r & 1) add1[S] += e; else add0[S] += e;
        } else if (pc == 41) {
            // (an explicit fma: left to the compiler the two row sets got different contractions -- set 1 a fused multiply-add, set 0 not --
            //  and identical rows 1-ulp different sums depending on the set they sat in: tools/probes/wide_identical_rows.py)
            s[S] = fmaf(s[S], sc_old[S], add0[S] + add1[S]);
        }
    }
    // pieces alternate between the row sets: piece RS q + S -> set S piece q
    // (masked / pw are literals at every call site and this is force-inlined: plain arguments, not template parameters -- hipcc's HOST
    //  pass rejects the template forms inside the nested generic lambdas of block() with "substitution failure")
    __device__ __forceinline__ void fold2(const bool masked, const int pw, int mtp, int lim, int pc2) {
        f32x16 (&pf)[RS] = accf[pw];
        i32x16 (&pi)[RS] = acci[pw];
        if (MXW_ABL & 1) {           // (the finished accumulators stay "used": without this the compiler drops every matrix instruction)
            if (pc2 == 0) { asm volatile("" :: "v"(pf[0]), "v"(pi[0])); if (RS == 2) asm volatile("" :: "v"(pf[RS - 1]), "v"(pi[RS - 1])); }
            return;
        }
        if ((MXW_ABL & 16) && (pc2 / RS) >= 16) {      // combine only
            if (pc2 == 16 * RS) { asm volatile("" :: "v"(pf[0])); if (RS == 2) asm volatile("" :: "v"(pf[RS - 1])); }
            return;
        }
        fold_piece(masked, pc2 % RS, pf[pc2 % RS], pi[pc2 % RS], mtp, lim, pc2 / RS);
    }
    // XB: the finished block's biases (its 32 words start at LDS byte `boff`) for this lane's 16 logits
    __device__ __forceinline__ void load_bias(int boff) {
        if (!XB) return;
        const unsigned char *bp = smem + boff + (4 * hf) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4 *>(bp + q * 32);
    }

    // one 32-word block: its matrix instructions into pair W; pair 1 - W (finished) treated between them; tile t + 1's LDS-DMA
    // instruction (mt NB + J) behind the first matrix instruction of 32-k block J
    template <bool MASKED, int mt, int W>
    __device__ __forceinline__ void block(int buf, int lim_p, int boff_p, const __amdgpu_buffer_rsrc_t rs_next, int voff_next, int wave) {
        constexpr int mtp = (mt + MTT - 1) % MTT;
        f32x16 (&wf)[RS] = accf[W];
        i32x16 (&wi)[RS] = acci[W];
        const unsigned char *bs = smem + buf * BUFB + mt * (4 * NB * 1024);
        load_bias(boff_p);
        __builtin_amdgcn_sched_barrier(0);
        const f32x16 zf = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const i32x16 zi = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        mx_for_each_ic([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            constexpr bool second = 2 * J + 1 < NS16;
            constexpr int R = 4 * ((mt * NB + J) & 1), Rn = 4 * ((mt * NB + J + 1) & 1);        // this 32-k block's ring half, the next one's
            // the fragment that will be needed one 32-k block from now (the next 32-word block's first, behind this block's last)
            auto rd = [&](int g4) {
                if (MXW_ABL & 32) return;
                if (J + 1 < NB) F[Rn + g4] = *reinterpret_cast<const i32x4 *>(bs + (J + 1) * 1024 + goff[g4]);
                else if (mt + 1 < MTT) F[Rn + g4] = *reinterpret_cast<const i32x4 *>(bs + (4 * NB * 1024) + goff[g4]);
            };
            auto pieces = [&](int q) {
                if (q < SKIP) return;
#pragma unroll
                for (int pc = (q - SKIP) * PP; pc < (q - SKIP + 1) * PP && pc < NPIECE; ++pc) fold2(MASKED, 1 - W, mtp, lim_p, pc);
            };
            // DMA piece (mt NB + J) of the next tile: row group wave + 4 i, 32-k block j
            if (!(MXW_ABL & 4)) {
                constexpr int q = mt * NB + J, i = q / NB, j = q % NB;
                unsigned char *dst = smem + (buf ^ 1) * BUFB + ((wave + 4 * i) * NB) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_next, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16,
                                                         voff_next + i * (32 * ROWB), j * 128, 0, 0);
            }
            constexpr int Q0 = 4 * RS * J;
            // fragment order: f16 step 2 J (F[R + 0]), int8 hi8 x the rows' lo8 (F[R + 2]), f16 step 2 J + 1 (F[R + 1]), int8 lo8 x the rows' hi8 (F[R + 3])
            mx_for_each_ic([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                wf[S] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, F[R + 0]), thi[S][2 * J], J == 0 ? zf : wf[S], 0, 0, 0);
                if (S == RS - 1) rd(0);
                pieces(Q0 + S);
            }, std::make_integer_sequence<int, RS>{});
            mx_for_each_ic([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                wi[S] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[R + 2], tlo8[S][J], J == 0 ? zi : wi[S], 0, 0, 0);
                if (S == RS - 1) rd(2);
                pieces(Q0 + RS + S);
            }, std::make_integer_sequence<int, RS>{});
            mx_for_each_ic([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                if constexpr (second) wf[S] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, F[R + 1]), thi[S][second ? 2 * J + 1 : 0], wf[S], 0, 0, 0);
                if (S == RS - 1) rd(1);
                pieces(Q0 + 2 * RS + S);
            }, std::make_integer_sequence<int, RS>{});
            mx_for_each_ic([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                wi[S] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[R + 3], thi8[S][J], wi[S], 0, 0, 0);
                if (S == RS - 1) rd(3);
                pieces(Q0 + 3 * RS + S);
            }, std::make_integer_sequence<int, RS>{});
            // issue order: matrix instruction, [the LDS-DMA instruction], [a fragment read], its share of the fold
            constexpr bool reads = (J + 1 < NB) || (mt + 1 < MTT);
#pragma unroll
            for (int i = 0; i < 4 * RS; ++i) {
                const bool has_m = second || (i / RS != 2);
                if (has_m) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i == 0 && !(MXW_ABL & 4)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if ((i % RS) == RS - 1 && reads && !(MXW_ABL & 32)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (Q0 + i >= SKIP && !(MXW_ABL & 1)) __builtin_amdgcn_sched_group_barrier(0x002, (FR ? 6 : 3) * PP, 0);
            }
        }, std::make_integer_sequence<int, NB>{});
        __builtin_amdgcn_sched_barrier(0);
    }

    __device__ __forceinline__ void run(const MxSeg &sg, int vt0, int vt1, int pt, int n_paths, const unsigned char *Tm, int ld_tm,
                                        float2 *__restrict__ part_row, unsigned char *smem_) {
        constexpr float LN2 = 0.6931471805599453f;
        smem = smem_;
        MXW_STAMP(0);
        int tid_ = threadIdx.x;
        asm volatile("" : "+v"(tid_));
        const int tid = tid_, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        hf = lane >> 5;
        const int li = lane & 31;
        // ---- 1. row operands of every set, straight into accumulation registers.  A row past the end reads its lane's slot of
        //         block 0 (inside the buffer whatever its size; its result is not stored and no other row sees it): selecting zeros
        //         would cost a VALU move per register.
        bool row_ok[RS];
        int prow[RS];
#pragma unroll
        for (int S = 0; S < RS; ++S) {
            prow[S] = pt * RW + wave * (32 * RS) + S * 32 + li;
            row_ok[S] = prow[S] < n_paths;
            // (granule-major packed rows, jlm_mixed_body.h: one contiguous kilobyte per load instruction; the instruction offset has
            //  12 bits, hence a base per 32-k block)
            const unsigned char *tblk = Tm + (row_ok[S] ? mx_tm_block(prow[S], ld_tm) : 0);
            const unsigned char *tb = tblk + mx_tm_granule(sg.tm_off, hf, prow[S]);
#pragma unroll
            for (int q = 0; q < NS16; ++q)
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&a"(thi[S][q]) : "v"(tb + (q >> 1) * 4096), "n"(2 * (q & 1) * 512) : "memory");
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&a"(thi8[S][j]) : "v"(tb + j * 4096), "n"(4 * 512) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&a"(tlo8[S][j]) : "v"(tb + j * 4096), "n"(6 * 512) : "memory");
            }
            const float s_t = *reinterpret_cast<const float *>(tblk + mx_tm_scale(ld_tm, prow[S], sg.seg));
            csr[S] = s_t * sg.cs;
        }
        descale = sg.descale;
        // ---- 2. LDS-DMA: wave w fills row groups w, w + 4, ... (8 rows each) of every 32-k block; lane = (row lane >> 3, slot lane & 7),
        //         source granule = slot ^ ((row >> 1) & 7)
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
        // XB: tile t's biases into slot `slot` (every wave writes the same words: no branch on the wave number; words past the
        // segment's end read 0 and are masked anyway)
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
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) goff[g4] = fbase + ((2 * g4 + hf) ^ x) * 16;
        // pair 1 starts as a finished block of sixteen -1e30 logits: its fold leaves (m, s) = (very negative, 16), which the first
        // real fold scales to 0 (as mx_body's v[])
#pragma unroll
        for (int S = 0; S < RS; ++S) {
            m[S] = FR ? 0.0f : JLM_NEG_BIG; s[S] = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { accf[1][S][r] = -1.0e30f; acci[1][S][r] = 0; accf[0][S][r] = 0.0f; acci[0][S][r] = 0; }
        }
        // the first tile (and its biases: slot 0)
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
            // (XB: the dummy pair's "biases" -- slot 2 is read by the first block's combine before anything was written there)
            if (XB && tid < TW) *reinterpret_cast<float *>(smem + BIAS_OFF + 2 * (TW * 4) + tid * 4) = 0.0f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int S = 0; S < RS; ++S) {
#pragma unroll
            for (int q = 0; q < NS16; ++q) asm volatile("" : "+a"(thi[S][q]));
#pragma unroll
            for (int j = 0; j < NB; ++j) { asm volatile("" : "+a"(thi8[S][j])); asm volatile("" : "+a"(tlo8[S][j])); }
        }
        __syncthreads();
        MXW_STAMP(1);
        int buf = 0;
        int bs_prev = 2, bs_cur = 0, bs_next = 1;           // bias slots of the tile before, this tile, the next one
        // P: the accumulator pair the tile's first block writes (an even number of blocks per tile: always 0)
        auto tile = [&](auto masked_c, auto p_c, int t) {
            constexpr bool MASKED = decltype(masked_c)::value != 0;
            constexpr int P = decltype(p_c)::value;
            const bool more = t + 1 < vt1;
            // (behind a sub-range's last tile: a descriptor of zero records -- the instructions stay in the stream, nothing is fetched)
            const __amdgpu_buffer_rsrc_t rs_next = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(bptr_u), 0, more ? nrec : 0, 0x00020000);
            const int voff_next = dvoff + (t + 1) * (TW * ROWB);
            const int lim = sg.n_vocab - t * TW;
            if (more) issue_bias(t + 1, bs_next);
            {
                const unsigned char *bs0 = smem + buf * BUFB;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) F[g4] = *reinterpret_cast<const i32x4 *>(bs0 + goff[g4]);
            }
            mx_for_each_ic([&](auto mc) {
                constexpr int mt = decltype(mc)::value;
                constexpr int W = (P + mt) & 1;
                // (the pair treated in a tile's first block belongs to the tile before -- whole -- or is the dummy: never masked)
                if constexpr (mt == 0) block<false, mt, W>(buf, lim, BIAS_OFF + bs_prev * (TW * 4) + (MTT - 1) * 128, rs_next, voff_next, wave);
                else block<MASKED, mt, W>(buf, lim, BIAS_OFF + bs_cur * (TW * 4) + (mt - 1) * 128, rs_next, voff_next, wave);
            }, std::make_integer_sequence<int, MTT>{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(MXW_ABL & 8)) __builtin_amdgcn_s_barrier();
            buf ^= 1;
            { const int o = bs_prev; bs_prev = bs_cur; bs_cur = bs_next; bs_next = o; }
            MXW_STAMP(2 + t - vt0);
        };
        const int t_full = min(vt1, sg.n_vocab / TW);
        int last_pair;                                       // the pair the last block wrote
        if constexpr (MTT % 2 == 0) {
            for (int t = vt0; t < t_full; ++t) tile(IC<0>{}, IC<0>{}, t);
            for (int t = max(vt0, t_full); t < vt1; ++t) tile(IC<1>{}, IC<0>{}, t);
            last_pair = 1;
        } else {
            // an odd number of blocks per tile: the tiles alternate.  Whole tiles two at a time, then what is left (at most one whole
            // tile and the segment's partial one)
            int t = vt0, par = 0;
            for (; t + 1 < t_full; t += 2) { tile(IC<0>{}, IC<0>{}, t); tile(IC<0>{}, IC<1>{}, t + 1); }
            for (; t < vt1; ++t) {
                const bool whole = t < t_full;
                if (par == 0) { if (whole) tile(IC<0>{}, IC<0>{}, t); else tile(IC<1>{}, IC<0>{}, t); }
                else { if (whole) tile(IC<0>{}, IC<1>{}, t); else tile(IC<1>{}, IC<1>{}, t); }
                par ^= 1;
            }
            last_pair = (par + MTT) & 1;                    // (par = pair of the NEXT tile's first block; MTT odd: the last tile started on par ^ 1)
        }
        {
            const int lim_last = sg.n_vocab - (vt1 - 1) * TW;
            load_bias(BIAS_OFF + bs_prev * (TW * 4) + (MTT - 1) * 128);
            if (last_pair) {
#pragma unroll
                for (int pc = 0; pc < NPIECE; ++pc) fold2(true, 1, MTT - 1, lim_last, pc);
            } else {
#pragma unroll
                for (int pc = 0; pc < NPIECE; ++pc) fold2(true, 0, MTT - 1, lim_last, pc);
            }
        }
#pragma unroll
        for (int S = 0; S < RS; ++S) {
            const float m2 = __shfl_xor(m[S], 32), s2 = __shfl_xor(s[S], 32);
            const float mm = fmaxf(m[S], m2);
            const float ss = fmaf(s[S], __builtin_amdgcn_exp2f(m[S] - mm), s2 * __builtin_amdgcn_exp2f(m2 - mm));
            if (hf == 0 && row_ok[S]) part_row[prow[S]] = make_float2(mm * LN2, ss);
        }
        MXW_STAMP(63);
    }
};

// (out of line: hosted inline, the three bodies of the D-softmax* kernel cost each other 64 spilled registers, some inside the tile loops)
template <int NB, int NS16, int RS, bool XB, bool FR>
__device__ __noinline__ void mxw_body(const MxSeg &sg, int vt0, int vt1, int pt, int n_paths, const unsigned char *Tm, int ld_tm, float2 *prow,
                                      unsigned char *smem) {
    // (arguments of a real call arrive in vector registers: make the wave-uniform ones provably uniform again -- loop counters,
    //  descriptor words and LDS addresses must be scalar, or every LDS-DMA instruction sits in a waterfall loop)
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto unip = [&](const void *q) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(q);
        return reinterpret_cast<const void *>(((unsigned long long)(unsigned)uni((int)(u >> 32)) << 32) | (unsigned)uni((int)u));
    };
    MxSeg u;
    u.B = static_cast<const unsigned char *>(unip(sg.B));
    u.n_vocab = uni(sg.n_vocab); u.k = uni(sg.k); u.t_off = uni(sg.t_off); u.nb = uni(sg.nb); u.tm_off = uni(sg.tm_off); u.seg = uni(sg.seg);
    u.descale = __int_as_float(uni(__float_as_int(sg.descale)));
    u.cs = __int_as_float(uni(__float_as_int(sg.cs)));
    u.bias2 = static_cast<const float *>(unip(sg.bias2));
    MxWide<NB, NS16, mx_blocks_per_tile(NB), RS, XB, FR> w;
    w.run(u, uni(vt0), uni(vt1), uni(pt), uni(n_paths), static_cast<const unsigned char *>(unip(Tm)), uni(ld_tm),
          static_cast<float2 *>(const_cast<void *>(unip(prow))), static_cast<unsigned char *>(const_cast<void *>(unip(smem))));
}

template <int RS, bool XB, bool FR, int... SH>      // SH = NB0, NS0, NB1, NS1, ...
struct MxwDispatch;
template <int RS, bool XB, bool FR>
struct MxwDispatch<RS, XB, FR> {
    static __device__ __forceinline__ void run(const MxSeg &, int, int, int, int, int, const unsigned char *, int, float2 *, unsigned char *) {}
};
template <int RS, bool XB, bool FR, int NB, int NS16, int... REST>
struct MxwDispatch<RS, XB, FR, NB, NS16, REST...> {
    static __device__ __forceinline__ void run(const MxSeg &sg, int ns16, int vt0, int vt1, int pt, int n_paths, const unsigned char *Tm, int ld_tm,
                                               float2 *prow, unsigned char *smem) {
        if (sg.nb == NB && ns16 == NS16) {
            mxw_body<NB, NS16, RS, XB, FR>(sg, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, smem);
        } else {
            MxwDispatch<RS, XB, FR, REST...>::run(sg, ns16, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, smem);
        }
    }
};

#ifdef JLM_WGTIME
// -DJLM_WGTIME: per workgroup [start, end] on the constant 100 MHz clock, the segment of its last sub-range, shader-clock cycles in
// between (tools/probes/mixed_clock.py with JLM_MX_WIDE=1)
static __device__ unsigned long long jlm_prof_wg_mxw[1024][4];
#define MXW_WG_T0 const unsigned long long wg_t0 = wall_clock64(), wg_c0 = clock64(); int si_last = 0;
#define MXW_WG_T1 if (threadIdx.x == 0 && b < 1024) { jlm_prof_wg_mxw[b][0] = wg_t0; jlm_prof_wg_mxw[b][1] = wall_clock64(); jlm_prof_wg_mxw[b][2] = si_last; jlm_prof_wg_mxw[b][3] = clock64() - wg_c0; }
#else
#define MXW_WG_T0
#define MXW_WG_T1
#endif

template <int RS, bool XB, bool FR, int... SH>
__global__ __launch_bounds__(256, 1) void vocab_lse_mixedw_kernel(MxArgs a, const unsigned char *__restrict__ Tm, int ld_tm, float2 *__restrict__ part,
                                                                  int ld_part, int n_rows_max, const int *n_dev, int n_ptiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mxw_smem[];
    constexpr int RW = 128 * RS;
    const int n_paths = n_dev ? min(*n_dev, n_rows_max) : n_rows_max;
    const int b = blockIdx.x;
    int p, pt;
    const int nb8 = (a.n_cols & ~7) * n_ptiles;
    if (b < nb8) { const int x = b & 7, jb = b >> 3; p = (jb / n_ptiles) * 8 + x; pt = jb % n_ptiles; }
    else { const int bb = b - nb8; p = (a.n_cols & ~7) + bb / n_ptiles; pt = bb % n_ptiles; }
    if (p >= a.n_cols || pt * RW >= n_paths) return;
    MXW_WG_T0
    for (int r = a.col_first[p]; r < a.col_first[p + 1]; ++r) {
        const MxSeg sg = a.seg[a.sub_seg[r]];
        const int vt0 = a.sub_t0[r], vt1 = a.sub_t1[r];
        float2 *prow = part + (size_t)r * ld_part;
        if (r != a.col_first[p]) __syncthreads();
        const int ns16 = XB ? 2 * sg.nb : (sg.k + 2 + 15) >> 4;
#ifdef JLM_WGTIME
        si_last = a.sub_seg[r];
#endif
        MxwDispatch<RS, XB, FR, SH...>::run(sg, ns16, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, mxw_smem);
    }
    MXW_WG_T1
}
// the shapes of BASELINE configs[1] (D-softmax* 200 / 100 / 50), two row sets per wave
#define MXW_KERNEL_DSOFTMAX vocab_lse_mixedw_kernel<2, false, false, 7, 13, 4, 7, 2, 4>
// a contraction of 512 (untied models at H = 512: the vocabulary matrix itself): sixteen 32-k blocks = 256 accumulation registers of
// row operands, one row set per wave, external biases
#define MXW_KERNEL_K512 vocab_lse_mixedw_kernel<1, true, false, 16, 32>
// the tied k = 256 models (BASELINE configs[2..4]) with two row sets per wave: 2 x 128 = all 256 accumulation registers (JLM_MX_WIDE=1: A/B)
#define MXW_KERNEL_TIED vocab_lse_mixedw_kernel<2, true, false, 8, 16>
// ... the same with the fixed reference (FR): no running maximum
#define MXW_KERNEL_TIED_FR vocab_lse_mixedw_kernel<2, true, true, 8, 16>
#define MXW_KERNEL_K512_FR vocab_lse_mixedw_kernel<1, true, true, 16, 32>


#ifdef JLM_MX_RESOURCES
#define MXW_RES(NB_, NS_, RS_, XB_) __global__ __launch_bounds__(256, 1) void mxw_res_##NB_##_##NS_(MxSeg sg, const unsigned char *Tm, int ld_tm, float2 *part) { \
        extern __shared__ __attribute__((aligned(16))) unsigned char sm_[]; MxWide<NB_, NS_, mx_blocks_per_tile(NB_), RS_, XB_, false> w; \
        w.run(sg, 0, 100, blockIdx.x, 2560, Tm, ld_tm, part, sm_); }
MXW_RES(2, 4, 2, false) MXW_RES(4, 7, 2, false) MXW_RES(7, 13, 2, false) MXW_RES(16, 32, 1, true)
#endif

}  // namespace

#ifdef JLM_WGTIME
extern "C" int jlm_prof_read_wg_mxw(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(jlm_prof_wg_mxw), sizeof(jlm_prof_wg_mxw)) == hipSuccess ? 0 : -1;
}
#endif

// Launch of the wide kernel for the shapes it hosts; called by jlm_vocab_lse_mixed (jlm_mixed.hip) with its column cuts.
//   which 0: the D-softmax* 200 / 100 / 50 model (every segment in the bias-column form with (nb, f16 steps) in {(7, 13), (4, 7), (2, 4)}),
//            256 rows per workgroup;
//   which 1: ONE segment of k = 512 in the external-bias form (nb = 16), 128 rows per workgroup;
//   which 2: segments of k = 256 in the external-bias form (nb = 8), 256 rows per workgroup;
//   which 3 / 4: forms 1 / 2 with the fixed reference (no running maximum).
// Returns 0, or -3 (LDS grant) / a negative HIP error like its caller.
int jlm_mx_wide_launch(int which, const MxArgs &a, const void *Tm, int ld_tm, float2 *part, int ld_part, int n_rows_max, const int *n_dev, int n_ptiles,
                       int lds, hipStream_t st) {
    static JlmLdsGrant grant[5];
    const void *fns[5] = {reinterpret_cast<const void *>(MXW_KERNEL_DSOFTMAX), reinterpret_cast<const void *>(MXW_KERNEL_K512),
                          reinterpret_cast<const void *>(MXW_KERNEL_TIED), reinterpret_cast<const void *>(MXW_KERNEL_K512_FR),
                          reinterpret_cast<const void *>(MXW_KERNEL_TIED_FR)};
    if (which < 0 || which > 4) return -1;
    if (int rc = jlm_grant_lds(grant[which], fns[which], lds)) return rc;
    const dim3 grid(a.n_cols * n_ptiles), block(256);
    const unsigned char *tm = reinterpret_cast<const unsigned char *>(Tm);
    if (which == 0) hipLaunchKernelGGL(MXW_KERNEL_DSOFTMAX, grid, block, lds, st, a, tm, ld_tm, part, ld_part, n_rows_max, n_dev, n_ptiles);
    else if (which == 1) hipLaunchKernelGGL(MXW_KERNEL_K512, grid, block, lds, st, a, tm, ld_tm, part, ld_part, n_rows_max, n_dev, n_ptiles);
    else if (which == 2) hipLaunchKernelGGL(MXW_KERNEL_TIED, grid, block, lds, st, a, tm, ld_tm, part, ld_part, n_rows_max, n_dev, n_ptiles);
    else if (which == 3) hipLaunchKernelGGL(MXW_KERNEL_K512_FR, grid, block, lds, st, a, tm, ld_tm, part, ld_part, n_rows_max, n_dev, n_ptiles);
    else hipLaunchKernelGGL(MXW_KERNEL_TIED_FR, grid, block, lds, st, a, tm, ld_tm, part, ld_part, n_rows_max, n_dev, n_ptiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(int)e - 100;
    return 0;
}
