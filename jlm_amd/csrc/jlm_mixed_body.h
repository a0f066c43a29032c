// jlm_mixed_body.h -- device side of the mixed (f16 hi.hi + int8 cross terms) vocabulary kernel: segment descriptor and the
// per-sub-range body, shared by jlm_mixed.hip (all segments mixed) and jlm_split.hip (the hybrid kernel: mixed bodies for the long
// contractions, split-f16 bodies for the short ones).  See jlm_mixed.hip for the scheme.
#pragma once
#include "jlm_common.h"
#include <type_traits>
#include <utility>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

#define MX_MAX_NB 8                // 32-k blocks per row: k + 2 <= 256 (the nine-block form does not fit the register file)
#define MXW_MAX_NB 16              // ... of the wide kernel's one-row-set form (jlm_mixed_w.hip): k = 512, the row operands in 256 accumulation registers
#define MX_MAX_PARTS 96
#define MX_MAX_SUB (MX_MAX_PARTS + JLM_MAX_SEGMENTS)

#ifndef MX_DMA_SPREAD
#define MX_DMA_SPREAD 1
#endif
#ifndef MX_ABL
#define MX_ABL 0      // measurement builds: 1 no in-stream fold, 2 no combine, 4 no DMA in the loop, 8 no barrier, 32 no LDS reads in the loop
#endif

namespace jlm_mx {

// 32-word blocks per tile by contraction length (host and device): ~54-64 matrix instructions per tile and wave
#ifndef MX_MTT_MID
#define MX_MTT_MID 4
#endif
__host__ __device__ constexpr int mx_blocks_per_tile(int nb) { return nb >= 9 ? 1 : nb >= 5 ? 2 : nb >= 3 ? MX_MTT_MID : 8; }

template <int N> using IC = std::integral_constant<int, N>;
template <class F, int... I>
__device__ __forceinline__ void mx_for_each_ic(F &&f, std::integer_sequence<int, I...>) { (f(IC<I>{}), ...); }

// ------------------------------------------------------------------------------------------------ packed hypothesis rows (Tm)
// jlm_pack_t_mixed writes, the vocabulary kernels read.  Round 5: GRANULE-MAJOR inside blocks of 32 rows -- a row's 16-byte granule
// g (8 per 128-byte mixed block, all segments concatenated) lives at
//      block (r / 32) x 32 ld_tm 4   +   g x 512   +   (r % 32) x 16
// so that the 64 lanes of a wave (32 rows x the two granules hf = 0 / 1 of an operand) read ONE contiguous kilobyte per load
// instruction.  Row-major rows (rounds 3-4) made every operand load 64 separate cache lines: 54 loads x 64 lines x 4 waves = 14 k
// cycles of the texture path per workgroup at k = 200, a quarter of a single-segment launch (profiles/r05_e_clock.txt).  The rows'
// int8 scales (JLM_MAX_SEGMENTS floats per row) follow the granules of the block: 32 ld_tm 4 - 1024 + (r % 32) x 32.
// The buffer holds ceil(rows / 32) x 32 rows of ld_tm floats.
__device__ __forceinline__ size_t mx_tm_block(int r, int ld_tm) { return (size_t)(r >> 5) * 32 * ld_tm * 4; }
__device__ __forceinline__ int mx_tm_granule(int tm_off, int g, int r) { return (tm_off / 16 + g) * 512 + (r & 31) * 16; }
__device__ __forceinline__ int mx_tm_scale(int ld_tm, int r, int seg) { return 32 * (ld_tm * 4 - 4 * JLM_MAX_SEGMENTS) + (r & 31) * (4 * JLM_MAX_SEGMENTS) + 4 * seg; }

// ------------------------------------------------------------------------------------------------ the kernel
struct MxSeg {
    const unsigned char *B;      // mixed rows
    int n_vocab, k, t_off, nb;   // words, true contraction length, column offset in T, 32-k blocks per row (k + 2 <= 32 nb)
    int tm_off, seg;             // byte offset of the segment's blocks inside a packed T row; segment number (row scale index)
    float descale;               // 2^-(eT + eB): f32 accumulator -> base-2 logit
    float cs;                    // s_b 2^-11: (row scale s_t x) int accumulator -> the f32 accumulator's units
    const float *bias2;          // rows WITHOUT bias columns (32 nb < k + 2: k = 256): the words' biases x log2 e, else unused
};

struct MxArgs {
    int n_cols, n_sub, n_segs;
    MxSeg seg[JLM_MAX_SEGMENTS];
    unsigned char col_first[MX_MAX_PARTS + 1];
    unsigned char sub_seg[MX_MAX_SUB];
    unsigned short sub_t0[MX_MAX_SUB], sub_t1[MX_MAX_SUB];
};

// one sub-range (tiles [vt0, vt1) of one segment) for this workgroup's 256 rows; NB 32-k blocks, NS16 f16 steps (2 NB or 2 NB - 1)
// XBIAS: the rows carry no bias columns (a contraction that fills its last block: the tied k = 256 models); the tile's biases
// (base-2 units) come into LDS beside it -- one 4-byte DMA per word, three slots: the block awaiting its fold belongs to the tile
// before the one being multiplied while the tile after it is already landing -- and join the logits in the combine:
// v = y descale + bias, one VALU instruction per logit more than the column form.
template <int NB, int NS16, int MTT, bool XBIAS = false>
__device__ __forceinline__ void mx_body(const MxSeg &sg, int vt0, int vt1, int pt, int n_paths, const float *__restrict__ T, int ldt,
                                        const int *__restrict__ rows, float2 *__restrict__ part_row, unsigned char *smem) {
    // (T = the PACKED hypothesis rows of jlm_pack_t_mixed, ldt = their stride in 4-byte units; the rows' int8 scales follow the
    //  rows' blocks: see pack_t_mixed_kernel)
    constexpr float LN2 = 0.6931471805599453f;
    constexpr int ROWB = NB * 128;                       // bytes per mixed row
    constexpr int TW = 32 * MTT;                         // words per tile: MTT 32-word blocks (2 at k = 200 ... 8 at k = 50: about the
                                                         // same MFMA count -- and DMA lead time -- per tile whatever the contraction length)
    constexpr int BUFB = TW * ROWB;                      // bytes per LDS buffer
    constexpr int BIAS_OFF = 2 * BUFB;                   // XBIAS: three slots of TW floats behind the two buffers
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = lane >> 5, li = lane & 31;
    // ---- 1. this lane's row operands, ready-made by jlm_pack_t_mixed (one round trip, no arithmetic: quantising the rows here
    //         cost 21 us of a 52-us launch at k = 200 -- every one of a row tile's 24 column workgroups repeated it)
    const int prow = pt * 256 + wave * 32 + li;
    const bool row_ok = prow < n_paths;
    // (granule-major packed rows: every load below is one contiguous kilobyte per wave; `rows` is not used -- the packed rows are compact)
    // (a row past the end reads block 0's slot of its lane -- inside the buffer whatever its size -- and is zeroed below)
    const unsigned char *tblk = reinterpret_cast<const unsigned char *>(T) + (row_ok ? mx_tm_block(prow, ldt) : 0);
    f16x8 thi[NS16];
    i32x4 thi8[NB], tlo8[NB];
    float s_t;
    {
        const unsigned char *tb = tblk + mx_tm_granule(sg.tm_off, hf, prow);
        i32x4 raw[NS16 + 2 * NB];
#pragma unroll
        for (int q = 0; q < NS16; ++q) raw[q] = *reinterpret_cast<const i32x4 *>(tb + ((q >> 1) * 8 + 2 * (q & 1)) * 512);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            raw[NS16 + 2 * j] = *reinterpret_cast<const i32x4 *>(tb + (j * 8 + 4) * 512);
            raw[NS16 + 2 * j + 1] = *reinterpret_cast<const i32x4 *>(tb + (j * 8 + 6) * 512);
        }
        s_t = *reinterpret_cast<const float *>(tblk + mx_tm_scale(ldt, prow, sg.seg));
        const i32x4 z = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < NS16; ++q) thi[q] = __builtin_bit_cast(f16x8, row_ok ? raw[q] : z);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            thi8[j] = row_ok ? raw[NS16 + 2 * j] : z;
            tlo8[j] = row_ok ? raw[NS16 + 2 * j + 1] : z;
        }
    }
    const float csr = s_t * sg.cs;
    const float descale = sg.descale;
    // (tried: starting the int accumulator at 0x4B400000 so that its bits read as the float 12582912 + sum -- no v_cvt_f32_i32 per
    //  logit, the constant leaves again in the fold's offset.  Interleaved A/B of two builds: 29.5 -> 28.5 us at k = 100, 26.4 -> 25.3
    //  at k = 50, and nothing on the three-segment launch (69.0 vs 69.5 us); not kept.)

    // ---- 2. LDS-DMA of a tile: wave w fills row group w (8 rows) of every block j: one 1-KB piece per (w, j), lane = (row lane >> 3,
    //         slot lane & 7), source granule = slot ^ ((row >> 1) & 7) (the swizzle sits on the source, the LDS image is lane-linear)
    // (the descriptor's words made provably wave-uniform: otherwise every DMA instruction sits in a waterfall loop)
    const unsigned long long bptr = reinterpret_cast<unsigned long long>(sg.B);
    const unsigned long long bptr_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bptr >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)bptr);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(bptr_u), 0,
                                                                          __builtin_amdgcn_readfirstlane(sg.n_vocab) * ROWB, 0x00020000);
    const int r8 = lane >> 3, dslot = lane & 7;
    const int drow = 8 * wave + r8;
    const int dvoff = drow * ROWB + ((dslot ^ ((drow >> 1) & 7)) * 16);
    constexpr int NRG = MTT / 2;                         // row groups (8 rows) per wave and tile: wave, wave + 8, ...
    __amdgpu_buffer_rsrc_t rs_bias = rs_b;
    if (XBIAS) {
        const unsigned long long p2 = reinterpret_cast<unsigned long long>(sg.bias2);
        const unsigned long long p2u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(p2 >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)p2);
        rs_bias = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(p2u), 0, __builtin_amdgcn_readfirstlane(sg.n_vocab) * 4, 0x00020000);
    }
    auto issue_bias = [&](int t) {
        if (XBIAS && wave == 0) {                         // (words past the segment's end read 0: they are masked anyway)
#pragma unroll
            for (int i = 0; i < (TW + 63) / 64; ++i)
                if (i * 64 + 64 <= TW || lane < TW - i * 64)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_bias, (__attribute__((address_space(3))) void *)(smem + BIAS_OFF + (t % 3) * (TW * 4) + i * 256),
                                                             4, (t * TW + i * 64 + lane) * 4, 0, 0, 0);
        }
    };
    constexpr int NDMA = NRG * NB;                       // LDS-DMA instructions per wave and tile
    // (instruction q rides behind matrix instruction q % 4 of 32-k block q / 4 of the tile's first 32-word block; the last 32-k block has
    //  three matrix instructions when NS16 is odd: shapes whose first block has no place for every q keep the burst)
    // Only the external-bias bodies (the tied k = 256 models: -1 ... -2 %): the D-softmax* launch measures the same either way and the tile
    // requested behind each sub-range's last one is 1.8 MB of its HBM traffic per launch.
    constexpr bool SPREAD = MX_DMA_SPREAD && XBIAS && NDMA <= 4 * (NB - 1) + (NS16 == 2 * NB ? 4 : 3);
    auto issue_rows = [&](int t, int buf) {
        // the row goes into the per-lane offset (the part the hardware range-checks: a row at or past n_vocab reads zeros), the
        // block into the scalar offset
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
    auto issue_piece = [&](int t, int buf, int q) {       // instruction q = i NB + j of issue_rows
        const int i = q / NB, j = q % NB;
        unsigned char *dst = smem + buf * BUFB + ((wave + 8 * i) * NB) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16,
                                                 dvoff + t * (TW * ROWB) + i * (64 * ROWB), j * 128, 0, 0);
    };
    // fragment addresses: block mt, row li: piece (4 mt + li / 8, j); inside it row li % 8, granule g ^ ((li >> 1) & 7)
    const int x = (li >> 1) & 7;
    const int fbase = (li >> 3) * (NB * 1024) + (li & 7) * 128;
    int goff[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) goff[g4] = fbase + ((2 * g4 + hf) ^ x) * 16;       // granules hf, 2 + hf (f16 steps), 4 + hf (hi8), 6 + hf (lo8)

    float m = JLM_NEG_BIG, s = 0.0f;
    // One accumulator pair.  When a block's last MFMA is out, its 16 logits are COMBINED (y = acc_f + s_t s_b 2^-11 acc_i, masked
    // past the segment's end) into v[16] -- a short VALU burst -- and the accumulators are free for the next block; the fold of
    // v (max; then scale, exp2, add) is issued between the next block's MFMAs.  (Folding as a burst after the block left the
    // matrix pipe idle as long as the block's MFMAs take: 108 us; a second accumulator pair for a fully in-stream fold does
    // not fit beside the 108 registers of row operands at k = 200: 44 of them went to scratch and every reload waited
    // vmcnt(0), i.e. for the tile in flight: 130 us.)  v starts at -1e30: the first fold leaves (m, s) = (very negative, 16),
    // which the first real fold scales to 0.
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = -1.0e30f;
    float tmax, nmn, sc_old, add0, add1;
    constexpr int NMF = NS16 + 2 * NB;                   // matrix instructions of a block
    constexpr int NPIECE = 8 + 1 + 16 + 1;               // 8 x max3, 1, 16 x exp, 1
    constexpr int PP = (NPIECE + NMF - 1) / NMF;
    auto fold_piece = [&](int pc) {
        if (MX_ABL & 1) { if (pc == 0) asm volatile("" :: "v"(v[0]), "v"(v[5]), "v"(v[10]), "v"(v[15])); return; }
        if (MX_ABL & 64) {       // timing model only (wrong numbers): accumulators in base-2 logit units, no running maximum: exp2 + add
            if (pc == 8) { add0 = 0.0f; add1 = 0.0f; }
            else if (pc > 8 && pc < 25) { const float e = __builtin_amdgcn_exp2f(v[pc - 9]); if (pc & 1) add1 += e; else add0 += e; }
            else if (pc == 25) s += add0 + add1;
            return;
        }
        if (pc < 8) {
            const float t2 = fmaxf(v[2 * pc], v[2 * pc + 1]);
            tmax = pc == 0 ? t2 : fmaxf(tmax, t2);
        } else if (pc == 8) {
            const float mn = fmaxf(m, XBIAS ? tmax : tmax * descale);     // (XBIAS: v is in base-2 logit units already)
            nmn = -mn;
            sc_old = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            add0 = 0.0f; add1 = 0.0f;
        } else if (pc < 25) {
            const int r = pc - 9;
            const float e = __builtin_amdgcn_exp2f(XBIAS ? v[r] + nmn : fmaf(v[r], descale, nmn));
            if (r & 1) add1 += e; else add0 += e;
        } else if (pc == 25) {
            s = s * sc_old + (add0 + add1);
        }
    };
    issue(vt0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int buf = 0;
    f32x16 accf;
    i32x16 acci;
    bool have = false;                                    // accf / acci hold a finished block that is not combined yet
    int lim_acc = 0, mt_acc = 0, t_acc = 0;               // ... of tile t_acc with lim_acc valid words, its block mt_acc
    // (the mask -- words past the segment's end, zeros from the range-checked DMA -- costs a compare and a select per logit, a
    //  quarter of the block's VALU work: only the tile loop of a segment's last, partial tile carries it)
    auto combine = [&](auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value != 0;
        if (MX_ABL & 2) { asm volatile("" :: "v"(accf), "v"(acci)); return; }
        f32x4 bq[4];
        if (XBIAS) {
            const unsigned char *bp = smem + BIAS_OFF + (t_acc % 3) * (TW * 4) + (mt_acc * 32 + 4 * hf) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4 *>(bp + q * 32);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y = (MX_ABL & 128) ? accf[r] + __int_as_float(acci[r]) : fmaf((float)acci[r], csr, accf[r]);
            if (XBIAS) y = fmaf(y, descale, bq[r >> 2][r & 3]);
            v[r] = (MASKED && mt_acc * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf >= lim_acc) ? JLM_NEG_BIG : y;
        }
    };
    auto tile = [&](auto masked_c, int t) {
        // The next tile's LDS-DMA instructions (~56 issue cycles each, 7-8 per wave) ride behind the first block's matrix instructions,
        // one each, instead of standing in front of them (MX_DMA_SPREAD; interleaved A/B in tools/gpu_lse_spread.sh).  For that they
        // are straight-line code: the tile behind a sub-range's last one is requested all the same (rows past the segment's end read
        // zeros; the buffer it lands in is not read again and the tile's closing vmcnt(0) covers it).
        if (!SPREAD) { if (!(MX_ABL & 4) && t + 1 < vt1) issue(t + 1, buf ^ 1); }
        else if (!(MX_ABL & 4) && t + 1 < vt1) issue_bias(t + 1);
        const int lim = sg.n_vocab - t * TW;               // valid words of this tile
        // Fragments of 32-k block j: F[0..1] the f16 granules of steps 2 j, 2 j + 1; F[2] hi8, F[3] lo8; one register set, each
        // refilled in place with block j + 1's (or the next 32-word block's first) right behind the MFMA that read it.  f16 and
        // int8 instructions alternate: consecutive ones never share an accumulator.  A tile's first fragments are requested
        // right behind the barrier; the combine of the previous tile's last block runs while they are on their way.
        i32x4 F[4];
        {
            const unsigned char *bs0 = smem + buf * BUFB;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) F[g4] = *reinterpret_cast<const i32x4 *>(bs0 + goff[g4]);
        }
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) {
            const unsigned char *bs = smem + buf * BUFB + mt * (4 * NB * 1024);
            if (have) combine(masked_c);                  // the block before this one: its logits into v (frees the accumulators)
            __builtin_amdgcn_sched_barrier(0);
            const f32x16 zf = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const i32x16 zi = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            auto rd = [&](int g4, int j) {
                if (MX_ABL & 32) return;
                if (j < NB) F[g4] = *reinterpret_cast<const i32x4 *>(bs + j * 1024 + goff[g4]);
                else if (mt + 1 < MTT) F[g4] = *reinterpret_cast<const i32x4 *>(bs + (4 * NB * 1024) + goff[g4]);
            };
            mx_for_each_ic([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                constexpr bool second = 2 * J + 1 < NS16;         // the block's second f16 step exists
                constexpr bool rdm = (J + 1 < NB);                // (reads behind the last block: only when a next 32-word block exists)
                auto pieces = [&](int q) {                        // the fold pieces that ride behind matrix instruction q of the block
#pragma unroll
                    for (int pc = q * PP; pc < (q + 1) * PP && pc < NPIECE; ++pc) fold_piece(pc);
                };
                auto dma = [&](int i) {                           // the next tile's LDS-DMA instruction that rides behind matrix instruction i of the block
                    if (SPREAD && !(MX_ABL & 4) && mt == 0 && 4 * J + i < NDMA) issue_piece(t + 1, buf ^ 1, 4 * J + i);
                };
                accf = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, F[0]), thi[2 * J], J == 0 ? zf : accf, 0, 0, 0);
                dma(0);
                rd(0, J + 1);
                pieces(4 * J);
                acci = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[2], tlo8[J], J == 0 ? zi : acci, 0, 0, 0);
                dma(1);
                rd(2, J + 1);
                pieces(4 * J + 1);
                if constexpr (second) {
                    accf = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, F[1]), thi[second ? 2 * J + 1 : 0], accf, 0, 0, 0);
                    dma(2);
                }
                rd(1, J + 1);
                pieces(4 * J + 2);
                acci = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[3], thi8[J], acci, 0, 0, 0);
                dma(second ? 3 : 2);
                rd(3, J + 1);
                pieces(4 * J + 3);
                // issue order of the block: matrix instruction, the read behind it, its share of the fold
                constexpr int NM = second ? 4 : 3;
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (SPREAD && !(MX_ABL & 4) && mt == 0 && 4 * J + i < NDMA) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    if (rdm || mt + 1 < MTT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (!(MX_ABL & 1)) __builtin_amdgcn_sched_group_barrier(0x002, 3 * PP, 0);
                }
                if constexpr (!second) { if (rdm || mt + 1 < MTT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            }, std::make_integer_sequence<int, NB>{});
            __builtin_amdgcn_sched_barrier(0);
            // pieces the block's instruction count did not reach (short contractions)
#pragma unroll
            for (int pc = 4 * NB * PP; pc < NPIECE; ++pc) fold_piece(pc);
            have = true; lim_acc = lim; mt_acc = mt; t_acc = t;
        }
        // the next tile has landed (this wave's pieces) and every wave is done reading this buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(MX_ABL & 8)) __builtin_amdgcn_s_barrier();
        buf ^= 1;
    };
    // every tile but a segment's last is whole (a partial tile's first block still combines the whole tile before it: its mask
    // is computed from that block's own limits and strikes nothing)
    const int t_full = min(vt1, sg.n_vocab / TW);
    for (int t = vt0; t < t_full; ++t) tile(IC<0>{}, t);
    for (int t = max(vt0, t_full); t < vt1; ++t) tile(IC<1>{}, t);
    // the last block: combine, fold
    if (have) combine(IC<1>{});
#pragma unroll
    for (int pc = 0; pc < NPIECE; ++pc) fold_piece(pc);
    const float m2 = __shfl_xor(m, 32), s2 = __shfl_xor(s, 32);
    {
        const float mm = fmaxf(m, m2);
        s = s * __builtin_amdgcn_exp2f(m - mm) + s2 * __builtin_amdgcn_exp2f(m2 - mm);
        m = mm * LN2;
    }
    if (hf == 0 && row_ok) part_row[prow] = make_float2(m, s);
}

// (Round 5 tried TWO accumulator pairs at 32 rows per wave -- the finished pair combined, max-ed and exponentiated in place between the
//  next block's matrix instructions, no combine burst, no v[] -- for every contraction length: 28.2-28.9 vs 28.6-29.6 us at k = 100,
//  nothing at k = 200 / 50 and on the three-segment launch (profiles/r05_a_acc2.txt).  The two-pair scheme lives on in the wide
//  kernel, jlm_mixed_w.hip.)

// (Round 3 also tried FOUR waves of 64 rows -- two 32-row sets per wave, every fragment feeding two matrix instructions, one wave per
//  SIMD owning the 512-register file with two accumulator sets: half the LDS reads, and no faster: 38.1 / 32.8 / 30.8 us against
//  32.7 / 29.3 / 26.1 for the k = 200 / 100 / 50 segments alone.  With nobody to cover them the wave's LDS waits, the ~60 cycles each
//  LDS-DMA instruction costs to issue and its VALU stream are all exposed.  DESIGN.md 6c.2; the code is in the history: mx_body2.)

}  // namespace jlm_mx
