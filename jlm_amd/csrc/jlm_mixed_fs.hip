// jlm_mixed_fs.hip -- round 5, EXPERIMENT (JLM_MX_FS=1): the mixed-row vocabulary kernel with FORMAT-SLICED WAVE PAIRS.
// Reference: project + softmax, decoder/model.py:141-193, 15-20.
//
// the eight-wave kernel (32 rows per wave) pays one 1-KB LDS fragment per matrix instruction; the wide kernel (64
// rows per wave, one wave per SIMD) halves that and then has nobody to issue its VALU work beside the matrix instructions.  Here the
// two waves of a SIMD own the SAME 64 hypothesis rows (two 32-row sets) and split the work by operand format:
//   wave p     (p < 4): the f16 hi.hi products -- row operands thi[2][NS16]            (104 registers at k = 200)
//   wave p + 4        : the int8 cross terms   -- row operands thi8[2][NB], tlo8[2][NB] (112 registers)
// so every fragment feeds two matrix instructions AND each wave's VALU work runs beside its partner's matrix instructions.  The
// price: per 32-word block the f16 wave hands its set-1 accumulator to the partner and takes the partner's set-0 integer accumulator
// (4 KB each way through LDS, one workgroup barrier per block); each wave then combines and folds ONE row set exactly as mx_body does
// -- same products, same order: the slices are bit-identical to the eight-wave kernel's.
// Tiles are single 32-word blocks in a ring of S stages (the per-block barrier is the ring's barrier too).
#include "jlm_common.h"
#include <type_traits>
#include <utility>

#include "jlm_mixed_body.h"
using namespace jlm_mx;

#ifndef MXF_ABL
#define MXF_ABL 0      // measurement builds (wrong numbers): 1 no fold, 2 no exchange, 4 no barrier
#endif

namespace {

template <int NB, int NS16, int S>
__device__ __forceinline__ void mx_body_fs(const MxSeg &sg, int vt0, int vt1, int pt, int n_paths, const unsigned char *__restrict__ Tm, int ldt,
                                           float2 *__restrict__ part_row, unsigned char *smem) {
#pragma clang fp contract(off)
    constexpr float LN2 = 0.6931471805599453f;
    constexpr int ROWB = NB * 128;                       // bytes per mixed row
    constexpr int BLKB = 32 * ROWB;                      // bytes per 32-word block
    constexpr int XOFF = S * BLKB;                       // exchange area: [pair 4][direction 2][parity 2][4 KB]
    constexpr int OLD_MTT = mx_blocks_per_tile(NB);      // the launcher's sub-ranges are in tiles of this many blocks
    constexpr int PF = (NB + 1) / 2, PI = NB / 2;        // LDS-DMA pieces per block: f16 wave (even 32-k blocks), int8 wave (odd)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, pair = wave & 3;         // role 0: f16 products, folds set 0; role 1: int8 products, folds set 1
    const int hf = lane >> 5, li = lane & 31;
    const int prow0 = pt * 256 + pair * 64 + li, prow1 = prow0 + 32;
    const bool ok0 = prow0 < n_paths, ok1 = prow1 < n_paths;
    const int prow_own = role ? prow1 : prow0;
    const bool ok_own = role ? ok1 : ok0;
    const unsigned char *tblk0 = Tm + (ok0 ? mx_tm_block(prow0, ldt) : 0);
    const unsigned char *tblk1 = Tm + (ok1 ? mx_tm_block(prow1, ldt) : 0);
    const unsigned char *tb0 = tblk0 + mx_tm_granule(sg.tm_off, hf, prow0);
    const unsigned char *tb1 = tblk1 + mx_tm_granule(sg.tm_off, hf, prow1);
    const float s_t = *reinterpret_cast<const float *>((role ? tblk1 : tblk0) + mx_tm_scale(ldt, prow_own, sg.seg));
    const float csr = s_t * sg.cs;
    const float descale = sg.descale;
    const int b0 = vt0 * OLD_MTT;
    const int nblk = (sg.n_vocab + 31) >> 5;
    const int b1 = min(vt1 * OLD_MTT, nblk);

    // ---- LDS-DMA: pair p brings row group p (8 words) of every block: piece j (32-k block j, 1 KB) by the f16 wave for even j, by the
    //      int8 wave for odd j; lane = (row lane >> 3, slot lane & 7), source granule = slot ^ ((row >> 1) & 7) as in mx_body
    const unsigned long long bptr = reinterpret_cast<unsigned long long>(sg.B);
    const unsigned long long bptr_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bptr >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)bptr);
    const int nrec = __builtin_amdgcn_readfirstlane(sg.n_vocab) * ROWB;
    const int r8 = lane >> 3, dslot = lane & 7;
    const int drow = 8 * pair + r8;
    const int dvoff = drow * ROWB + ((dslot ^ ((drow >> 1) & 7)) * 16);
    auto issue = [&](int b, int slot, auto role_c) {
        constexpr int ROLE = decltype(role_c)::value;
        // (a block at or past the sub-range's end is requested through a zero-sized descriptor: straight-line code, no traffic)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(bptr_u), 0, b < b1 ? nrec : 0, 0x00020000);
        unsigned char *dst = smem + slot * BLKB + (pair * NB) * 1024;
        const int voff = dvoff + b * BLKB;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if ((j & 1) == ROLE)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, voff, j * 128, 0, 0);
    };
    // fragment addresses (block image as mx_body's): row li: piece (li / 8, j); inside it row li % 8, granule g ^ ((li >> 1) & 7)
    const int x = (li >> 1) & 7;
    const int fbase = (li >> 3) * (NB * 1024) + (li & 7) * 128;
    int goff[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) goff[g4] = fbase + ((2 * g4 + hf) ^ x) * 16;
    const int xlane = lane * 16;
    unsigned char *xsend = smem + XOFF + ((pair * 2 + role) * 2) * 4096 + xlane;            // this wave writes direction `role`
    const unsigned char *xrecv = smem + XOFF + ((pair * 2 + (role ^ 1)) * 2) * 4096 + xlane;  // ... and reads the partner's

    float m = JLM_NEG_BIG, s = 0.0f;
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = -1.0e30f;
    float tmax, nmn, sc_old, add0, add1;
    constexpr int NPIECE = 8 + 1 + 16 + 1;
    auto fold_piece = [&](int pc) {
        if (MXF_ABL & 1) { if (pc == 0) asm volatile("" :: "v"(v[0]), "v"(v[5]), "v"(v[10]), "v"(v[15])); return; }
        if (pc < 8) {
            const float t2 = fmaxf(v[2 * pc], v[2 * pc + 1]);
            tmax = pc == 0 ? t2 : fmaxf(tmax, t2);
        } else if (pc == 8) {
            const float mn = fmaxf(m, tmax * descale);
            nmn = -mn;
            sc_old = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            add0 = 0.0f; add1 = 0.0f;
        } else if (pc < 25) {
            const int r = pc - 9;
            const float e = __builtin_amdgcn_exp2f(fmaf(v[r], descale, nmn));
            if (r & 1) add1 += e; else add0 += e;
        } else if (pc == 25) {
            s = fmaf(s, sc_old, add0 + add1);              // (explicit: the two roles' copies of the fold must round alike)
        }
    };
    const f32x16 zf = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const i32x16 zi = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const i32x4 z4 = {0, 0, 0, 0};

    constexpr int NF = 4;                                 // fragment ring: every ds_read_b128 is issued NF fragments (2 NF matrix instructions) ahead
    const int bfull = min(b1, sg.n_vocab >> 5);           // blocks [b0, bfull) are whole; the segment's last one may be partial (masked combine)
    if (role == 0) {
        // ================================================================================== the f16 wave
        f16x8 thi0[NS16], thi1[NS16];
        {
            i32x4 raw0[NS16], raw1[NS16];
#pragma unroll
            for (int q = 0; q < NS16; ++q) {
                raw0[q] = *reinterpret_cast<const i32x4 *>(tb0 + ((q >> 1) * 8 + 2 * (q & 1)) * 512);
                raw1[q] = *reinterpret_cast<const i32x4 *>(tb1 + ((q >> 1) * 8 + 2 * (q & 1)) * 512);
            }
#pragma unroll
            for (int q = 0; q < NS16; ++q) {
                thi0[q] = __builtin_bit_cast(f16x8, ok0 ? raw0[q] : z4);
                thi1[q] = __builtin_bit_cast(f16x8, ok1 ? raw1[q] : z4);
            }
        }
#pragma unroll
        for (int i = 0; i < S; ++i) issue(b0 + i, i, IC<0>{});
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * PF) : "memory");
        __builtin_amdgcn_s_barrier();
        constexpr int NMF = 2 * NS16;
        constexpr int PP = (NPIECE + NMF - 1) / NMF;
        i32x4 F[NF];
        auto rdfrag = [&](const unsigned char *bs, int q) { return *reinterpret_cast<const i32x4 *>(bs + (q >> 1) * 1024 + goff[q & 1]); };
#pragma unroll
        for (int i = 0; i < NF && i < NS16; ++i) F[i] = rdfrag(smem, i);
        int slot = 0;
        auto blk = [&](auto masked_c, int b) {
            constexpr bool MASKED = decltype(masked_c)::value != 0;
            const unsigned char *bs = smem + slot * BLKB;
            const int nslot = slot + 1 == S ? 0 : slot + 1;
            const int par = (b - b0) & 1;
            f32x16 accA = zf, accB = zf;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NS16; ++q) {
                accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, F[q % NF]), thi0[q], accA, 0, 0, 0);
#pragma unroll
                for (int pc = (2 * q) * PP; pc < (2 * q + 1) * PP && pc < NPIECE; ++pc) fold_piece(pc);
                accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, F[q % NF]), thi1[q], accB, 0, 0, 0);
                if (q + NF < NS16) F[q % NF] = rdfrag(bs, q + NF);
#pragma unroll
                for (int pc = (2 * q + 1) * PP; pc < (2 * q + 2) * PP && pc < NPIECE; ++pc) fold_piece(pc);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (!(MXF_ABL & 1)) __builtin_amdgcn_sched_group_barrier(0x002, 3 * PP, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (q + NF < NS16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (!(MXF_ABL & 1)) __builtin_amdgcn_sched_group_barrier(0x002, 3 * PP, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pc = NMF * PP; pc < NPIECE; ++pc) fold_piece(pc);
            // hand set 1 to the partner, take the partner's set-0 integer accumulator
            if (!(MXF_ABL & 2)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w = {accB[4 * q], accB[4 * q + 1], accB[4 * q + 2], accB[4 * q + 3]};
                    *reinterpret_cast<f32x4 *>(xsend + par * 4096 + q * 1024) = w;
                }
            }
            if (S == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((S - 2) * PF) : "memory");
            if (!(MXF_ABL & 4)) __builtin_amdgcn_s_barrier();
            issue(b + S, slot, IC<0>{});                   // every wave is done with this block: its slot takes block b + S
            i32x4 g[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = (MXF_ABL & 2) ? z4 : *reinterpret_cast<const i32x4 *>(xrecv + par * 4096 + q * 1024);
#pragma unroll
            for (int i = 0; i < NF && i < NS16; ++i) F[i] = rdfrag(smem + nslot * BLKB, i);      // the next block's first fragments
            const int lim = sg.n_vocab - b * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float y = fmaf((float)g[r >> 2][r & 3], csr, accA[r]);
                v[r] = (MASKED && (r & 3) + 8 * (r >> 2) + 4 * hf >= lim) ? JLM_NEG_BIG : y;
            }
            slot = nslot;
        };
        for (int b = b0; b < bfull; ++b) blk(IC<0>{}, b);
        for (int b = max(b0, bfull); b < b1; ++b) blk(IC<1>{}, b);
    } else {
        // ================================================================================== the int8 wave
        i32x4 thi8_0[NB], tlo8_0[NB], thi8_1[NB], tlo8_1[NB];
        {
            i32x4 raw[4 * NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                raw[4 * j] = *reinterpret_cast<const i32x4 *>(tb0 + (j * 8 + 4) * 512);
                raw[4 * j + 1] = *reinterpret_cast<const i32x4 *>(tb0 + (j * 8 + 6) * 512);
                raw[4 * j + 2] = *reinterpret_cast<const i32x4 *>(tb1 + (j * 8 + 4) * 512);
                raw[4 * j + 3] = *reinterpret_cast<const i32x4 *>(tb1 + (j * 8 + 6) * 512);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                thi8_0[j] = ok0 ? raw[4 * j] : z4;
                tlo8_0[j] = ok0 ? raw[4 * j + 1] : z4;
                thi8_1[j] = ok1 ? raw[4 * j + 2] : z4;
                tlo8_1[j] = ok1 ? raw[4 * j + 3] : z4;
            }
        }
#pragma unroll
        for (int i = 0; i < S; ++i) issue(b0 + i, i, IC<1>{});
        if (PI == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * PI) : "memory");
        __builtin_amdgcn_s_barrier();
        constexpr int NMF = 4 * NB;
        constexpr int PP = (NPIECE + NMF - 1) / NMF;
        constexpr int NFR = 2 * NB;                        // fragments of a block: (32-k block j, hi8 / lo8) = 2 j, 2 j + 1
        i32x4 F[NF];
        auto rdfrag = [&](const unsigned char *bs, int f) { return *reinterpret_cast<const i32x4 *>(bs + (f >> 1) * 1024 + goff[2 + (f & 1)]); };
#pragma unroll
        for (int i = 0; i < NF && i < NFR; ++i) F[i] = rdfrag(smem, i);
        int slot = 0;
        auto blk = [&](auto masked_c, int b) {
            constexpr bool MASKED = decltype(masked_c)::value != 0;
            const unsigned char *bs = smem + slot * BLKB;
            const int nslot = slot + 1 == S ? 0 : slot + 1;
            const int par = (b - b0) & 1;
            i32x16 accA = zi, accB = zi;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NFR; ++f) {
                // fragment f = (block j = f / 2, hi8 | lo8): B's hi8 with the rows' lo8, B's lo8 with the rows' hi8
                const int j = f >> 1;
                accA = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[f % NF], (f & 1) ? thi8_0[j] : tlo8_0[j], accA, 0, 0, 0);
#pragma unroll
                for (int pc = (2 * f) * PP; pc < (2 * f + 1) * PP && pc < NPIECE; ++pc) fold_piece(pc);
                accB = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[f % NF], (f & 1) ? thi8_1[j] : tlo8_1[j], accB, 0, 0, 0);
                if (f + NF < NFR) F[f % NF] = rdfrag(bs, f + NF);
#pragma unroll
                for (int pc = (2 * f + 1) * PP; pc < (2 * f + 2) * PP && pc < NPIECE; ++pc) fold_piece(pc);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (!(MXF_ABL & 1)) __builtin_amdgcn_sched_group_barrier(0x002, 3 * PP, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (f + NF < NFR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (!(MXF_ABL & 1)) __builtin_amdgcn_sched_group_barrier(0x002, 3 * PP, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pc = NMF * PP; pc < NPIECE; ++pc) fold_piece(pc);
            if (!(MXF_ABL & 2)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const i32x4 w = {accA[4 * q], accA[4 * q + 1], accA[4 * q + 2], accA[4 * q + 3]};
                    *reinterpret_cast<i32x4 *>(xsend + par * 4096 + q * 1024) = w;
                }
            }
            if (S == 2 || PI == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((S - 2) * PI) : "memory");
            if (!(MXF_ABL & 4)) __builtin_amdgcn_s_barrier();
            issue(b + S, slot, IC<1>{});
            f32x4 g[4];
            const f32x4 zf4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = (MXF_ABL & 2) ? zf4 : *reinterpret_cast<const f32x4 *>(xrecv + par * 4096 + q * 1024);
#pragma unroll
            for (int i = 0; i < NF && i < NFR; ++i) F[i] = rdfrag(smem + nslot * BLKB, i);
            const int lim = sg.n_vocab - b * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float y = fmaf((float)accB[r], csr, g[r >> 2][r & 3]);
                v[r] = (MASKED && (r & 3) + 8 * (r >> 2) + 4 * hf >= lim) ? JLM_NEG_BIG : y;
            }
            slot = nslot;
        };
        for (int b = b0; b < bfull; ++b) blk(IC<0>{}, b);
        for (int b = max(b0, bfull); b < b1; ++b) blk(IC<1>{}, b);
    }
    // the last block's logits
#pragma unroll
    for (int pc = 0; pc < NPIECE; ++pc) fold_piece(pc);
    const float m2 = __shfl_xor(m, 32), s2 = __shfl_xor(s, 32);
    {
        const float mm = fmaxf(m, m2);
        s = s * __builtin_amdgcn_exp2f(m - mm) + s2 * __builtin_amdgcn_exp2f(m2 - mm);
        m = mm * LN2;
    }
    if (hf == 0 && ok_own) part_row[prow_own] = make_float2(m, s);
}

constexpr int fs_stages(int nb) { return nb >= 5 ? 3 : nb >= 3 ? 5 : 10; }      // 3 / 5 / 10 stages of 28 / 16 / 8 KB: 84 / 80 / 80 KB + 64 KB of exchange

__global__ __launch_bounds__(512, 1) void vocab_lse_mixedfs_kernel(MxArgs a, const unsigned char *__restrict__ Tm, int ld_tm, float2 *__restrict__ part,
                                                                   int ld_part, int n_rows_max, const int *n_dev, int n_ptiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fs_smem[];
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
        const int ns16 = (sg.k + 2 + 15) >> 4;
        if (sg.nb == 7 && ns16 == 13) mx_body_fs<7, 13, fs_stages(7)>(sg, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, fs_smem);
        else if (sg.nb == 4 && ns16 == 7) mx_body_fs<4, 7, fs_stages(4)>(sg, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, fs_smem);
        else if (sg.nb == 2 && ns16 == 4) mx_body_fs<2, 4, fs_stages(2)>(sg, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, fs_smem);
    }
}

}  // namespace

// the D-softmax* shapes (k + 2 = 202, 102, 52) only; same arguments and slices as the eight-wave kernel.  0, or a negative error.
int jlm_mx_fs_launch(const MxArgs &a, const void *Tm, int ld_tm, float2 *part, int ld_part, int n_rows_max, const int *n_dev, int n_ptiles,
                     hipStream_t st) {
    int lds = 0;
    for (int i = 0; i < a.n_segs; ++i) {
        const int nb = a.seg[i].nb, ns16 = (a.seg[i].k + 2 + 15) / 16;
        if (!((nb == 7 && ns16 == 13) || (nb == 4 && ns16 == 7) || (nb == 2 && ns16 == 4))) return -2;
        const int l = fs_stages(nb) * 32 * nb * 128 + 65536;
        if (l > lds) lds = l;
    }
    static JlmLdsGrant grant;
    if (int rc = jlm_grant_lds(grant, reinterpret_cast<const void *>(vocab_lse_mixedfs_kernel), lds)) return rc;
    hipLaunchKernelGGL(vocab_lse_mixedfs_kernel, dim3(a.n_cols * n_ptiles), dim3(512), lds, st, a, reinterpret_cast<const unsigned char *>(Tm), ld_tm,
                       part, ld_part, n_rows_max, n_dev, n_ptiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(int)e - 100;
    return 0;
}
