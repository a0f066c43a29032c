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

template <int NB, int NS16, int MTT>
struct MxWide {
    static_assert(MTT % 2 == 0, "blocks alternate between two accumulator pairs");
    static constexpr int ROWB = NB * 128;
    static constexpr int TW = 32 * MTT;
    static constexpr int BUFB = TW * ROWB;
    static constexpr int NMF = NS16 + 2 * NB;             // matrix instructions of a block and row set
    static constexpr int NSLOT = 8 * NB;                  // issue slots of a block (two per fragment; NS16 odd: two of the last 32-k block's are empty)
    static constexpr int NPIECE = 2 * 42;                 // per row set: 16 x (cvt, fma), 8 x max3, 1, 16 x (fma, exp2, add), 1
    static constexpr int PP = (NPIECE + NSLOT - 3) / (NSLOT - 2);        // (slots 0, 1 carry nothing: the pairs' last instructions are still in the pipe)

    // row operands: MFMA B operands, in accumulation registers for the whole sub-range
    f16x8 thi[2][NS16];
    i32x4 thi8[2][NB], tlo8[2][NB];
    float csr[2], descale;
    float m[2], s[2];
    float tmax[2], nmn[2], sc_old[2], add0[2], add1[2];
    f32x16 fa[2], fb[2];                                  // pair a (even blocks) / b (odd blocks) of each row set
    i32x16 ia[2], ib[2];
    i32x4 F[8];                                           // fragment ring: 32-k block J lives in F[4 (J & 1) ..]
    int goff[4];
    int hf;
    unsigned char *smem;

    // one piece of the treatment of the finished pair (pf, pi) of row set S
    template <bool MASKED>
    __device__ __forceinline__ void fold_piece(int S, f32x16 &pf, const i32x16 &pi, int mtp, int lim, int pc) {
        if (pc < 16) {
            const int r = pc;
            const float y = fmaf((float)pi[r], csr[S], pf[r]);
            pf[r] = (MASKED && mtp * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf >= lim) ? JLM_NEG_BIG : y;
        } else if (pc < 24) {
            const int q = pc - 16;
            const float t2 = fmaxf(pf[2 * q], pf[2 * q + 1]);
            tmax[S] = q == 0 ? t2 : fmaxf(tmax[S], t2);
        } else if (pc == 24) {
            const float mn = fmaxf(m[S], tmax[S] * descale);
            nmn[S] = -mn;
            sc_old[S] = __builtin_amdgcn_exp2f(m[S] - mn);
            m[S] = mn;
            add0[S] = 0.0f; add1[S] = 0.0f;
        } else if (pc < 41) {
            const int r = pc - 25;
            const float e = (MXW_ABL & 2) ? fmaf(pf[r], descale, nmn[S]) : __builtin_amdgcn_exp2f(fmaf(pf[r], descale, nmn[S]));
            if (r & 1) add1[S] += e; else add0[S] += e;
        } else if (pc == 41) {
            s[S] = s[S] * sc_old[S] + (add0[S] + add1[S]);
        }
    }
    // pieces alternate between the two row sets: piece 2 q -> set 0 piece q, 2 q + 1 -> set 1 piece q
    template <bool MASKED>
    __device__ __forceinline__ void fold2(f32x16 (&pf)[2], const i32x16 (&pi)[2], int mtp, int lim, int pc2) {
        if (MXW_ABL & 1) {           // (the finished accumulators stay "used": without this the compiler drops every matrix instruction)
            if (pc2 == 0) asm volatile("" :: "v"(pf[0]), "v"(pi[0]), "v"(pf[1]), "v"(pi[1]));
            return;
        }
        if ((MXW_ABL & 16) && (pc2 >> 1) >= 16) {      // combine only
            if (pc2 == 32) asm volatile("" :: "v"(pf[0]), "v"(pf[1]));
            return;
        }
        fold_piece<MASKED>(pc2 & 1, pf[pc2 & 1], pi[pc2 & 1], mtp, lim, pc2 >> 1);
    }

    // one 32-word block: its matrix instructions into (wf, wi); the finished pairs (pf, pi) treated between them; tile t + 1's
    // LDS-DMA instruction (mt NB + J) behind the first matrix instruction of 32-k block J
    template <bool MASKED, int mt>
    __device__ __forceinline__ void block(int buf, int lim_p, f32x16 (&wf)[2], i32x16 (&wi)[2], f32x16 (&pf)[2], i32x16 (&pi)[2],
                                          const __amdgpu_buffer_rsrc_t rs_next, int voff_next, int wave) {
        constexpr int mtp = (mt + MTT - 1) % MTT;
        const unsigned char *bs = smem + buf * BUFB + mt * (4 * NB * 1024);
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
                if (q < 2) return;
#pragma unroll
                for (int pc = (q - 2) * PP; pc < (q - 1) * PP && pc < NPIECE; ++pc) fold2<MASKED>(pf, pi, mtp, lim_p, pc);
            };
            // DMA piece (mt NB + J) of the next tile: row group wave + 4 i, 32-k block j
            if (!(MXW_ABL & 4)) {
                constexpr int q = mt * NB + J, i = q / NB, j = q % NB;
                unsigned char *dst = smem + (buf ^ 1) * BUFB + ((wave + 4 * i) * NB) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_next, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16,
                                                         voff_next + i * (32 * ROWB), j * 128, 0, 0);
            }
            wf[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, F[R + 0]), thi[0][2 * J], J == 0 ? zf : wf[0], 0, 0, 0);
            pieces(8 * J);
            wf[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, F[R + 0]), thi[(MXW_ABL & 64) ? 0 : 1][2 * J], J == 0 ? zf : wf[1], 0, 0, 0);
            rd(0);
            pieces(8 * J + 1);
            wi[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[R + 2], tlo8[0][J], J == 0 ? zi : wi[0], 0, 0, 0);
            pieces(8 * J + 2);
            wi[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[R + 2], tlo8[(MXW_ABL & 64) ? 0 : 1][J], J == 0 ? zi : wi[1], 0, 0, 0);
            rd(2);
            pieces(8 * J + 3);
            if constexpr (second) {
                wf[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, F[R + 1]), thi[0][second ? 2 * J + 1 : 0], wf[0], 0, 0, 0);
                pieces(8 * J + 4);
                wf[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, F[R + 1]), thi[(MXW_ABL & 64) ? 0 : 1][second ? 2 * J + 1 : 0], wf[1], 0, 0, 0);
            } else {
                pieces(8 * J + 4);
            }
            rd(1);
            pieces(8 * J + 5);
            wi[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[R + 3], thi8[0][J], wi[0], 0, 0, 0);
            pieces(8 * J + 6);
            wi[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[R + 3], thi8[(MXW_ABL & 64) ? 0 : 1][J], wi[1], 0, 0, 0);
            rd(3);
            pieces(8 * J + 7);
            // issue order: matrix instruction, [the LDS-DMA instruction], [a fragment read], its share of the fold
            constexpr bool reads = (J + 1 < NB) || (mt + 1 < MTT);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool has_m = second || (i != 4 && i != 5);
                if (has_m) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i == 0 && !(MXW_ABL & 4)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if ((i & 1) && reads && !(MXW_ABL & 32)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (8 * J + i >= 2 && !(MXW_ABL & 1)) __builtin_amdgcn_sched_group_barrier(0x002, 3 * PP, 0);
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
        // ---- 1. row operands of both sets, straight into accumulation registers.  A row past the end reads its lane's slot of
        //         block 0 (inside the buffer whatever its size; its result is not stored and no other row sees it): selecting zeros would cost
        //         a VALU move per register.
        bool row_ok[2];
        int prow[2];
#pragma unroll
        for (int S = 0; S < 2; ++S) {
            prow[S] = pt * 256 + wave * 64 + S * 32 + li;
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
        const int r8 = lane >> 3, dslot = lane & 7;
        const int drow = 8 * wave + r8;
        const int dvoff = drow * ROWB + ((dslot ^ ((drow >> 1) & 7)) * 16);
        constexpr int NDMA = MTT * NB;
        const int x = (li >> 1) & 7;
        const int fbase = (li >> 3) * (NB * 1024) + (li & 7) * 128;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) goff[g4] = fbase + ((2 * g4 + hf) ^ x) * 16;
#pragma unroll
        for (int S = 0; S < 2; ++S) {
            m[S] = JLM_NEG_BIG; s[S] = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { fb[S][r] = -1.0e30f; ib[S][r] = 0; fa[S][r] = 0.0f; ia[S][r] = 0; }
        }
        // the first tile
        {
            const int voff = dvoff + vt0 * (TW * ROWB);
#pragma unroll
            for (int q = 0; q < NDMA; ++q) {
                const int i = q / NB, j = q % NB;
                unsigned char *dst = smem + ((wave + 4 * i) * NB) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, voff + i * (32 * ROWB),
                                                         j * 128, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int S = 0; S < 2; ++S) {
#pragma unroll
            for (int q = 0; q < NS16; ++q) asm volatile("" : "+a"(thi[S][q]));
#pragma unroll
            for (int j = 0; j < NB; ++j) { asm volatile("" : "+a"(thi8[S][j])); asm volatile("" : "+a"(tlo8[S][j])); }
        }
        __builtin_amdgcn_s_barrier();
        MXW_STAMP(1);
        int buf = 0;
        auto tile = [&](auto masked_c, int t) {
            constexpr bool MASKED = decltype(masked_c)::value != 0;
            const bool more = t + 1 < vt1;
            // (behind a sub-range's last tile: a descriptor of zero records -- the instructions stay in the stream, nothing is fetched)
            const __amdgpu_buffer_rsrc_t rs_next = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(bptr_u), 0, more ? nrec : 0, 0x00020000);
            const int voff_next = dvoff + (t + 1) * (TW * ROWB);
            const int lim = sg.n_vocab - t * TW;
            {
                const unsigned char *bs0 = smem + buf * BUFB;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) F[g4] = *reinterpret_cast<const i32x4 *>(bs0 + goff[g4]);
            }
            mx_for_each_ic([&](auto hc) {
                constexpr int mt0 = 2 * decltype(hc)::value;
                // (the pairs treated in a tile's first block belong to the tile before -- whole -- or are the dummy: never masked)
                if constexpr (mt0 == 0) block<false, mt0>(buf, lim, fa, ia, fb, ib, rs_next, voff_next, wave);
                else block<MASKED, mt0>(buf, lim, fa, ia, fb, ib, rs_next, voff_next, wave);
                block<MASKED, mt0 + 1>(buf, lim, fb, ib, fa, ia, rs_next, voff_next, wave);
            }, std::make_integer_sequence<int, MTT / 2>{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(MXW_ABL & 8)) __builtin_amdgcn_s_barrier();
            buf ^= 1;
            MXW_STAMP(2 + t - vt0);
        };
        const int t_full = min(vt1, sg.n_vocab / TW);
        for (int t = vt0; t < t_full; ++t) tile(IC<0>{}, t);
        for (int t = max(vt0, t_full); t < vt1; ++t) tile(IC<1>{}, t);
        {
            const int lim_last = sg.n_vocab - (vt1 - 1) * TW;
#pragma unroll
            for (int pc = 0; pc < NPIECE; ++pc) fold2<true>(fb, ib, MTT - 1, lim_last, pc);
        }
#pragma unroll
        for (int S = 0; S < 2; ++S) {
            const float m2 = __shfl_xor(m[S], 32), s2 = __shfl_xor(s[S], 32);
            const float mm = fmaxf(m[S], m2);
            const float ss = s[S] * __builtin_amdgcn_exp2f(m[S] - mm) + s2 * __builtin_amdgcn_exp2f(m2 - mm);
            if (hf == 0 && row_ok[S]) part_row[prow[S]] = make_float2(mm * LN2, ss);
        }
        MXW_STAMP(63);
    }
};

// (out of line: hosted inline, the three bodies of the D-softmax* kernel cost each other 64 spilled registers, some inside the tile loops)
template <int NB, int NS16>
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
    u.bias2 = nullptr;
    MxWide<NB, NS16, mx_blocks_per_tile(NB)> w;
    w.run(u, uni(vt0), uni(vt1), uni(pt), uni(n_paths), static_cast<const unsigned char *>(unip(Tm)), uni(ld_tm),
          static_cast<float2 *>(const_cast<void *>(unip(prow))), static_cast<unsigned char *>(const_cast<void *>(unip(smem))));
}

template <int... SH>      // SH = NB0, NS0, NB1, NS1, ...
struct MxwDispatch;
template <>
struct MxwDispatch<> {
    static __device__ __forceinline__ void run(const MxSeg &, int, int, int, int, int, const unsigned char *, int, float2 *, unsigned char *) {}
};
template <int NB, int NS16, int... REST>
struct MxwDispatch<NB, NS16, REST...> {
    static __device__ __forceinline__ void run(const MxSeg &sg, int ns16, int vt0, int vt1, int pt, int n_paths, const unsigned char *Tm, int ld_tm,
                                               float2 *prow, unsigned char *smem) {
        if (sg.nb == NB && ns16 == NS16) {
            mxw_body<NB, NS16>(sg, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, smem);
        } else {
            MxwDispatch<REST...>::run(sg, ns16, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, smem);
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

template <int... SH>
__global__ __launch_bounds__(256, 1) void vocab_lse_mixedw_kernel(MxArgs a, const unsigned char *__restrict__ Tm, int ld_tm, float2 *__restrict__ part,
                                                                  int ld_part, int n_rows_max, const int *n_dev, int n_ptiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mxw_smem[];
    const int n_paths = n_dev ? min(*n_dev, n_rows_max) : n_rows_max;
    const int b = blockIdx.x;
    int p, pt;
    const int nb8 = (a.n_cols & ~7) * n_ptiles;
    if (b < nb8) { const int x = b & 7, jb = b >> 3; p = (jb / n_ptiles) * 8 + x; pt = jb % n_ptiles; }
    else { const int bb = b - nb8; p = (a.n_cols & ~7) + bb / n_ptiles; pt = bb % n_ptiles; }
    if (p >= a.n_cols || pt * 256 >= n_paths) return;
    MXW_WG_T0
    for (int r = a.col_first[p]; r < a.col_first[p + 1]; ++r) {
        const MxSeg sg = a.seg[a.sub_seg[r]];
        const int vt0 = a.sub_t0[r], vt1 = a.sub_t1[r];
        float2 *prow = part + (size_t)r * ld_part;
        if (r != a.col_first[p]) __syncthreads();
        const int ns16 = (sg.k + 2 + 15) >> 4;
#ifdef JLM_WGTIME
        si_last = a.sub_seg[r];
#endif
        MxwDispatch<SH...>::run(sg, ns16, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, mxw_smem);
    }
    MXW_WG_T1
}
#define MXW_KERNEL_DSOFTMAX vocab_lse_mixedw_kernel<7, 13, 4, 7, 2, 4>

#ifdef JLM_MX_RESOURCES
#define MXW_RES(NB_, NS_) __global__ __launch_bounds__(256, 1) void mxw_res_##NB_##_##NS_(MxSeg sg, const unsigned char *Tm, int ld_tm, float2 *part) { \
        extern __shared__ __attribute__((aligned(16))) unsigned char sm_[]; MxWide<NB_, NS_, mx_blocks_per_tile(NB_)> w; \
        w.run(sg, 0, 100, blockIdx.x, 2560, Tm, ld_tm, part, sm_); }
MXW_RES(2, 4) MXW_RES(4, 7) MXW_RES(7, 13)
#endif

}  // namespace

#ifdef JLM_WGTIME
extern "C" int jlm_prof_read_wg_mxw(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(jlm_prof_wg_mxw), sizeof(jlm_prof_wg_mxw)) == hipSuccess ? 0 : -1;
}
#endif

// Launch of the wide kernel for the shapes it hosts (the D-softmax* 200 / 100 / 50 model: every segment in the bias-column form
// with (nb, f16 steps) in {(7, 13), (4, 7), (2, 4)}); called by jlm_vocab_lse_mixed (jlm_mixed.hip) with its column cuts.
// Returns 0, or -3 (LDS grant) / a negative HIP error like its caller.
int jlm_mx_wide_launch(const MxArgs &a, const void *Tm, int ld_tm, float2 *part, int ld_part, int n_rows_max, const int *n_dev, int n_ptiles,
                       int lds, hipStream_t st) {
    static JlmLdsGrant grant;
    if (int rc = jlm_grant_lds(grant, reinterpret_cast<const void *>(MXW_KERNEL_DSOFTMAX), lds)) return rc;
    hipLaunchKernelGGL(MXW_KERNEL_DSOFTMAX, dim3(a.n_cols * n_ptiles), dim3(256), lds, st, a, reinterpret_cast<const unsigned char *>(Tm), ld_tm, part,
                       ld_part, n_rows_max, n_dev, n_ptiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(int)e - 100;
    return 0;
}
