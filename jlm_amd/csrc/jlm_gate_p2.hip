// jlm_gate_p2.hip -- the fused LSTM step of the decode on 128-hypothesis x 256-gate-column tiles, a 2 x 2 register block per wave, persistent
// (round 6; jlm_lstm_step_xg's JLM_GATE_V=4 form: csrc/jlm_gate.hip picks by the launch's row bound).
// Reference: decoder/model.py:125-139 with the state gather of decoder/decoder.py:206-218.
//
// Why this tile.  The ablations of the 160 x 128 kernels (profiles/history/r03_a_gate_ablation.txt; HISTORY.md round 4) say a k-step of theirs is
// set by the L2 -> LDS path: 36 pieces of 1 KB per 32-value k-step and CU take ~1 300 cycles whatever else runs beside them, the 120 matrix
// instructions of the k-step 960.  Bytes per matrix instruction follow the tile, (rows + columns) / (rows x columns): 160 x 128 brings 307 B
// per instruction, 128 x 256 brings 256 B (48 pieces for 192 instructions = 1 536 matrix cycles), and a wave that owns 2 gate blocks x 2
// hypothesis blocks reads 8 fragments per 12 instructions where the 3 x 1 / 2 x 1 waves read 8 per 9 / 6 per 6.  The price is the grid:
// 8 column tiles x (rows / 128) row tiles -- 160 tiles at the decode's 2 560 rows (96 CUs idle: the one-tile kernel keeps that shape),
// 2.5 per CU at 10 240, 5 at 20 480.
//
//   * 8 waves = two per SIMD; wave w owns gate blocks 2 (w & 3), 2 (w & 3) + 1 and hypothesis blocks 2 (w >> 2), 2 (w >> 2) + 1: four
//     32 x 32 accumulators (64 registers), gate matrix = MFMA A operand, hypotheses = B operand, so that a lane's 16 accumulator registers
//     of a block are the four gates of four units of ONE hypothesis (the packed row order of csrc/jlm_gate.hip) and the cell update runs
//     in registers;
//   * two LDS rings, all 160 KB: the gate matrix's k-steps (256 rows x 128 B = 32 KB) in THREE slots -- its 512 KB per column tile stay in
//     the XCD's L2, one k-step of lead covers that latency -- and the gathered state rows' (128 x 128 B = 16 KB) in FOUR: they come over the
//     fabric and get two k-steps.  16 k-steps are not whole laps of three slots: the gate ring's slot is a run-time (uniform) number;
//   * one fragment set per wave (32 registers), refilled IN PLACE with the next half step's fragments right behind the matrix instruction
//     that read a register last (order below: every refill has >= 6 matrix instructions = 190 cycles of cover); stage kt + 1 is read during
//     k-step kt, so it has landed at the barrier on top of k-step kt and the slot refilled behind that barrier is the one of stage kt - 1;
//   * every wait is a counted s_waitcnt vmcnt(n), n derived at compile time from the issue order (p2_* functions below); loads the
//     compiler must not see (it would drain the queue in front of them) are inline asm, as in the other persistent kernels; stores are
//     issued by every lane (rows past the edge: a dump page) so that the counts are exact;
//   * the ring does not stop at a tile's end: the last k-steps of a tile request the first stages of the next one, whose row indices
//     were fetched under the first k-steps; a tile's own epilogue indices and operands (one 128-byte table line and the old cell state per
//     block) are fetched under its k-steps 0 / 2 / 4, 7, 10, 13.
#include "jlm_gate.h"

#ifdef JLM_PROFILE
// tools/probes/gate_p2_profile.py: waves 0 and 4 stamp, for a workgroup's first eight tiles, the shader clock at the top of the tile's
// k-steps [0], behind them [1], when the first epilogue operands are in [2], behind the cell update [3]; [4] the 100 MHz wall clock at [0]
static __device__ unsigned long long jlm_gate_p2_time[256][2][8][5];
#define JLM_P2_T(i) do { if ((threadIdx.x & 255) == 0 && tix < 8) { jlm_gate_p2_time[blockIdx.x & 255][threadIdx.x >> 8][tix][i] = clock64(); \
    if ((i) == 0) jlm_gate_p2_time[blockIdx.x & 255][threadIdx.x >> 8][tix][4] = wall_clock64(); } } while (0)
extern "C" int jlm_prof_read_gate_p2(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(jlm_gate_p2_time), sizeof(jlm_gate_p2_time)) == hipSuccess ? 0 : -1;
}
#else
#define JLM_P2_T(i) (void)0
#endif

// measurement builds (-DP2_ABL=bits; results wrong, the time is the answer): 1 no LDS-DMA in the k-steps, 2 no fragment reads, 4 no MFMAs,
// 8 every epilogue operand from the same table line / state row (L2 hits instead of scattered lines)
#ifndef P2_ABL
#define P2_ABL 0
#endif

namespace {

constexpr int P2_BM = 128, P2_BN = 256;                   // hypotheses x gate columns of a tile
constexpr int P2_NK = 16;                                 // k-steps (H = 512)
constexpr int P2_WS = 3, P2_HS = 4;                       // ring slots: gate matrix, state rows
constexpr int P2_WSTAGE = P2_BN * 32, P2_HSTAGE = P2_BM * 32;       // floats per slot
constexpr int P2_LDS_BYTES = (P2_WS * P2_WSTAGE + P2_HS * P2_HSTAGE) * 4;
static_assert(P2_LDS_BYTES == 160 * 1024, "both rings fill the CU's LDS");
constexpr int P2_NPW = 8, P2_NPH = 4;                     // LDS-DMA pieces (8 rows x 128 B) per k-step: of a gate wave (0-3) / a state wave (4-7)
constexpr int P2_NBLK = 4;                                // accumulator blocks of a wave: (gate block j, hypothesis block i) = block 2 j + i
constexpr int P2_OPS = 5;                                 // epilogue operand loads per block: four table quads + the old cell state

__device__ float gate_p2_dump_page[4 * 512 + 8];          // where the stores of hypothesis rows past the edge go

// ---- issue order of a tile's vector-memory operations, per wave type, and the counted waits that follow from it ----
// The two waves of a SIMD (w, w + 4) run the same matrix instructions in lock step (one barrier per k-step); an LDS-DMA instruction costs
// its wave 25-56 issue cycles during which only the OTHER wave can feed the matrix pipe, so they must not request at the same time:
//   gate waves  (0-3, LATE = false): k-step j issues  D  8 gate-matrix pieces of stage j + 2 behind the first matrix instructions of its FIRST
//               half step, then  X  its extra loads;
//   state waves (4-7, LATE = true):  k-step j issues  X  behind its first half step, then  D  4 state pieces of stage j + 3 (of the NEXT tile
//               from j = 13 on) behind the first matrix instructions of its SECOND half step.
//   X:  j = 0 row ids of the epilogue (2) . j = 1 [state waves] row ids of the next tile's pieces (4) . j = 2 prev / word of the epilogue rows (4)
//       . j = 4 [state waves] prev of the next tile's piece rows (4) . j = p2_kx(b) the epilogue operands of block b (5): k-steps 4, 7, 10, 13
//       (state waves: 5, 8, 11, 14).  An X of k-step j has to be back on top of k-step j + 2 -- the counter retires in order and the stage
//       waited for there was requested behind it -- and the operands are scattered 64-byte pieces of a table of hundreds of MB: all four
//       blocks in four consecutive k-steps cost 14 % of the launch (profiles/r06_v_gate_p2_*.txt); spread out they are a quarter of the burst.
// the cell update behind k-step 15 issues ST stores per block.
constexpr int p2_np(bool late) { return late ? P2_NPH : P2_NPW; }
constexpr int p2_kx(bool late, int b) { return 4 + 3 * b + (late ? 1 : 0); }
constexpr int p2_x(bool late, int j) {
    int n = j == 0 ? 2 : j == 2 ? 4 : ((j == 1 || j == 4) && late) ? P2_NPH : 0;
    for (int b = 0; b < P2_NBLK; ++b) n += j == p2_kx(late, b) ? P2_OPS : 0;
    return n;
}
constexpr int p2_clamp(int n) { return n > 63 ? 63 : n; }        // (six bits; a smaller count only waits longer)
// operations issued behind X of k-step j0 up to the X position of k-step j1 (gate waves: D of j0 + 1 .. j1; state waves: D of j0 .. j1 - 1)
constexpr int p2_behind_x(bool late, int j0, int j1) {
    int n = (j1 - j0) * p2_np(late);
    for (int j = j0 + 1; j < j1; ++j) n += p2_x(late, j);
    return n;
}
// top of k-step kt: this wave's pieces of stage kt + 1 have landed.  Gate waves requested them in k-step kt - 1, in front of its X; state
// waves in k-step kt - 2: behind them X and D of k-step kt - 1.  k-step "-1" / "-2" is the tile before's 15 / 14, whose stores follow k-step 15.
constexpr int p2_top(bool late, int kt, int stores) {
    const int xprev = kt == 0 ? 0 : p2_x(late, kt - 1);
    return late ? P2_NPH + xprev + (kt <= 1 ? stores : 0) : xprev + (kt == 0 ? stores : 0);
}
// in front of the cell update of block b: younger than its operands are the X behind them, the D of the k-steps behind (gate waves: from the
// next k-step on, state waves: from the same) and the stores of the blocks before
constexpr int p2_cell(bool late, int b, int st) {
    const int kx = p2_kx(late, b);
    int n = p2_np(late) * (P2_NK - (late ? 0 : 1) - kx) + st * b;
    for (int j = kx + 1; j < P2_NK; ++j) n += p2_x(late, j);
    return n;
}
static_assert(p2_behind_x(false, 0, 2) == 16 && p2_behind_x(true, 0, 2) == 12 && p2_behind_x(true, 1, 4) == 16 && p2_behind_x(false, 2, 4) == 16 &&
              p2_top(false, 3, 12) == 4 && p2_top(true, 1, 12) == 18 && p2_top(true, 0, 0) == 4 && p2_top(false, 5, 0) == 5 &&
              p2_cell(true, 3, 3) == 17 && p2_cell(false, 0, 3) == 8 * 11 + 15, "issue-order bookkeeping");

template <bool HF32, bool LATE>
struct GateP2 {
    static constexpr int ST = HF32 ? 4 : 3;               // stores per block and lane (c, h hi, h lo[, h f32])
    static constexpr int OOB_ROW = 0x7fffffff;
    const GateXgArgs &a;
    float *smem;
    int lane, wave, gp, hp, li, hf, lrow, lslot, H, ld, M, Q, n0, tiles_m, tm;
    __amdgpu_buffer_rsrc_t rs_w, rs_h;
    float *dump;
    int goff[2][2];                                       // fragment offsets (floats) inside a slot: (half step, plane) of row li, swizzled
    int wslot;                                            // slot of the gate stage of the CURRENT k-step (uniform, run time)
    int eg[2], ep[2], ew[2];                              // epilogue: global row, predecessor, word of the lane's hypothesis in its two blocks
    bool eok[2];
    static constexpr int NPC = LATE ? P2_NPH : P2_NPW;    // this wave's LDS-DMA pieces per k-step
    int vidx[NPC], vslot_h[P2_NPH];                       // LDS-DMA pieces: 16-byte record index of the lane (row x records per row + swizzled slot)
    int w16, h16;
    int rn[P2_NPH];                                       // next tile: piece row ids, then their predecessors
    bool nok[P2_NPH];
    f32x16 acc[2][2];                                     // [gate block j][hypothesis block i]
    f16x8 A[2][2], B[2][2];                               // fragments of ONE half step: [block][plane 0 = hi, 1 = lo]
    f32x4 xg[P2_NBLK][4], cp[P2_NBLK];
#ifdef JLM_PROFILE
    int tix = 0;
#endif

    __device__ __forceinline__ GateP2(const GateXgArgs &a_, float *smem_) : a(a_), smem(smem_) {}

    __device__ __forceinline__ float *wring(int slot) const { return smem + slot * P2_WSTAGE; }
    __device__ __forceinline__ float *hring(int slot) const { return smem + P2_WS * P2_WSTAGE + slot * P2_HSTAGE; }
    __device__ __forceinline__ static int inc3(int s) { return s == 2 ? 0 : s + 1; }

    // piece i of this wave: gate-matrix rows 8 (8 wave + i) .. + 7 of the tile (gate waves) / state rows 8 (4 (wave - 4) + i) .. + 7 (state waves)
    template <int KT>
    __device__ __forceinline__ void issue_piece(int i) {                     // k-step KT's requests: gate stage KT + 2 / state stage KT + 3
        if constexpr (!LATE)
            __builtin_amdgcn_struct_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(wring(inc3(inc3(wslot))) + 8 * (P2_NPW * wave + i) * 32),
                                                        16, vidx[i], 0, ((KT + 2) & (P2_NK - 1)) * 128, 0, 0);
        else
            __builtin_amdgcn_struct_ptr_buffer_load_lds(rs_h, (__attribute__((address_space(3))) void *)(hring((KT + 3) & (P2_HS - 1)) + 8 * (P2_NPH * (wave - 4) + i) * 32),
                                                        16, vidx[i], 0, ((KT + 3) & (P2_NK - 1)) * 128, 0, 0);
    }
    __device__ __forceinline__ void rdA(const float *wbase, int st, int j, int p) {
        if ((P2_ABL & 2)) { asm volatile("" : "+v"(A[j][p])); return; }
        A[j][p] = *reinterpret_cast<const f16x8 *>(wbase + (2 * gp + j) * 1024 + goff[st][p]);
    }
    __device__ __forceinline__ void rdB(const float *hbase, int st, int i, int p) {
        if ((P2_ABL & 2)) { asm volatile("" : "+v"(B[i][p])); return; }
        B[i][p] = *reinterpret_cast<const f16x8 *>(hbase + (2 * hp + i) * 1024 + goff[st][p]);
    }
    __device__ __forceinline__ void load_int(int &d, const int *src) { asm volatile("global_load_dword %0, %1, off" : "=&v"(d) : "v"(src) : "memory"); }
    template <int N>
    __device__ __forceinline__ void wait_int(int &v) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N) : "memory"); }
    // epilogue operands of block b = (gate block j = b >> 1, hypothesis block i = b & 1)
    __device__ __forceinline__ void load_ops(int b) {
        const int j = b >> 1, i = b & 1;
        const float *xrp = a.xg + (size_t)((P2_ABL & 8) ? 0 : ew[i]) * (size_t)(4 * H) + n0 + 32 * (2 * gp + j) + 4 * hf;
        const float *cpp = a.c_in + (size_t)((P2_ABL & 8) ? 0 : ep[i] >= 0 ? ep[i] : 0) * ld + (n0 >> 2) + 8 * (2 * gp + j) + 4 * hf;
        asm volatile("global_load_dwordx4 %0, %5, off\n\t"
                     "global_load_dwordx4 %1, %5, off offset:32\n\t"
                     "global_load_dwordx4 %2, %5, off offset:64\n\t"
                     "global_load_dwordx4 %3, %5, off offset:96\n\t"
                     "global_load_dwordx4 %4, %6, off"
                     : "=&v"(xg[b][0]), "=&v"(xg[b][1]), "=&v"(xg[b][2]), "=&v"(xg[b][3]), "=&v"(cp[b])
                     : "v"(xrp), "v"(cpp) : "memory");
    }
    template <int N>
    __device__ __forceinline__ void wait_ops(int b) {
        asm volatile("s_waitcnt vmcnt(%5)" : "+v"(xg[b][0]), "+v"(xg[b][1]), "+v"(xg[b][2]), "+v"(xg[b][3]), "+v"(cp[b]) : "n"(N) : "memory");
    }

    __device__ __forceinline__ void mf(int j, int pa, int i, int pb, bool zero) {
        const f32x16 zf = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (P2_ABL & 4) asm volatile("" : "+v"(acc[j][i]) : "v"(A[j][pa]), "v"(B[i][pb]));
        else acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[j][pa], B[i][pb], zero ? zf : acc[j][i], 0, 0, 0);
    }

    // One half step: 12 matrix instructions over the wave's 2 x 2 blocks, the fragment registers refilled in place with the NEXT half
    // step's (wn / hn: the slots it lives in, nst: its half) behind the instruction that read them last:
    //    1  A_lo0 . B_hi0     2  A_lo0 . B_hi1  -> A_lo0      3  A_lo1 . B_hi0     4  A_lo1 . B_hi1  -> A_lo1
    //    5  A_hi0 . B_hi0     6  A_hi1 . B_hi0  -> B_hi0      7  A_hi0 . B_hi1     8  A_hi1 . B_hi1  -> B_hi1
    //    9  A_hi0 . B_lo0    10  A_hi0 . B_lo1  -> A_hi0     11  A_hi1 . B_lo0 -> B_lo0    12  A_hi1 . B_lo1  -> A_hi1, B_lo1
    // (no two consecutive instructions share an accumulator; the next half step uses B_hi0 first, 7 instructions behind its refill).
    // DMA: the k-step's six LDS-DMA instructions go out one behind each of the first six matrix instructions (first half step only).
    // The order is pinned by a scheduling barrier behind every matrix instruction and what follows it (left to sched_group_barrier masks
    // hipcc picks the matrix instructions of a group in an order of its own and bunches the reads at the end of the half step).
    template <int DMA_KT, bool ZERO, bool REFILL>
    __device__ __forceinline__ void half(const float *wn, const float *hn, int nst) {
        auto dma = [&](int i) { if constexpr (DMA_KT > -8 && !(P2_ABL & 1)) { if (i < NPC) this->template issue_piece<(DMA_KT > -8 ? DMA_KT : 0)>(i); } };
        auto sb = [] { __builtin_amdgcn_sched_barrier(0); };
        mf(0, 1, 0, 0, ZERO); dma(0); sb();
        mf(0, 1, 1, 0, ZERO); if (REFILL) rdA(wn, nst, 0, 1); dma(1); sb();
        mf(1, 1, 0, 0, ZERO); dma(2); sb();
        mf(1, 1, 1, 0, ZERO); if (REFILL) rdA(wn, nst, 1, 1); dma(3); sb();
        mf(0, 0, 0, 0, false); dma(4); sb();
        mf(1, 0, 0, 0, false); if (REFILL) rdB(hn, nst, 0, 0); dma(5); sb();
        mf(0, 0, 1, 0, false); dma(6); sb();
        mf(1, 0, 1, 0, false); if (REFILL) rdB(hn, nst, 1, 0); dma(7); sb();
        mf(0, 0, 0, 1, false); sb();
        mf(0, 0, 1, 1, false); if (REFILL) rdA(wn, nst, 0, 0); sb();
        mf(1, 0, 0, 1, false); if (REFILL) rdB(hn, nst, 0, 1); sb();
        mf(1, 0, 1, 1, false); if (REFILL) { rdA(wn, nst, 1, 0); rdB(hn, nst, 1, 1); } sb();
    }

    template <int KT>
    __device__ __forceinline__ void kstep(int m0, int m0n, bool has_next, bool first) {
        // ---- stage KT + 1 of both rings has landed (every wave waits for its own pieces), every wave is through k-step KT - 1
        if constexpr (KT <= 1) {
            if (first) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(p2_clamp(p2_top(LATE, KT, 0))) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(p2_clamp(p2_top(LATE, KT, P2_NBLK * ST))) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(p2_clamp(p2_top(LATE, KT, 0))) : "memory");
        }
        const int wcur = wslot, wnext = inc3(wslot);
        constexpr int hcur = KT & (P2_HS - 1), hnext = (KT + 1) & (P2_HS - 1);
        if constexpr (KT == 0) {
            if (first) {
                // the very first fragments (stage 0 landed: it is older than stage 1)
#pragma unroll
                for (int j = 0; j < 2; ++j) { rdA(wring(wcur), 0, j, 1); rdA(wring(wcur), 0, j, 0); rdB(hring(hcur), 0, j, 0); rdB(hring(hcur), 0, j, 1); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KT == 13 && LATE) {
            // in front of the next tile's first state stage: its pieces' predecessors (X of k-step 4) -> record indices
#pragma unroll
            for (int i = 0; i < P2_NPH; ++i) wait_int<p2_clamp(p2_behind_x(true, 4, 13))>(rn[i]);
#pragma unroll
            for (int i = 0; i < P2_NPH; ++i) vidx[i] = (nok[i] && rn[i] >= 0) ? rn[i] * h16 + vslot_h[i] : OOB_ROW;
        }
        half<(LATE ? -8 : KT), KT == 0, true>(wring(wcur), hring(hcur), 1);
        // ---- the k-step's extra loads (X)
        if constexpr (KT == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = m0 + 32 * (2 * hp + i) + li;
                eok[i] = r < M;
                load_int(eg[i], a.rows + (eok[i] ? r : M - 1));
            }
        }
        if constexpr (KT == 1 && LATE) {
#pragma unroll
            for (int i = 0; i < P2_NPH; ++i) {
                const int r = m0n + 8 * (P2_NPH * (wave - 4) + i) + lrow;
                nok[i] = has_next && r < M;
                load_int(rn[i], a.rows + (nok[i] ? r : M - 1));
            }
        }
        if constexpr (KT == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) wait_int<p2_clamp(p2_behind_x(LATE, 0, 2))>(eg[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) { load_int(ep[i], a.prev + eg[i]); load_int(ew[i], a.word + eg[i]); }
        }
        if constexpr (KT == 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { wait_int<p2_clamp(p2_behind_x(LATE, 2, 4))>(ep[i]); wait_int<p2_clamp(p2_behind_x(LATE, 2, 4))>(ew[i]); }
        }
        if constexpr (KT == 4 && LATE) {
#pragma unroll
            for (int i = 0; i < P2_NPH; ++i) wait_int<p2_clamp(p2_behind_x(true, 1, 4))>(rn[i]);
#pragma unroll
            for (int i = 0; i < P2_NPH; ++i) load_int(rn[i], a.prev + rn[i]);
        }
        // the epilogue operands of block b at k-step p2_kx(b)  (P2_ABL & 8: the same loads from ONE table line / state row -- L2 hits)
        if constexpr (KT >= 4 && (KT - 4 - (LATE ? 1 : 0)) % 3 == 0 && (KT - 4 - (LATE ? 1 : 0)) / 3 < P2_NBLK) load_ops((KT - 4 - (LATE ? 1 : 0)) / 3);
        __builtin_amdgcn_sched_barrier(0);
        // second half: refills come from stage KT + 1 (KT = 15: stage 0 of the next tile)
        half<(LATE ? KT : -8), false, true>(wring(wnext), hring(hnext), 0);
        wslot = wnext;
    }

    // cell update of block b in registers: acc[4 gate + e] + table = pre-activation (x 1 / descale) of gate `gate`, unit u0 + e, hypothesis li
    template <int BK>
    __device__ __forceinline__ void cell() {
        constexpr int j = BK >> 1, i = BK & 1;
        wait_ops<p2_clamp(p2_cell(LATE, BK, ST))>(BK);
        if constexpr (BK == 0) JLM_P2_T(2);
        const float ks = a.descale * -1.4426950408889634f, kt = a.descale * -2.8853900817779268f;      // (jlm_common.h: exact for ds = 2^-S)
        const int g = eok[i] ? eg[i] : -1;
        const int u0 = (n0 >> 2) + 8 * (2 * gp + j) + 4 * hf;
        f32x4 cn, hn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gi = jlm_sigmoid_k(acc[j][i][e] + xg[BK][0][e], ks), gf = jlm_sigmoid_k(acc[j][i][4 + e] + xg[BK][1][e], ks);
            const float go = jlm_sigmoid_k(acc[j][i][8 + e] + xg[BK][2][e], ks), gg = jlm_tanh_k(acc[j][i][12 + e] + xg[BK][3][e], kt);
            cn[e] = (ep[i] >= 0 ? cp[BK][e] : 0.0f) * gf + gg * gi;
            hn[e] = jlm_tanh(cn[e]) * go;
        }
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x4 hi4, lo4;
        jlm_split4(hn, a.h_scale, hi4, lo4);
        float *crow = g >= 0 ? a.c_out + (size_t)g * ld + u0 : dump;
        *reinterpret_cast<f32x4 *>(crow) = cn;
        // units u0 .. u0 + 3 = one half of an 8-value block [8 x f16 hi][8 x f16 lo] of the split row
        _Float16 *blk = g >= 0 ? reinterpret_cast<_Float16 *>(a.h_out + (size_t)g * ld + (u0 & ~7)) + (u0 & 7) : reinterpret_cast<_Float16 *>(dump);
        *reinterpret_cast<f16x4 *>(blk) = hi4;
        *reinterpret_cast<f16x4 *>(g >= 0 ? blk + 8 : blk + 4) = lo4;
        if constexpr (HF32) *reinterpret_cast<f32x4 *>(g >= 0 ? a.h_f32 + (size_t)g * ld + u0 : dump) = hn;
    }

    __device__ __forceinline__ void tile(bool has_next, bool first) {
        const int m0 = tm * P2_BM, m0n = (tm + Q) * P2_BM;
        JLM_P2_T(0);
        gate_for_each_ic([&](auto ktc) { this->template kstep<decltype(ktc)::value>(m0, m0n, has_next, first); }, std::make_integer_sequence<int, P2_NK>{});
        JLM_P2_T(1);
        gate_for_each_ic([&](auto bc) { this->template cell<decltype(bc)::value>(); }, std::make_integer_sequence<int, P2_NBLK>{});
        JLM_P2_T(3);
#ifdef JLM_PROFILE
        ++tix;
#endif
    }

    __device__ __forceinline__ void run() {
        lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        gp = wave & 3; hp = wave >> 2;
        li = lane & 31; hf = lane >> 5; lrow = lane >> 3; lslot = lane & 7;
        H = a.H; ld = a.ld;
        // block b runs on XCD b % 8 (observed; speed only): XCD x owns gate-column tile x -- its 512 KB of the gate matrix stay in that L2 --
        // and its 32 workgroups walk the row tiles q, q + 32, ...
        const int b = blockIdx.x;
        const int tn = b & 7, q = b >> 3;
        Q = (int)gridDim.x >> 3;
        M = a.ndev ? min(*a.ndev, a.nrows) : a.nrows;
        tiles_m = (M + P2_BM - 1) / P2_BM;
        if (q >= tiles_m) return;
        n0 = tn * P2_BN;
        rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wt), (short)16, 0x7ffffff0, 0x00020000);     // 16-byte records, one address
        rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.h), (short)16, 0x7ffffff0, 0x00020000);      // register (csrc/jlm_gate.hip)
        w16 = H >> 2; h16 = ld >> 2;
        dump = gate_p2_dump_page + 4 * (int)threadIdx.x;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int p = 0; p < 2; ++p) goff[st][p] = li * 32 + (((4 * st + 2 * hf + p) ^ ((li >> 1) & 7)) * 4);
        tm = q;
        if constexpr (!LATE) {
#pragma unroll
            for (int i = 0; i < P2_NPW; ++i) {
                const int row = 8 * (P2_NPW * wave + i) + lrow;
                vidx[i] = (n0 + row) * w16 + (lslot ^ ((row >> 1) & 7));
                asm volatile("" : "+v"(vidx[i]));
            }
            // W0 W1: both there on top of k-step 0 (p2_top(false, 0, 0) = 0)
            wslot = 1;                                    // (the gate pieces of "k-step -2" go into slot wslot + 2 = 0, ...)
#pragma unroll
            for (int i = 0; i < P2_NPW; ++i) issue_piece<-2>(i);
            wslot = 2;
#pragma unroll
            for (int i = 0; i < P2_NPW; ++i) issue_piece<-1>(i);
        } else {
            // the first tile's state pieces by ordinary loads (nothing asynchronous of this wave is in flight yet)
            const int m0 = tm * P2_BM;
#pragma unroll
            for (int i = 0; i < P2_NPH; ++i) {
                const int row = 8 * (P2_NPH * (wave - 4) + i) + lrow;
                vslot_h[i] = lslot ^ ((row >> 1) & 7);
                const int r = m0 + row;
                const bool ok = r < M;
                const int p = a.prev[a.rows[ok ? r : M - 1]];
                vidx[i] = (ok && p >= 0) ? p * h16 + vslot_h[i] : OOB_ROW;
            }
#pragma unroll
            for (int i = 0; i < P2_NPH; ++i) asm volatile("" : "+v"(vidx[i]));
            // H0 H1 H2: on top of k-step 0 everything but H2 has to be there (p2_top(true, 0, 0) = 4)
#pragma unroll
            for (int i = 0; i < P2_NPH; ++i) issue_piece<-3>(i);
#pragma unroll
            for (int i = 0; i < P2_NPH; ++i) issue_piece<-2>(i);
#pragma unroll
            for (int i = 0; i < P2_NPH; ++i) issue_piece<-1>(i);
        }
        wslot = 0;
        tile(tm + Q < tiles_m, true);
        tm += Q;
        for (; tm < tiles_m; tm += Q) tile(tm + Q < tiles_m, false);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the last tile's look-ahead pieces)
    }
};

template <bool HF32>
__global__ __launch_bounds__(512, 1) void gate_p2_kernel(GateXgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (wave < 4) { GateP2<HF32, false> k(a, smem); k.run(); }
    else { GateP2<HF32, true> k(a, smem); k.run(); }
}

}  // namespace

int jlm_gate::p2_launch(const GateXgArgs &a, hipStream_t stream) {
    // (with the plain f32 copy of h' -- untied models -- the state waves spill four registers, and scratch traffic would break the counted
    //  waits: those launches stay with the other kernels)
    if (a.h_f32 != nullptr) return -2;
    const void *fn = reinterpret_cast<const void *>(gate_p2_kernel<false>);
    static JlmLdsGrant grant[1];
    if (int rc = jlm_grant_lds(grant[0], fn, P2_LDS_BYTES)) return rc;
    GateXgArgs args = a;
    void *params[] = {&args};
    const int tiles_m = (a.nrows + P2_BM - 1) / P2_BM;
    const int Q = tiles_m < 32 ? tiles_m : 32;             // row-tile sequences per gate-column tile: one resident workgroup per CU
    hipError_t e = hipLaunchKernel(fn, dim3(8 * Q), dim3(512), params, P2_LDS_BYTES, stream);
    return e == hipSuccess ? 0 : (int)e;
}
