// jlm_gate_ws.hip -- the fused LSTM step of the decode, W-stationary persistent form (round 4; what jlm_lstm_step_xg launches at
// H = 512 from 16 384 rows on: csrc/jlm_gate.hip picks by the launch's row bound).  Reference: decoder/model.py:125-139 with the state gather of decoder/decoder.py:206-218.
//
// Its own translation unit because it is compiled with  -mllvm -amdgpu-mfma-vgpr-form : the wave's 256 accumulation registers hold
// its slice of the gate matrix for the whole kernel, so the MFMA accumulators must live in the architectural VGPRs -- left to its
// heuristics hipcc puts them in a[0:79], copies pieces of the gate matrix out of the way (v_accvgpr_read / write around loads that
// are still in flight) and spills them to scratch.
#include "jlm_gate.h"

// Per-workgroup timeline (-DJLM_PROFILE builds only, tools/probes/gate_ws_profile.py): wave 0 stamps the 100 MHz wall clock at
// kernel start [0], after the index loads [1], when the first stage has landed [2], and per tile t at the end of its k-steps
// [3 + 2 t] and of its cell update [4 + 2 t]; [30], [31]: shader clock at [2] and [3].
#ifdef JLM_PROFILE
static __device__ unsigned long long jlm_gate_ws_time[256][32];
#define JLM_WS_T(i) do { if (threadIdx.x == 0 && (i) < 30) { jlm_gate_ws_time[blockIdx.x & 255][i] = wall_clock64(); \
    if ((i) == 2 || (i) == 3) jlm_gate_ws_time[blockIdx.x & 255][(i) == 2 ? 30 : 31] = clock64(); } } while (0)
extern "C" int jlm_prof_read_gate_ws(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(jlm_gate_ws_time), sizeof(jlm_gate_ws_time)) == hipSuccess ? 0 : -1;
}
#else
#define JLM_WS_T(i) (void)0
#endif

namespace {

__device__ float gate_dump_page[4 * 256 + 8];              // where the stores of hypothesis rows past the edge go

// Round 4: the W-STATIONARY, PERSISTENT form of the same step (gate_ws_kernel; H = 512).
//
// What the per-workgroup timeline of gate_xg_u16_kernel says (profiles/r03_a_gate_timeline.txt): of the 21.5 us a 160 x 128 tile
// costs at ANY number of rows, the sixteen k-steps take 14.2 (1 311 cycles each against the 960 their MFMAs need: two waves per
// SIMD trading the matrix pipe, 36 KB of LDS-DMA and 8 x 17 fragment reads per k-step, one 8-wave barrier) and 7 lie outside the
// loop -- index chains, first ring stages, epilogue, launch gaps -- with nothing overlapping them; and every one of the 16 row
// tiles that share a gate-column tile stages the same 256 KB slice of the gate matrix through its LDS again.  Here:
//   * a workgroup is FOUR waves, one per SIMD, each owning one 32-column gate block of the tile for the whole kernel: its slice
//     of the gate matrix (32 columns x 512 k x two f16 planes = 64 KB) is loaded ONCE, straight from L2 into the wave's 256
//     accumulation registers (global_load_dwordx4 a[..]: AGPRs are legal MFMA A operands on gfx950), and stays there while the
//     workgroup walks its row tiles (grid = 256 workgroups = 16 column tiles x 16 row-tile sequences: 8 tiles each at 20 480 rows);
//   * only the hypothesis rows go through LDS: 20 KB per 32-value k-step instead of 36, five 1-KB pieces per wave, an 8-slot
//     ring (all 160 KB; 16 k-steps = two laps, so slot numbers are compile-time constants) filled seven k-steps ahead;
//   * every wave multiplies ALL five hypothesis blocks of the tile by its gate block: 15 MFMAs per half k-step and wave with
//     10 fragment reads (the gate fragments are registers), refilled in place behind the MFMA that read them last;
//   * the ring does not stop at a tile's end: the first seven stages of the NEXT tile are requested under the last seven
//     k-steps of this one (its row indices under the first ones), so a tile's epilogue is followed by the next tile's first MFMA;
//   * the epilogue takes its operands (one 128-byte table line and the old cell state per hypothesis) block by block, two
//     blocks in flight, the first two requested under the last two k-steps; stores go through buffer descriptors (rows past
//     the edge are dropped by the range check, not branched around) so that the number of vector-memory operations in flight
//     is known at every wait: every wait below is a COUNTED s_waitcnt vmcnt(n), n derived at compile time from the issue
//     order (ws_* functions), never a drain.
// In the first tile the gate-matrix loads ride in the same in-order queue, four per k-step, seven k-steps ahead like the ring.
constexpr int WS_NB = 5;                                  // hypothesis blocks of a tile, all of them on every wave
constexpr int WS_NP = 5;                                  // LDS-DMA pieces (8 rows x 128 B) per wave and stage: 20 pieces = 160 rows
constexpr int WS_STAGE_FLOATS = GT_BM * 32;               // one 32-value k-step of 160 hypothesis rows: 20 KB
constexpr int WS_NK = 16;                                 // k-steps (H = 512)
#ifndef WS_ABL
#define WS_ABL 0      // measurement builds (results wrong): 1 no LDS-DMA in the k-steps, 2 no fragment reads in the k-steps, 4 no MFMAs, 8 no barrier
#endif
constexpr int WS_ST = 3;                                  // stores per hypothesis block and lane that every launch issues (c, h hi, h lo)
// Round 6: the cell update took 5.3-5.8 us of an 18-us tile with ~730 VALU instructions in it (profiles/r04_a_gate_ws_ablate.txt): the
// operands of blocks 2..4 (table line, old cell state) are requested one block ahead, two in flight -- three exposed round trips to HBM /
// the fabric.  The registers for five blocks in flight are not there (W 256 + 250 of 256), so the LINES are warmed instead: at k-step
// WS_PF every lane touches one dword of each block's table line and cell-state line (10 loads into one dummy register, kept alive up to
// the first counted wait behind them); the epilogue's own loads then hit the XCD's L2.  -DWS_PF=-1: off (A/B builds).
#ifndef WS_PF
#define WS_PF -1
#endif
constexpr int WS_NPF = 2 * WS_NB;                         // the warming loads of a tile
// Round 6, the epilogue's operands FOUR blocks in flight instead of two (-DWS_OPS4=0: the round-4 form).  With every operand an L2 hit the
// launch takes 123 instead of 149 us at 20 480 rows (profiles/r06_w_gate_ws_ops.txt): blocks 2..4 were requested one block ahead, a round
// trip exposed per block.  The ten registers of two more operand sets are the wave's B fragments: the last half step of a tile no longer
// refills them with the next tile's first fragments (those are read on top of its first k-step, behind the barrier, as in the first
// tile), so behind the last matrix instruction they take the operands of blocks 2 and 3; block 4 goes into block 0's set behind its use.
// WS_OPS4 = 1: sets 0 / 1 under k-steps 14 / 15, 2 and 3 behind the last k-step; 2: set 2 under k-step 15 as well; 3: sets 0 / 1 / 2 under k-steps
// 13 / 14 / 15, 3 behind the last (the compiler finds the registers: no spills in any form -- scratch traffic would break the counted waits).
#ifndef WS_OPS4
#define WS_OPS4 1
#endif
constexpr int ws_kx(int b) {                              // k-step under which the operands of block b < 4 are requested (16: behind the k-steps)
    return WS_OPS4 == 3 ? (b < 3 ? 13 + b : 16) : WS_OPS4 == 2 ? (b == 0 ? 14 : b < 3 ? 15 : 16) : WS_OPS4 == 1 ? (b < 2 ? 14 + b : 16) : (b < 2 ? 14 + b : -1);
}
constexpr int ws_nops(int j) { int n = 0; for (int b = 0; b < 4; ++b) n += ws_kx(b) == j ? 5 : 0; return n; }

// ---- the issue order of a tile's vector-memory operations, and the counted waits that follow from it ----
// k-step j issues, in this order:  D  the 5 pieces of stage j + L (of the next tile from j = 16 - L on);
//                                  X  the tile's extra loads: j = 0 row ids of the epilogue (5; first tile: in the prologue),
//                                     j = 1 row ids of the next tile's pieces (5), j = 3 prev / word of the epilogue rows (10; first
//                                     tile: prologue), j = 4 prev of the next tile's piece rows (5), j = WS_PF the warming loads (10),
//                                     j = 14 / 15 the epilogue operands of blocks 0 / 1 (5 each);
//                                  W  first tile only: the 4 gate-matrix loads of k-step j + L + 1 (while there is one).
constexpr int ws_x(bool first, int j) {
    return (j == 0 ? (first ? 0 : 5) : j == 1 ? 5 : j == 3 ? (first ? 0 : 10) : j == 4 ? 5 : 0) + ws_nops(j) + (j == WS_PF ? WS_NPF : 0);
}
static_assert(WS_PF < 0 || (WS_PF > 4 && WS_PF < 14), "the warming loads need prev / word of the epilogue rows (X of k-step 3)");
constexpr int ws_w(bool first, int L, int j) { return (first && j + L + 1 < WS_NK) ? 4 : 0; }
constexpr int ws_issued(bool first, int L, int j) { return WS_NP + ws_x(first, j) + ws_w(first, L, j); }
constexpr int ws_sum(bool first, int L, int j0, int j1) {      // operations issued by k-steps j0 .. j1 - 1
    int n = 0;
    for (int j = j0; j < j1; ++j) n += ws_issued(first, L, j);
    return n;
}
constexpr int ws_clamp(int n) { return n > 63 ? 63 : n; }       // (the counter has six bits; a smaller count only waits longer)
// top of k-step kt: stage kt + 1 has landed.  Prologue of the first tile: W0 D0 W1 D1 .. W(L-1) D(L-1) W(L).
constexpr int ws_top(bool first, int L, int kt) {
    if (kt + 1 < L) return first ? ws_clamp(9 * (L - 2 - kt) + 4 + ws_sum(first, L, 0, kt)) : 63;     // (later tiles: requested under the
    const int j = kt + 1 - L;                                                                       //  tile before, landed behind its epilogue)
    return ws_clamp(ws_x(first, j) + ws_w(first, L, j) + ws_sum(first, L, j + 1, kt));
}
// behind D of k-step 3: the epilogue's row ids (X of k-step 0) are there
constexpr int ws_wait_eg(int L) { return ws_clamp(ws_w(false, L, 0) + ws_sum(false, L, 1, 3) + WS_NP); }
// behind D of k-step 4: the next tile's piece row ids (X of k-step 1)
constexpr int ws_wait_rn(bool first, int L) { return ws_clamp(ws_w(first, L, 1) + ws_sum(first, L, 2, 4) + WS_NP); }
// in front of D of k-step 16 - L: prev of the next tile's piece rows (X of k-step 4)
constexpr int ws_wait_ppn(bool first, int L) { return ws_clamp(ws_w(first, L, 4) + ws_sum(first, L, 5, WS_NK - L)); }
// behind D of k-step WS_PF (14 without the warming loads): prev / word of the epilogue rows (X of k-step 3; first tile: prologue)
constexpr int WS_EPEW_AT = WS_PF >= 0 ? WS_PF : ws_kx(0);
constexpr int ws_wait_epew(int L) { return ws_clamp(ws_w(false, L, 3) + ws_sum(false, L, 4, WS_EPEW_AT) + WS_NP); }
static_assert(ws_top(false, 7, 8) == 40 && ws_top(false, 7, 3) == 63 && ws_top(true, 7, 0) == 49 && ws_wait_eg(7) == 20, "issue-order bookkeeping");

// (a struct with member templates, not lambdas: clang rejects inline-asm operands that name captured locals inside a GENERIC lambda)
template <int L, bool HF32>
struct GateWs {
    static constexpr int S = L + 1;                       // ring slots
    static_assert(S == 8 || S == 4, "16 k-steps must be whole laps of the ring");
    static constexpr int OOB_ROW = 0x7fffffff;
    const GateXgArgs &a;
    float *smem;
    int lane, wave, li, hf, lrow, lslot, H, ld, M, Q, n0, u0, tiles_m, tm;
    __amdgpu_buffer_rsrc_t rs_h;
    float *dump;
    const float *wrow;
    f16x8 W[2 * WS_NK][2];                                // the wave's gate block: MFMA A operand of k16-step s (plane 0 = hi, 1 = lo)
    int goff[2][2];                                       // fragment offsets (floats) inside a stage
    int eg[WS_NB], ep[WS_NB], ew[WS_NB];                  // epilogue: global row, predecessor row, word of the lane's hypothesis in each block
    bool eok[WS_NB];
    int vidx[WS_NP], vslot[WS_NP];                        // LDS-DMA pieces of this wave: 16-byte record index (row x records per row + swizzled slot, or OOB), the slot
    int h16;
    int rn[WS_NP];                                        // next tile: piece row ids, then their predecessors
    bool nok[WS_NP];
    f32x16 acc[WS_NB];
    f16x8 B[WS_NB][2];                                    // fragments of ONE half step, refilled in place (plane 0 hi, 1 lo)
    f32x4 xg[2][4], cp[2];                                // epilogue operands, two blocks in flight
    int warm;                                             // destination of the warming loads (never read)

    __device__ __forceinline__ GateWs(const GateXgArgs &a_, float *smem_) : a(a_), smem(smem_) {}

    __device__ __forceinline__ int piece_row(int m0, int i) const { return m0 + 8 * (WS_NP * wave + i) + lrow; }

    template <int KT>
    __device__ __forceinline__ void load_w() {            // the four loads of k-step KT (two k16-steps x two planes)
        asm volatile("global_load_dwordx4 %0, %4, off offset:%5\n\t"
                     "global_load_dwordx4 %1, %4, off offset:%6\n\t"
                     "global_load_dwordx4 %2, %4, off offset:%7\n\t"
                     "global_load_dwordx4 %3, %4, off offset:%8"
                     : "=&a"(W[2 * KT][0]), "=&a"(W[2 * KT][1]), "=&a"(W[2 * KT + 1][0]), "=&a"(W[2 * KT + 1][1])
                     : "v"(wrow), "n"(KT * 128), "n"(KT * 128 + 16), "n"(KT * 128 + 64), "n"(KT * 128 + 80) : "memory");
    }
    template <int SG>
    __device__ __forceinline__ void issue() {             // stage SG (mod 16) of whatever tile vidx describes
        float *base = smem + (SG & (S - 1)) * WS_STAGE_FLOATS;
#pragma unroll
        for (int i = 0; i < WS_NP; ++i)
            __builtin_amdgcn_struct_ptr_buffer_load_lds(rs_h, (__attribute__((address_space(3))) void *)(base + 8 * (WS_NP * wave + i) * 32),
                                                        16, vidx[i], 0, (SG & (WS_NK - 1)) * 128, 0, 0);
    }
    __device__ __forceinline__ void rdB(int stage, int st, int nb, int p) {
        if ((WS_ABL & 2) && !(stage == 0 && st == 0)) { asm volatile("" : "+v"(B[nb][p])); return; }
        B[nb][p] = *reinterpret_cast<const f16x8 *>(smem + (stage & (S - 1)) * WS_STAGE_FLOATS + nb * 1024 + goff[st][p]);
    }
    // one 128-byte table line (4 gates x the lane's 4 units) + the old cell state of block nb -> slot
    __device__ __forceinline__ void load_ops(int nb, int slot) {
        const float *xrp = a.xg + (size_t)((WS_ABL & 16) ? 0 : ew[nb]) * (size_t)(4 * H) + n0 + 32 * wave + 4 * hf;      // (WS_ABL & 16: every operand
        const float *cpp = a.c_in + (size_t)((WS_ABL & 16) ? 0 : ep[nb] >= 0 ? ep[nb] : 0) * ld + u0;                     //  from one line: L2 hits)
        asm volatile("global_load_dwordx4 %0, %5, off\n\t"
                     "global_load_dwordx4 %1, %5, off offset:32\n\t"
                     "global_load_dwordx4 %2, %5, off offset:64\n\t"
                     "global_load_dwordx4 %3, %5, off offset:96\n\t"
                     "global_load_dwordx4 %4, %6, off"
                     : "=&v"(xg[slot][0]), "=&v"(xg[slot][1]), "=&v"(xg[slot][2]), "=&v"(xg[slot][3]), "=&v"(cp[slot])
                     : "v"(xrp), "v"(cpp) : "memory");
    }
    // the same into the fragment registers: set 2 = B[0][0], B[0][1], B[1][0], B[1][1], B[2][0]; set 3 = B[2][1], B[3][0], B[3][1], B[4][0], B[4][1]
    template <int SET>
    __device__ __forceinline__ void load_ops_b(int nb) {
        const float *xrp = a.xg + (size_t)((WS_ABL & 16) ? 0 : ew[nb]) * (size_t)(4 * H) + n0 + 32 * wave + 4 * hf;
        const float *cpp = a.c_in + (size_t)((WS_ABL & 16) ? 0 : ep[nb] >= 0 ? ep[nb] : 0) * ld + u0;
        constexpr int o = SET == 2 ? 0 : 5;
        asm volatile("global_load_dwordx4 %0, %5, off\n\t"
                     "global_load_dwordx4 %1, %5, off offset:32\n\t"
                     "global_load_dwordx4 %2, %5, off offset:64\n\t"
                     "global_load_dwordx4 %3, %5, off offset:96\n\t"
                     "global_load_dwordx4 %4, %6, off"
                     : "=&v"(B[(o + 0) >> 1][(o + 0) & 1]), "=&v"(B[(o + 1) >> 1][(o + 1) & 1]), "=&v"(B[(o + 2) >> 1][(o + 2) & 1]),
                       "=&v"(B[(o + 3) >> 1][(o + 3) & 1]), "=&v"(B[(o + 4) >> 1][(o + 4) & 1])
                     : "v"(xrp), "v"(cpp) : "memory");
    }
    template <int SET, int N>
    __device__ __forceinline__ void wait_ops_b(f32x4 (&xq)[4], f32x4 &cq) {
        constexpr int o = SET == 2 ? 0 : 5;
        asm volatile("s_waitcnt vmcnt(%5)" : "+v"(B[(o + 0) >> 1][(o + 0) & 1]), "+v"(B[(o + 1) >> 1][(o + 1) & 1]), "+v"(B[(o + 2) >> 1][(o + 2) & 1]),
                                             "+v"(B[(o + 3) >> 1][(o + 3) & 1]), "+v"(B[(o + 4) >> 1][(o + 4) & 1]) : "n"(N) : "memory");
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) xq[g4] = __builtin_bit_cast(f32x4, B[(o + g4) >> 1][(o + g4) & 1]);
        cq = __builtin_bit_cast(f32x4, B[(o + 4) >> 1][(o + 4) & 1]);
    }
    template <int N>
    __device__ __forceinline__ void wait_ops(int slot) {
        asm volatile("s_waitcnt vmcnt(%5)" : "+v"(xg[slot][0]), "+v"(xg[slot][1]), "+v"(xg[slot][2]), "+v"(xg[slot][3]), "+v"(cp[slot]) : "n"(N) : "memory");
    }
    template <int N>
    __device__ __forceinline__ void wait_int(int &v) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N) : "memory"); }
    // a warming load: one dword of a line into `warm`, read-write so that all of them live in ONE register that stays allocated until
    // the counted wait of cell<0> (an output-only operand would be dead at once and its register handed out under the load in flight)
    __device__ __forceinline__ void touch_line(const float *src) { asm volatile("global_load_dword %0, %1, off" : "+v"(warm) : "v"(src) : "memory"); }
    __device__ __forceinline__ void load_int(int &dst, const int *src) { asm volatile("global_load_dword %0, %1, off" : "=&v"(dst) : "v"(src) : "memory"); }
    template <bool FIRST, int KT>
    __device__ __forceinline__ void top() {               // stage KT + 1 landed (this wave's pieces), every wave through k-step KT - 1
        if constexpr (FIRST) {
            // the gate-matrix registers of this k-step are "produced" here: their loads are older than the stage waited for
            asm volatile("s_waitcnt vmcnt(%4)"
                         : "+a"(W[2 * KT][0]), "+a"(W[2 * KT][1]), "+a"(W[2 * KT + 1][0]), "+a"(W[2 * KT + 1][1])
                         : "n"(ws_top(true, L, KT)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ws_top(false, L, KT)) : "memory");
        }
        if constexpr (!(WS_ABL & 8)) asm volatile("s_barrier" ::: "memory");
    }

    // one half step (k-step KT, half ST): 15 MFMAs;  group 1  W_lo . B_hi[nb];  group 2  W_hi . B_hi[nb], B_hi[nb] refilled behind
    // its MFMA;  group 3  W_hi . B_lo[nb], B_lo[nb] refilled.  The refill is the NEXT half step's fragment (stage KT + 1 from half 1)
    template <int KT, int ST, int DMA, bool ZERO>         // DMA: stage to request behind the first MFMA (-1: none); ZERO: the tile's first half step
    __device__ __forceinline__ void half() {
        constexpr int NKT = ST == 0 ? KT : KT + 1, NST = 1 - ST;       // the half step the refills belong to
        constexpr bool REFILL = !(WS_OPS4 && KT == WS_NK - 1 && ST == 1);   // (the tile's last: the registers take epilogue operands instead)
        const f32x16 zf = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nb = 0; nb < WS_NB; ++nb) {
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[2 * KT + ST][1], B[nb][0], ZERO ? zf : acc[nb], 0, 0, 0);
            if constexpr (DMA >= 0 && !(WS_ABL & 1)) { if (nb == 0) issue<DMA>(); }
        }
#pragma unroll
        for (int nb = 0; nb < WS_NB; ++nb) {
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[2 * KT + ST][0], B[nb][0], acc[nb], 0, 0, 0);
            if constexpr (REFILL) rdB(NKT, NST, nb, 0);
        }
#pragma unroll
        for (int nb = 0; nb < WS_NB; ++nb) {
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[2 * KT + ST][0], B[nb][1], acc[nb], 0, 0, 0);
            if constexpr (REFILL) rdB(NKT, NST, nb, 1);
        }
#pragma unroll
        for (int i = 0; i < 3 * WS_NB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < WS_NP && DMA >= 0 && !(WS_ABL & 1)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // (one LDS-DMA instruction behind each of the first five MFMAs)
            if (i >= WS_NB && !(WS_ABL & 2) && REFILL) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    template <bool FIRST, int KT>
    __device__ __forceinline__ void kstep(int m0, int m0n, bool has_next) {
        top<FIRST, KT>();
        if constexpr (KT == 0 && FIRST) JLM_WS_T(2);
        if constexpr (KT == 0 && (FIRST || WS_OPS4)) {
            // the tile's first fragments (stage 0 landed: it is older than stage 1; later tiles: both landed under the tile before)
#pragma unroll
            for (int nb = 0; nb < WS_NB; ++nb) { rdB(0, 0, nb, 0); rdB(0, 0, nb, 1); }
        }
        __builtin_amdgcn_sched_barrier(0);
        // in front of the first stage of the NEXT tile: its piece rows' predecessors have arrived -> the pieces' row indices
        if constexpr (KT == WS_NK - L) {
#pragma unroll
            for (int i = 0; i < WS_NP; ++i) wait_int<ws_wait_ppn(FIRST, L)>(rn[i]);
#pragma unroll
            for (int i = 0; i < WS_NP; ++i) vidx[i] = (nok[i] && rn[i] >= 0) ? rn[i] * h16 + vslot[i] : OOB_ROW;
        }
        half<KT, 0, (KT + L) & (WS_NK - 1), KT == 0>();
        // ---- the k-step's extra loads (X), then the first tile's gate-matrix loads (W)
        if constexpr (KT == 0 && !FIRST) {
#pragma unroll
            for (int nb = 0; nb < WS_NB; ++nb) {
                const int r = m0 + 32 * nb + li;
                eok[nb] = r < M;
                load_int(eg[nb], a.rows + (eok[nb] ? r : M - 1));
            }
        }
        if constexpr (KT == 1) {
#pragma unroll
            for (int i = 0; i < WS_NP; ++i) {
                const int r = piece_row(m0n, i);
                nok[i] = has_next && r < M;
                load_int(rn[i], a.rows + (nok[i] ? r : M - 1));
            }
        }
        if constexpr (KT == 3 && !FIRST) {
#pragma unroll
            for (int nb = 0; nb < WS_NB; ++nb) wait_int<ws_wait_eg(L)>(eg[nb]);
#pragma unroll
            for (int nb = 0; nb < WS_NB; ++nb) {
                load_int(ep[nb], a.prev + eg[nb]);
                load_int(ew[nb], a.word + eg[nb]);
            }
        }
        if constexpr (KT == 4) {
#pragma unroll
            for (int i = 0; i < WS_NP; ++i) wait_int<ws_wait_rn(FIRST, L)>(rn[i]);
#pragma unroll
            for (int i = 0; i < WS_NP; ++i) load_int(rn[i], a.prev + rn[i]);
        }
        if constexpr (KT == WS_EPEW_AT && !FIRST) {
#pragma unroll
            for (int nb = 0; nb < WS_NB; ++nb) { wait_int<ws_wait_epew(L)>(ep[nb]); wait_int<ws_wait_epew(L)>(ew[nb]); }
        }
        if constexpr (KT == WS_PF) {
#pragma unroll
            for (int nb = 0; nb < WS_NB; ++nb) {
                touch_line(a.xg + (size_t)ew[nb] * (size_t)(4 * H) + n0 + 32 * wave + 4 * hf);
                touch_line(a.c_in + (size_t)(ep[nb] >= 0 ? ep[nb] : 0) * ld + u0);
            }
        }
        if constexpr (KT == ws_kx(0)) load_ops(0, 0);
        if constexpr (KT == ws_kx(1)) load_ops(1, 1);
        if constexpr (KT == ws_kx(2)) load_ops_b<2>(2);
        if constexpr (FIRST && KT + L + 1 < WS_NK) load_w<(KT + L + 1) & (WS_NK - 1)>();
        __builtin_amdgcn_sched_barrier(0);
        half<KT, 1, -1, false>();
    }

    // cell update of block NBK in registers; operands two blocks ahead.  acc[nb][4 gate + e] + table = pre-activation (x 1 / descale) of
    // gate `gate`, unit u0 + e, hypothesis li of block nb
    template <int NBK>
    __device__ __forceinline__ void cell() {
        f32x4 xq[4], cq;
        if constexpr (WS_OPS4) {
            // issue order behind k-step 14's D:  ops0 | D15 ops1 | ops2 ops3 | [cell 0: ops4, stores] [cell 1: stores] ...  (stores counted as
            // WS_ST per block: with the f32 copy there is one more -- a smaller count only waits longer)
            if constexpr (NBK == 0) {
                wait_ops<WS_NP * (15 - ws_kx(0)) + 3 * 5>(0);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) xq[g4] = xg[0][g4];
                cq = cp[0];
                asm volatile("" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]), "+v"(cq));       // (copied out before the set is requested again)
                load_ops(4, 0);
            } else if constexpr (NBK == 1) {
                wait_ops<WS_NP * (15 - ws_kx(1)) + 3 * 5 + WS_ST>(1);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) xq[g4] = xg[1][g4];
                cq = cp[1];
            } else if constexpr (NBK == 2) {
                wait_ops_b<2, 2 * 5 + 2 * WS_ST>(xq, cq);
            } else if constexpr (NBK == 3) {
                wait_ops_b<3, 5 + 3 * WS_ST>(xq, cq);
            } else {
                wait_ops<4 * WS_ST>(0);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) xq[g4] = xg[0][g4];
                cq = cp[0];
            }
        } else {
            constexpr int sl = NBK & 1;
            // younger than this block's operands: the next block's (5), and what was issued between them (stores, the block after)
            constexpr int younger = NBK == 0 ? 10 : NBK == 1 ? 5 + WS_ST : NBK == 4 ? 2 * WS_ST : 2 * WS_ST + 5;
            wait_ops<younger>(sl);
            if constexpr (NBK == 0 && WS_PF >= 0) asm volatile("" : "+v"(warm));        // (in-order returns: the warming loads are older)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) xq[g4] = xg[sl][g4];
            cq = cp[sl];
            if constexpr (NBK + 2 < WS_NB) {
                asm volatile("" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]), "+v"(cq));       // (copied out before the slot is requested again)
                load_ops(NBK + 2, sl);
            }
        }
        const float ks = a.descale * -1.4426950408889634f, kt = a.descale * -2.8853900817779268f;      // (jlm_common.h: exact for ds = 2^-S)
        const int g = eok[NBK] ? eg[NBK] : -1;
        f32x4 cn, hn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gi = jlm_sigmoid_k(acc[NBK][e] + xq[0][e], ks), gf = jlm_sigmoid_k(acc[NBK][4 + e] + xq[1][e], ks);
            const float go = jlm_sigmoid_k(acc[NBK][8 + e] + xq[2][e], ks), gg = jlm_tanh_k(acc[NBK][12 + e] + xq[3][e], kt);
            cn[e] = (ep[NBK] >= 0 ? cq[e] : 0.0f) * gf + gg * gi;
            hn[e] = jlm_tanh(cn[e]) * go;
        }
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x4 hi4, lo4;
        jlm_split4(hn, a.h_scale, hi4, lo4);
        // (rows past the edge store into the dump page: the stores are issued all the same)
        float *crow = g >= 0 ? a.c_out + (size_t)g * ld + u0 : dump;
        *reinterpret_cast<f32x4 *>(crow) = cn;
        // units u0 .. u0+3 = one half of an 8-value block [8 x f16 hi][8 x f16 lo] of the split row
        _Float16 *blk = g >= 0 ? reinterpret_cast<_Float16 *>(a.h_out + (size_t)g * ld + (u0 & ~7)) + (u0 & 7) : reinterpret_cast<_Float16 *>(dump);
        *reinterpret_cast<f16x4 *>(blk) = hi4;
        *reinterpret_cast<f16x4 *>(g >= 0 ? blk + 8 : blk + 4) = lo4;
        if constexpr (HF32) *reinterpret_cast<f32x4 *>(g >= 0 ? a.h_f32 + (size_t)g * ld + u0 : dump) = hn;
    }

    // one tile: sixteen k-steps, then the cell update of its five blocks
    template <bool FIRST>
    __device__ __forceinline__ void tile(bool has_next, int t_no) {
        const int m0 = tm * GT_BM;
        const int m0n = (tm + Q) * GT_BM;                 // the next tile of this workgroup (past the edge: every row masked)
        gate_for_each_ic([&](auto ktc) { this->template kstep<FIRST, decltype(ktc)::value>(m0, m0n, has_next); },
                         std::make_integer_sequence<int, WS_NK>{});
        JLM_WS_T(3 + 2 * t_no);
        if constexpr (ws_kx(2) == 16) load_ops_b<2>(2);
        if constexpr (ws_kx(3) == 16) load_ops_b<3>(3);
        gate_for_each_ic([&](auto nbc) { this->template cell<decltype(nbc)::value>(); }, std::make_integer_sequence<int, WS_NB>{});
        JLM_WS_T(4 + 2 * t_no);
    }

    __device__ __forceinline__ void run() {
        JLM_WS_T(0);
        warm = 0;
        lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);            // = the wave's gate block
        li = lane & 31; hf = lane >> 5;
        lrow = lane >> 3; lslot = lane & 7;
        H = a.H; ld = a.ld;
        // tile map (block b runs on XCD b % 8, observed; speed only).  The gate matrix is read once per workgroup, so what the map
        // has to keep small is the fabric traffic of the STATE rows: an XCD whose workgroups cover `cx` of the 16 gate-column tiles
        // brings a hypothesis row into its L2 once for all of them -- 16 / cx XCDs fetch each row (2 KB) and each XCD fetches
        // cx / 16 of the gate matrix (4 MB).  cx = tiles_n (every column tile of a row tile on ONE XCD) when the launch has many
        // rows, fewer for few rows (the launcher picks: a.cx); grids below 256 workgroups keep the plain order.
        const int b = blockIdx.x;
        int tn, q;
        if ((int)gridDim.x == 256 && a.cx >= 2) {
            const int x = b & 7, jb = b >> 3;             // XCD, workgroup inside it (32 per XCD)
            const int gc = a.tiles_n / a.cx;              // column groups; XCD x = (row group, column group)
            const int cg = x % gc, rg = x / gc, per = 32 / a.cx;
            tn = cg * a.cx + jb % a.cx;
            q = rg * per + jb / a.cx;
            Q = 256 / a.tiles_n;
        } else {
            Q = (int)gridDim.x / a.tiles_n;
            tn = b % a.tiles_n;
            q = b / a.tiles_n;
        }
        M = a.ndev ? min(*a.ndev, a.nrows) : a.nrows;
        tiles_m = (M + GT_BM - 1) / GT_BM;
        if (q >= tiles_m) return;
        n0 = tn * GT_BN;
        u0 = (n0 >> 2) + 8 * wave + 4 * hf;               // the lane's four hidden units
        // (16-byte records and one address register per piece: csrc/jlm_gate.hip gate_xg_body_u says why)
        rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.h), (short)16, 0x7ffffff0, 0x00020000);
        h16 = ld >> 2;
        // stores are issued by EVERY lane, always (the number of vector-memory operations in flight is known at every wait): a
        // hypothesis row past the edge writes into the lane's own 16 bytes of a dump page instead
        dump = gate_dump_page + 4 * (int)threadIdx.x;
        wrow = a.wt + (size_t)(n0 + 32 * wave + li) * H + hf * 8;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int p = 0; p < 2; ++p) goff[st][p] = li * 32 + (((4 * st + 2 * hf + p) ^ ((li >> 1) & 7)) * 4);
#pragma unroll
        for (int i = 0; i < WS_NP; ++i) {
            const int row = 8 * (WS_NP * wave + i) + lrow;
            vslot[i] = lslot ^ ((row >> 1) & 7);
        }
        // ---- prologue (first tile): every index by ordinary loads, nothing asynchronous is in flight yet (a.rows != NULL: the
        //      launcher sends calls without a row list to gate_xg_u16_kernel).  (Requesting the first gate fragments in front of
        //      the index loads was measured and is slower: the two dependent index round trips then queue behind 128 KB per CU.)
        tm = q;
        {
            const int m0 = tm * GT_BM;
#pragma unroll
            for (int nb = 0; nb < WS_NB; ++nb) {
                const int r = m0 + 32 * nb + li;
                eok[nb] = r < M;
                eg[nb] = a.rows[eok[nb] ? r : M - 1];
            }
#pragma unroll
            for (int i = 0; i < WS_NP; ++i) {
                const int r = piece_row(m0, i);
                nok[i] = r < M;
                rn[i] = a.rows[nok[i] ? r : M - 1];
            }
#pragma unroll
            for (int nb = 0; nb < WS_NB; ++nb) {
                ep[nb] = a.prev[eg[nb]];
                ew[nb] = a.word[eg[nb]];
            }
#pragma unroll
            for (int i = 0; i < WS_NP; ++i) {
                const int p = a.prev[rn[i]];
                vidx[i] = (nok[i] && p >= 0) ? p * h16 + vslot[i] : OOB_ROW;
            }
            // "used" here: with LDS-DMA pieces in flight hipcc waits vmcnt(0) in front of the first use of an ordinary load's result
#pragma unroll
            for (int nb = 0; nb < WS_NB; ++nb) { asm volatile("" : "+v"(eg[nb])); asm volatile("" : "+v"(ep[nb])); asm volatile("" : "+v"(ew[nb])); }
#pragma unroll
            for (int i = 0; i < WS_NP; ++i) asm volatile("" : "+v"(vidx[i]));
        }
        JLM_WS_T(1);
        // W0 D0 W1 D1 ... W(L-1) D(L-1) W(L)
        gate_for_each_ic([&](auto sc) {
            constexpr int SG = decltype(sc)::value;
            this->template load_w<SG>();
            if constexpr (SG < L) this->template issue<SG>();
        }, std::make_integer_sequence<int, L + 1>{});
        tile<true>(tm + Q < tiles_m, 0);
        tm += Q;
        for (int t_no = 1; tm < tiles_m; tm += Q, ++t_no) tile<false>(tm + Q < tiles_m, t_no);
    }
};

template <int L, bool HF32>
__global__ __launch_bounds__(256, 1) void gate_ws_kernel(GateXgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    GateWs<L, HF32> k(a, smem);
    k.run();
}


}  // namespace

int jlm_gate::ws_launch(const GateXgArgs &a, int L, int Q, hipStream_t stream) {
    const bool hf32 = a.h_f32 != nullptr;
    const void *fn = L == 3 ? (hf32 ? reinterpret_cast<const void *>(gate_ws_kernel<3, true>) : reinterpret_cast<const void *>(gate_ws_kernel<3, false>))
                            : (hf32 ? reinterpret_cast<const void *>(gate_ws_kernel<7, true>) : reinterpret_cast<const void *>(gate_ws_kernel<7, false>));
    const int lds = (L + 1) * WS_STAGE_FLOATS * 4;
    static JlmLdsGrant grant[4];
    if (int rc = jlm_grant_lds(grant[(L == 3 ? 0 : 2) + (hf32 ? 1 : 0)], fn, lds)) return rc;
    GateXgArgs args = a;
    void *params[] = {&args};
    hipError_t e = hipLaunchKernel(fn, dim3(a.tiles_n * Q), dim3(256), params, lds, stream);
    return e == hipSuccess ? 0 : (int)e;
}
