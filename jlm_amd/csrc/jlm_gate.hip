// jlm_gate.hip -- the fused LSTM step of the decode, table form (jlm_lstm_step_xg, include/jlm_hip.h).
//
//   z[g][n] = sum_k h[prev[g]][k] * W_h[n][k]  +  xgate[word[g]][n]          (decoder/model.py:125-131)
//   c[g] = c[prev[g]] * sig(z_f) + tanh(z_g) * sig(z_i);  h[g] = tanh(c[g]) * sig(z_o)   (model.py:133-139)
//
// Shape at the decode: R = 2 560 rows x 4H = 2 048 gate columns x K = H = 512: 16 GFLOP-passes (6.4 us of the matrix
// pipe at its nominal rate, 7.9 us at the ~2.0 GHz the chip sustains under MFMAs).  Every output tile needs
// (rows + columns) x 2 KB of split rows from L2; the tile form it replaces (128 x 64 tiles, 640 workgroups) staged 245 MB
// per launch.  Here:
//   * ONE 160-hypothesis x 128-gate-column tile per CU (16 x 16 = 256 workgroups at R = 2 560, H = 512):
//     (160 + 128) x 2 KB = 576 KB per CU, 147 MB per launch -- the minimum of (m + n) at m n = 20 480;
//   * 8 waves = two per SIMD: wave w owns gate block w & 3 (32 gate columns) and hypothesis blocks 0..2 (w < 4) or
//     3..4 (w >= 4), so each SIMD has 5 blocks of MFMA work and its two waves cover each other;
//   * a 4-stage LDS ring (36 KB per 32-value k-step) filled by global_load_lds_dwordx4, three stages (108 KB per CU) in
//     flight across the one raw s_barrier per k-step, counted s_waitcnt vmcnt;
//   * a software pipeline at HALF k-step granularity with the fragments double-buffered in registers: fragment reads and
//     LDS-DMA issue run under MFMAs, not in front of them;
//   * SWAPPED orientation: the gate matrix is the MFMA A operand (D rows = gate columns), the hypotheses are the B
//     operand (D columns), and the gate matrix is packed in the row order
//         n = (u / 8) * 32 + gate * 8 + (u % 8)          gate order i, f, o, g
//     so that a lane's 16 accumulator registers of a 32 x 32 block are the FOUR gates of FOUR units
//     (acc[4 gate + e] = gate of unit 8 (n / 32) + 4 (lane >> 5) + e) of ONE hypothesis (lane & 31): the LSTM cell update
//     runs in registers -- no transposition through LDS, no second barrier phase;
//   * the input side xgate[word] (table in the same column order, pre-multiplied by 1 / descale) and the old cell state
//     are requested in the MIDDLE of the mainloop, wave w at k-step 1 + w, and added in the epilogue: their HBM round trip
//     (the table is hundreds of MB, rows are random) hides under three k-steps, the chip's 26 MB of them spread over eight
//     k-steps, and -- the vmcnt counter being in order -- they do not sit in front of the ring's first stage as they would
//     if they were requested at kernel start.
// Measured and rejected (tools/probes/gate_xg_profile.py, gate_loop.hip): warming L2 with one load per operand
// line at kernel start (+5 us: the loop is not bound by L2 misses -- with every operand L2-resident and no DMA at all a
// k-step still takes 0.8 us); a whole-k-step register double buffer (spills at 3 hypothesis blocks).
#include "jlm_common.h"
#include "jlm_gate.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include <vector>

#define GLDS16(gp, lp)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp),      \
                                     (__attribute__((address_space(3))) void *)(lp), 16, 0, 0)

// Per-workgroup timeline (-DJLM_PROFILE builds only, tools/probes/gate_xg_profile.py): waves 0 and 4 of a workgroup stamp the
// 100 MHz wall clock at kernel start, after the index chains, when the first stage has landed, after the steady-state
// k-steps, after the mainloop, at the end.
#ifdef JLM_PROFILE
static __device__ unsigned long long jlm_gate_time[2048][2][8];   // [6], [7]: shader clock at stamps 2 and 3
#define JLM_GT_T(i) do { if ((threadIdx.x & 255) == 0) { jlm_gate_time[blockIdx.x & 2047][threadIdx.x >> 8][i] = wall_clock64(); \
    if ((i) == 2 || (i) == 3) jlm_gate_time[blockIdx.x & 2047][threadIdx.x >> 8][(i) == 2 ? 6 : 7] = clock64(); } } while (0)
extern "C" int jlm_prof_read_gate(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(jlm_gate_time), sizeof(jlm_gate_time)) == hipSuccess ? 0 : -1;
}
#else
#define JLM_GT_T(i) (void)0
#endif

// the NP LDS-DMA instructions of a k-step: 1 one behind each of the half step's first NP MFMAs, 0 all behind the first (rounds 2-3).
// Interleaved A/B, tools/gpu_gate_spread.sh: the persistent kernels -2 ... -3.5 % (72.4 vs 73.5 us at 10 240 rows, 151 vs 157 at
// 20 480), the one-tile kernel unchanged; a wider spacing (every second / third MFMA) measures the same as 1.
#ifndef GT_DMA_SPREAD
#define GT_DMA_SPREAD 1
#endif

namespace {

constexpr int GT_STAGES = 4;
constexpr int GT_STAGE_FLOATS = (GT_BM + GT_BN) * 32;    // one 32-value k-step of both operands: 36 KB
constexpr int GT_LDS_BYTES = GT_STAGES * GT_STAGE_FLOATS * 4;

__device__ float gate_zero_page[64];



// NB hypothesis blocks of MFMA work, NP LDS-DMA pieces (8 rows x 128 B each) per stage for this wave.
template <int NB, int NP>
__device__ __forceinline__ void gate_xg_body(const GateXgArgs &a, const int m0, const int n0, const int M, const int wave,
                                             const int lane, float *smem) {
    const int gb = wave & 3;                             // gate block of this wave
    const int hb0 = (wave >> 2) ? 3 : 0;                 // its first hypothesis block
    const int li = lane & 31, hf = lane >> 5;
    const int H = a.H, ld = a.ld;
    const int u0 = (n0 >> 2) + 8 * gb + 4 * hf;          // the lane's four hidden units
    constexpr int NXG = 5 * NB;                          // loads of the epilogue operands: 4 table quads + 1 cell quad per block

    JLM_GT_T(0);
    // ---- index chains first, all of them together (two dependent round trips, not two per consumer):
    //      rows of the epilogue (this lane's hypothesis in each of its NB blocks) and of the wave's NP - 2 state pieces.
    //      Loads are unconditional (a clamped row is always readable) and masked afterwards: no branch per load.
    const int lrow = lane >> 3, lslot = lane & 7;
    int eg[NB], ep[NB], ew[NB], pp[NP - 2];
    bool eok[NB], pok[NP - 2];
    int prow[NP - 2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int r = m0 + 32 * (hb0 + nb) + li;
        eok[nb] = r < M;
        const int rc = eok[nb] ? r : M - 1;
        eg[nb] = a.rows ? a.rows[rc] : rc;
    }
#pragma unroll
    for (int i = 0; i < NP - 2; ++i) {
        const int pidx = (NP == 4) ? 2 * wave + i : 8 + 3 * (wave - 4) + i;
        prow[i] = 8 * pidx;
        const int r = m0 + prow[i] + lrow;
        pok[i] = r < M;
        const int rc = pok[i] ? r : M - 1;
        pp[i] = a.rows ? a.rows[rc] : rc;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        ep[nb] = a.prev[eg[nb]];
        ew[nb] = a.word[eg[nb]];
    }
#pragma unroll
    for (int i = 0; i < NP - 2; ++i) pp[i] = a.prev[pp[i]];
    // the indices are "used" here, before the first LDS-DMA goes out: with pieces in flight hipcc waits vmcnt(0) at the
    // first use of an ordinary load's result, and the first use of ew / ep is the request of the epilogue operands in the
    // middle of the mainloop -- it would drain the ring there.  They arrive with pp (same round trip), which the source
    // addresses below need anyway.
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        asm volatile("" : "+v"(ew[nb]));
        asm volatile("" : "+v"(ep[nb]));
    }
#ifdef JLM_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    JLM_GT_T(1);
#endif

    // ---- LDS-DMA pieces of this wave: 2 of the gate matrix, NP - 2 of the gathered state rows.
    //      lane = (row lane >> 3 of the piece's 8 rows, 16-byte slot lane & 7); the source granule is
    //      slot ^ ((row >> 1) & 7): the swizzle sits on the SOURCE address, the LDS image stays lane-linear
    //      and the fragment reads below apply the same XOR (conflict-free ds_read_b128, jlm_gemm.hip).
    const char *src[NP];
    int inc[NP];
    int dst[NP];                                         // float offset inside a stage (wave-uniform)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i < 2) {
            const int pr = 8 * (2 * wave + i);
            const int row = pr + lrow;
            src[i] = reinterpret_cast<const char *>(a.wt + (size_t)(n0 + row) * H + ((lslot ^ ((row >> 1) & 7)) * 4));
            inc[i] = 128;
            dst[i] = pr * 32;
        } else {
            const int row = prow[i - 2] + lrow;
            const int p = pok[i - 2] ? pp[i - 2] : -1;
            const int gr = (lslot ^ ((row >> 1) & 7)) * 4;
            src[i] = p >= 0 ? reinterpret_cast<const char *>(a.h + (size_t)p * ld + gr)
                            : reinterpret_cast<const char *>(gate_zero_page + gr);
            inc[i] = p >= 0 ? 128 : 0;
            dst[i] = (GT_BN + prow[i - 2]) * 32;
        }
    }
    auto issue = [&](int stg) {
        float *base = smem + stg * GT_STAGE_FLOATS;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            GLDS16(src[i], base + dst[i]);
            src[i] += inc[i];
        }
    };

    // epilogue operands: the word's table row (the four gates of the lane's four units) and the old cell state.
    // Requested by INLINE-ASM loads, hidden from hipcc: in front of a load it can see, with LDS-DMA pieces in flight, it
    // drains the queue (s_waitcnt vmcnt(0) before the load is even issued).  The destinations are plain asm outputs:
    // nothing reads them until the asm wait in front of the epilogue, which takes every one of them as an in/out operand.
    f32x4 xg[NB][4], cp[NB];
    const float *xrp[NB], *cpp[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        xrp[nb] = a.xg + (size_t)ew[nb] * (size_t)(4 * H) + n0 + 32 * gb + 4 * hf;
        cpp[nb] = a.c_in + (size_t)(ep[nb] >= 0 ? ep[nb] : 0) * ld + u0;
    }
    auto load_epilogue_operands = [&]() {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            asm volatile("global_load_dwordx4 %0, %4, off\n\t"
                         "global_load_dwordx4 %1, %4, off offset:32\n\t"
                         "global_load_dwordx4 %2, %4, off offset:64\n\t"
                         "global_load_dwordx4 %3, %4, off offset:96"
                         : "=&v"(xg[nb][0]), "=&v"(xg[nb][1]), "=&v"(xg[nb][2]), "=&v"(xg[nb][3]) : "v"(xrp[nb]) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(cp[nb]) : "v"(cpp[nb]) : "memory");    // masked in the epilogue
        }
    };
    auto wait_epilogue_operands = [&]() {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(xg[nb][0]), "+v"(xg[nb][1]), "+v"(xg[nb][2]), "+v"(xg[nb][3]), "+v"(cp[nb])::"memory");
    };

    // fragment offsets (floats) inside a stage: granule (4 step + 2 half + plane) of row li, swizzled
    int goff[2][2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int p = 0; p < 2; ++p) goff[st][p] = li * 32 + (((4 * st + 2 * hf + p) ^ ((li >> 1) & 7)) * 4);
    const int w_off = gb * 32 * 32;                      // this wave's gate block inside the stage
    const int h_off = (GT_BN + hb0 * 32) * 32;           // its first hypothesis block

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;

    struct Frag { f16x8 aw[2]; f16x8 bh[NB][2]; };
    auto read_half = [&](int stage, int st, Frag &f) {
        const float *ws = smem + (stage & 3) * GT_STAGE_FLOATS + w_off;
        const float *hs = smem + (stage & 3) * GT_STAGE_FLOATS + h_off;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            f.aw[p] = *reinterpret_cast<const f16x8 *>(ws + goff[st][p]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) f.bh[nb][p] = *reinterpret_cast<const f16x8 *>(hs + nb * 1024 + goff[st][p]);
        }
    };
    auto mfmas = [&](const Frag &f) {
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)                   // lo.hi, hi.lo, hi.hi; blocks inner: no two consecutive MFMAs share an accumulator
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.aw[pr == 0 ? 1 : 0], f.bh[nb][pr == 1 ? 1 : 0], acc[nb], 0, 0, 0);
    };
    auto touch = [&](Frag &f) {
        // "use" the fragments HERE: hipcc puts the wait for a ds_read in front of its first use and merges the pending
        // state of all paths into a block, so without this it waits lgkmcnt(0) in front of the next MFMA group -- right
        // after that group's own reads were issued.  Behind the MFMAs the reads have long returned.
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            asm volatile("" : "+v"(f.aw[p]));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) asm volatile("" : "+v"(f.bh[nb][p]));
        }
    };
    // One k-step of the software pipeline (fa holds step 0 of stage kt on entry):
    //     read(kt, step 1) -> fb ; MFMAs(kt, step 0) from fa ; [fb has returned]
    //     wait(stage kt+1 landed, VM younger operations may stay in flight) ; barrier ; [request stage kt+4 into the slot of stage kt]
    //     read(kt+1, step 0) -> fa ; MFMAs(kt, step 1) from fb ; [request stage kt+4] ; [fa has returned]
    // Every wave has its reads of stage kt back before it arrives at the barrier of k-step kt, so the slot is free behind
    // it.  The two waves of a SIMD are staggered: waves 0-3 (9 MFMAs per half step) request right after the barrier,
    // waves 4-7 (6 MFMAs) after their MFMAs, so one is on the matrix pipe while the other pays the DMA issue cost.
    constexpr bool DMA_FIRST = (NP == 4);
    Frag fa, fb;
    // issue order of a half step: ONE fragment read behind every MFMA (tools/probes/gate_loop.hip: the 2 + 2 NB reads grouped in
    // front of the MFMAs cost 12 % of the k-step, interleaved 1:1 they are free)
    auto interleave = [&](bool with_reads) {
#pragma unroll
        for (int i = 0; i < 3 * NB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (with_reads && i < 2 + 2 * NB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto kstep = [&](auto has_next, auto vm, auto do_issue, int kt) {
        __builtin_amdgcn_sched_barrier(0);
        read_half(kt, 1, fb);
        mfmas(fa);
        interleave(true);
        touch(fb);
        if constexpr (decltype(has_next)::value) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(decltype(vm)::value) : "memory");
            if constexpr (DMA_FIRST && decltype(do_issue)::value) issue(kt & 3);
            __builtin_amdgcn_sched_barrier(0);
            read_half(kt + 1, 0, fa);
        }
        mfmas(fb);
        interleave(decltype(has_next)::value);
        if constexpr (!DMA_FIRST && decltype(do_issue)::value) issue(kt & 3);
        if constexpr (decltype(has_next)::value) touch(fa);
    };
    using T = std::true_type;
    using F = std::false_type;

    const int nk = H / 32;
    if (nk >= GT_STAGES) {
#pragma unroll
        for (int q = 0; q < GT_STAGES; ++q) issue(q);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(3 * NP) : "memory");
        JLM_GT_T(2);
        read_half(0, 0, fa);
        touch(fa);
        int kt = 0;
        // The epilogue's operands (NXG loads: the word's table row out of a 400-MB table -- HBM -- and the old cell state) go
        // out in the MIDDLE of the mainloop, each wave at its own k-step (1 + wave: the chip's 26 MB spread over eight k-steps
        // instead of one burst).  The vmcnt counter is in order: the three k-steps behind the request wait with the loads
        // allowed in flight (they are younger than the stages waited for); the k-step after those waits for a stage that was
        // requested behind them, so by then -- 3 k-steps, ~2.7 us -- they must have landed.  With too few k-steps for that
        // they go out at the start of the last four k-steps instead (nothing is requested behind them there).
        const int xg_at = 1 + wave;
        const bool mid = xg_at + 3 + 4 <= nk;
        if (mid) {
            for (; kt < xg_at; ++kt) kstep(T{}, IC<2 * NP>{}, T{}, kt);
            load_epilogue_operands();
            kstep(T{}, IC<2 * NP + NXG>{}, T{}, kt);
            kstep(T{}, IC<2 * NP + NXG>{}, T{}, kt + 1);
            kstep(T{}, IC<2 * NP + NXG>{}, T{}, kt + 2);
            kt += 3;
        }
        for (; kt + 4 < nk; ++kt) kstep(T{}, IC<2 * NP>{}, T{}, kt);       // steady state: stages kt+1 .. kt+4 exist
        JLM_GT_T(5);
        // the last four k-steps: nothing left to request for the ring
        if (mid) {
            kstep(T{}, IC<2 * NP>{}, F{}, kt);
            kstep(T{}, IC<NP>{}, F{}, kt + 1);
            kstep(T{}, IC<0>{}, F{}, kt + 2);
            kstep(F{}, IC<0>{}, F{}, kt + 3);
        } else {
            load_epilogue_operands();
            kstep(T{}, IC<2 * NP + NXG>{}, F{}, kt);
            kstep(T{}, IC<NP + NXG>{}, F{}, kt + 1);
            kstep(T{}, IC<NXG>{}, F{}, kt + 2);
            kstep(F{}, IC<0>{}, F{}, kt + 3);
        }
    } else {
        // short contractions (H < 128): everything requested up front, plain waits
        load_epilogue_operands();
        for (int q = 0; q < nk; ++q) issue(q);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        JLM_GT_T(2);
        read_half(0, 0, fa);
        touch(fa);
        JLM_GT_T(5);
        for (int kt = 0; kt < nk; ++kt) {
            read_half(kt, 1, fb);
            mfmas(fa);
            if (kt + 1 < nk) read_half(kt + 1, 0, fa);
            mfmas(fb);
        }
    }
    JLM_GT_T(3);
    wait_epilogue_operands();

    // ---- cell update in registers: acc[4 gate + e] + table = pre-activation (x 1 / descale) of gate `gate`, unit u0 + e,
    //      hypothesis li
    const float ks = a.descale * -1.4426950408889634f, kt = a.descale * -2.8853900817779268f;      // (jlm_common.h: exact for ds = 2^-S)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int g = eok[nb] ? eg[nb] : -1;
        if (g < 0) continue;
        f32x4 cn, hn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gi = jlm_sigmoid_k(acc[nb][e] + xg[nb][0][e], ks), gf = jlm_sigmoid_k(acc[nb][4 + e] + xg[nb][1][e], ks);
            const float go = jlm_sigmoid_k(acc[nb][8 + e] + xg[nb][2][e], ks), gg = jlm_tanh_k(acc[nb][12 + e] + xg[nb][3][e], kt);
            cn[e] = (ep[nb] >= 0 ? cp[nb][e] : 0.0f) * gf + gg * gi;
            hn[e] = jlm_tanh(cn[e]) * go;
        }
        *reinterpret_cast<f32x4 *>(a.c_out + (size_t)g * ld + u0) = cn;
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x4 hi4, lo4;
        jlm_split4(hn, a.h_scale, hi4, lo4);
        // units u0 .. u0+3 = one half of an 8-value block [8 x f16 hi][8 x f16 lo] of the split row
        _Float16 *blk = reinterpret_cast<_Float16 *>(a.h_out + (size_t)g * ld + (u0 & ~7)) + (u0 & 7);
        *reinterpret_cast<f16x4 *>(blk) = hi4;
        *reinterpret_cast<f16x4 *>(blk + 8) = lo4;
        if (a.h_f32) *reinterpret_cast<f32x4 *>(a.h_f32 + (size_t)g * ld + u0) = hn;
    }
#ifdef JLM_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    JLM_GT_T(4);
#endif
}


// ---- the same tile with the fragment reads a whole half step ahead ("refill in place"; H / 32 = NK k-steps, unrolled) ----
// The form above reads the fragments of half step s + 1 behind the MFMAs of half step s and drains the LDS queue
// (lgkmcnt(0)) at the end of every half step: the last read goes out one MFMA before the wait, so a full LDS round trip
// (hundreds of cycles with 8 waves' reads and the DMA writes in the queue) is exposed twice per k-step and wave.  Here
// the three products of a half step run in the order  lo.hi -> hi.hi -> hi.lo  so that the fragment registers die one
// after the other -- A_lo behind the first group, B_hi[nb] behind MFMA nb of the second, B_lo[nb] behind MFMA nb of the
// third, A_hi at the end -- and each register is REFILLED IN PLACE with its value of half step s + 2 right behind the
// MFMA that read it last.  Two register sets (even / odd half steps), as before; every read now has >= 12 (NB = 3) / 8
// (NB = 2) MFMAs = 380 / 250 cycles of cover, and the waits are the counted lgkmcnt(n) hipcc derives in straight-line
// code (in-order LDS returns), never a drain.  Stage kt + 1 is read DURING k-step kt, so it must have landed at the
// barrier at the START of k-step kt; the slot refilled behind that barrier is the one of stage kt - 1 (stage kt's last
// reads are still in the queue there): three stages resident + one being filled, two k-steps of DMA lead.

template <int NB, int NP, int NK, int BS, int ABL>
__device__ __forceinline__ void gate_xg_body_u(const GateXgArgs &a, const int m0, const int n0, const int M, const int wave,
                                               const int lane, float *smem) {
    const int gb = wave & 3;
    const int hb0 = (wave >> 2) ? 3 : 0;
    const int li = lane & 31, hf = lane >> 5;
    const int H = a.H, ld = a.ld;
    const int u0 = (n0 >> 2) + 8 * gb + 4 * hf;
    constexpr int NXG = 5 * NB;

    JLM_GT_T(0);
    const int lrow = lane >> 3, lslot = lane & 7;
    int eg[NB], ep[NB], ew[NB], pp[NP - 2];
    bool eok[NB], pok[NP - 2];
    int prow[NP - 2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int r = m0 + 32 * (hb0 + nb) + li;
        eok[nb] = r < M;
        const int rc = eok[nb] ? r : M - 1;
        eg[nb] = a.rows ? a.rows[rc] : rc;
    }
#pragma unroll
    for (int i = 0; i < NP - 2; ++i) {
        const int pidx = (NP == 4) ? 2 * wave + i : 8 + 3 * (wave - 4) + i;
        prow[i] = 8 * pidx;
        const int r = m0 + prow[i] + lrow;
        pok[i] = r < M;
        const int rc = pok[i] ? r : M - 1;
        pp[i] = a.rows ? a.rows[rc] : rc;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        ep[nb] = a.prev[eg[nb]];
        ew[nb] = a.word[eg[nb]];
    }
#pragma unroll
    for (int i = 0; i < NP - 2; ++i) pp[i] = a.prev[pp[i]];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        asm volatile("" : "+v"(ew[nb]));
        asm volatile("" : "+v"(ep[nb]));
    }
#ifdef JLM_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    JLM_GT_T(1);
#endif

    // LDS-DMA pieces of this wave (2 of the gate matrix, NP - 2 of the gathered state rows), as BUFFER loads:
    //   buffer_load_dwordx4 v[row index : byte offset], rsrc, s(128 * k-step) idxen offen lds
    // address = base + index * stride + offset + soffset, computed by the texture unit in 64 bits: no per-k-step address
    // arithmetic (the k-step is the scalar offset -- NOT the instruction's immediate offset, which a load to LDS also
    // adds to the LDS address), two registers per piece for the whole kernel,
    // and a row index at or above num_records reads zeros (rows past the edge, hypotheses without a predecessor).
    // And, decisive for the pipeline below: hipcc treats global_load_lds (a FLAT encoding) as an access that may
    // return through either counter and, while one is outstanding -- here always --, turns EVERY wait for an LDS
    // read into s_waitcnt lgkmcnt(0); behind buffer_load ... lds it counts (lgkmcnt(n), in-order returns).
    constexpr int OOB_ROW = 0x7fffffff;
    // Round 4: 16-byte records and ONE address register (idxen): index = row x (row bytes / 16) + the lane's swizzled 16-byte slot.  The
    // two-register form (idxen + offen: row index, byte offset) costs 32.5 issue cycles per instruction against 25
    // (tools/probes/lds_dma_issue.hip), and hipcc builds the register pair with moves in front of every instruction.
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wt), (short)16, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.h), (short)16, 0x7ffffff0, 0x00020000);
    const int w16 = H >> 2, h16 = ld >> 2;               // 16-byte records per row
    int vidx[NP];
    int dst[NP];                                         // float offset inside a stage (wave-uniform)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i < 2) {
            const int pr = 8 * (2 * wave + i);
            const int row = pr + lrow;
            vidx[i] = (n0 + row) * w16 + (lslot ^ ((row >> 1) & 7));
            dst[i] = pr * 32;
        } else {
            const int row = prow[i - 2] + lrow;
            const int p = pok[i - 2] ? pp[i - 2] : -1;
            vidx[i] = p >= 0 ? p * h16 + (lslot ^ ((row >> 1) & 7)) : OOB_ROW;
            dst[i] = (GT_BN + prow[i - 2]) * 32;
        }
    }
    auto issue = [&](auto stage_c) {
        constexpr int S = decltype(stage_c)::value;
        float *base = smem + (S & 3) * GT_STAGE_FLOATS;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            __builtin_amdgcn_struct_ptr_buffer_load_lds(i < 2 ? rs_w : rs_h, (__attribute__((address_space(3))) void *)(base + dst[i]),
                                                        16, vidx[i], 0, S * 128, 0, 0);
    };

    f32x4 xg[NB][4], cp[NB];
    if constexpr (ABL & 16) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            cp[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) xg[nb][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float *xrp[NB], *cpp[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        xrp[nb] = a.xg + (size_t)ew[nb] * (size_t)(4 * H) + n0 + 32 * gb + 4 * hf;
        cpp[nb] = a.c_in + (size_t)(ep[nb] >= 0 ? ep[nb] : 0) * ld + u0;
    }
    auto load_epilogue_operands = [&]() {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            asm volatile("global_load_dwordx4 %0, %4, off\n\t"
                         "global_load_dwordx4 %1, %4, off offset:32\n\t"
                         "global_load_dwordx4 %2, %4, off offset:64\n\t"
                         "global_load_dwordx4 %3, %4, off offset:96"
                         : "=&v"(xg[nb][0]), "=&v"(xg[nb][1]), "=&v"(xg[nb][2]), "=&v"(xg[nb][3]) : "v"(xrp[nb]) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(cp[nb]) : "v"(cpp[nb]) : "memory");
        }
    };
    auto wait_epilogue_operands = [&]() {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(xg[nb][0]), "+v"(xg[nb][1]), "+v"(xg[nb][2]), "+v"(xg[nb][3]), "+v"(cp[nb])::"memory");
    };

    int goff[2][2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int p = 0; p < 2; ++p) goff[st][p] = li * 32 + (((4 * st + 2 * hf + p) ^ ((li >> 1) & 7)) * 4);
    const int w_off = gb * 32 * 32;
    const int h_off = (GT_BN + hb0 * 32) * 32;

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;

    // fragment registers: plane 0 = hi, 1 = lo.  The A (gate matrix) fragments exist twice (even / odd half steps); the B
    // (hypothesis) fragments twice where the registers allow it (BS = 2: the two-block waves) and once where they do not
    // (BS = 1, the three-block waves: 24 registers less; B_hi then has 2 NB - 1 = 5 MFMAs of cover, B_lo 8).
    f16x8 A[2][2], B[BS][NB][2];
    // ABL (measurement builds, -DJLM_GATE_ABLATE): 1 no MFMAs, 2 no fragment reads in the loop, 4 no LDS-DMA in the loop,
    // 8 no barrier, 16 no epilogue operand loads.  Results are wrong by construction; the time is the answer.
    auto rdA = [&](int stage, int st, int p) {
        if ((ABL & 2) && stage > 0) return;
        A[st][p] = *reinterpret_cast<const f16x8 *>(smem + (stage & 3) * GT_STAGE_FLOATS + w_off + goff[st][p]);
    };
    auto rdB = [&](int stage, int st, int nb, int p) {
        if ((ABL & 2) && (stage > 0 || st >= BS)) return;
        B[st % BS][nb][p] = *reinterpret_cast<const f16x8 *>(smem + (stage & 3) * GT_STAGE_FLOATS + h_off + nb * 1024 + goff[st][p]);
    };
    auto mf = [&](int st, int pa, int nb, int pb) {
        if constexpr (ABL & 1) asm volatile("" : "+v"(acc[nb]) : "v"(A[st][pa]), "v"(B[st % BS][nb][pb]));
        else acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[st][pa], B[st % BS][nb][pb], acc[nb], 0, 0, 0);
    };
    // One half step (k-step kt, half st): 3 NB MFMAs; behind them the refills.
    //   group 1  A_lo . B_hi[nb]     then A_lo is free
    //   group 2  A_hi . B_hi[nb]     B_hi[nb] free behind MFMA nb
    //   group 3  A_hi . B_lo[nb]     B_lo[nb] free behind MFMA nb, A_hi behind the last
    // A is refilled with half st of k-step kt + 1; B with the same (BS = 2) or with the NEXT half step (BS = 1).
    auto half = [&](auto kt_c, auto st_c, auto dma_c) {
        constexpr int KT = decltype(kt_c)::value, ST = decltype(st_c)::value;
        constexpr int DMA = decltype(dma_c)::value;      // stage to request behind the first MFMA (-1: none)
        constexpr bool RA = KT + 1 < NK;                 // a next k-step exists
        constexpr int BKT = (BS == 2) ? KT + 1 : (ST == 0 ? KT : KT + 1), BST = (BS == 2) ? ST : 1 - ST;
        constexpr bool RB = BKT < NK;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            mf(ST, 1, nb, 0);
            if constexpr (DMA >= 0 && !(ABL & 4)) { if (nb == 0) issue(IC<DMA>{}); }
        }
        if (RA) rdA(KT + 1, ST, 1);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            mf(ST, 0, nb, 0);
            if (RB) rdB(BKT, BST, nb, 0);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            mf(ST, 0, nb, 1);
            if (RB) rdB(BKT, BST, nb, 1);
        }
        if (RA) rdA(KT + 1, ST, 0);
        // pin the issue order: one MFMA, then the read(s) that follow it in the text above
#pragma unroll
        for (int i = 0; i < 3 * NB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (GT_DMA_SPREAD ? (i < NP && DMA >= 0) : (i == 0 && DMA >= 0)) __builtin_amdgcn_sched_group_barrier(0x020, GT_DMA_SPREAD ? 1 : NP, 0);
            if (i == NB - 1 && RA) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (i >= NB && RB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (i == 3 * NB - 1 && RA) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    static_assert(NK >= 4, "ring of four stages");
    issue(IC<0>{}); issue(IC<1>{}); issue(IC<2>{});
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * NP) : "memory");
    JLM_GT_T(2);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        rdA(0, st, 1); rdA(0, st, 0);
        if (st < BS) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) { rdB(0, st, nb, 0); rdB(0, st, nb, 1); }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // the epilogue's operands go out at k-step xg_at (1 + wave), behind that k-step's DMA request; the waits of the two
    // k-steps behind it leave them in flight (they are younger than the stage waited for), the third one retires them
    const int xg_at = 1 + wave;
    gate_for_each_ic([&](auto ktc) {
        constexpr int KT = decltype(ktc)::value;
        if constexpr (KT + 1 < NK) {
            constexpr int base = (KT + 2 < NK) ? NP : 0;
            if constexpr (ABL & 8) {
                if ((unsigned)(KT - 1 - xg_at) < 2u) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(base + NXG) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(base) : "memory");
            } else {
                if ((unsigned)(KT - 1 - xg_at) < 2u) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(base + NXG) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(base) : "memory");
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KT == NK - 4) JLM_GT_T(5);
        half(ktc, IC<0>{}, IC<(KT + 3 < NK) ? KT + 3 : -1>{});
        if (!(ABL & 16) && KT == xg_at) load_epilogue_operands();
        half(ktc, IC<1>{}, IC<-1>{});
    }, std::make_integer_sequence<int, NK>{});
    JLM_GT_T(3);
    wait_epilogue_operands();

    const float ks = a.descale * -1.4426950408889634f, kt = a.descale * -2.8853900817779268f;      // (jlm_common.h: exact for ds = 2^-S)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int g = eok[nb] ? eg[nb] : -1;
        if (g < 0) continue;
        f32x4 cn, hn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gi = jlm_sigmoid_k(acc[nb][e] + xg[nb][0][e], ks), gf = jlm_sigmoid_k(acc[nb][4 + e] + xg[nb][1][e], ks);
            const float go = jlm_sigmoid_k(acc[nb][8 + e] + xg[nb][2][e], ks), gg = jlm_tanh_k(acc[nb][12 + e] + xg[nb][3][e], kt);
            cn[e] = (ep[nb] >= 0 ? cp[nb][e] : 0.0f) * gf + gg * gi;
            hn[e] = jlm_tanh(cn[e]) * go;
        }
        *reinterpret_cast<f32x4 *>(a.c_out + (size_t)g * ld + u0) = cn;
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x4 hi4, lo4;
        jlm_split4(hn, a.h_scale, hi4, lo4);
        _Float16 *blk = reinterpret_cast<_Float16 *>(a.h_out + (size_t)g * ld + (u0 & ~7)) + (u0 & 7);
        *reinterpret_cast<f16x4 *>(blk) = hi4;
        *reinterpret_cast<f16x4 *>(blk + 8) = lo4;
        if (a.h_f32) *reinterpret_cast<f32x4 *>(a.h_f32 + (size_t)g * ld + u0) = hn;
    }
#ifdef JLM_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    JLM_GT_T(4);
#endif
}



// =====================================================================================================================
// Round 4: the same tile, PERSISTENT (gate_pu_kernel; H = 512, launches of two or more row tiles per CU).
//
// gate_xg_u16_kernel costs 21.5 us per 160 x 128 tile at any number of rows: 14.2 in its sixteen k-steps, the rest -- index
// chains 0.8, first ring stages 1.7, cell update 2.5, the gap until the next workgroup of the CU is running ~2.3 -- with the
// matrix pipe idle (profiles/r03_a_gate_timeline.txt, r03_a_gate_kbench_ab.txt).  Here a workgroup keeps its gate-column tile
// and walks the row tiles q, q + Q, ... (grid = 256: 16 column tiles x Q = 16 sequences) with the k-step loop of the kernel
// above, and the RING DOES NOT STOP at a tile's end: the last three k-steps of a tile request stages 0 .. 2 of the next one
// and read its first fragments, the row indices of the next tile's state pieces are fetched under the first k-steps, the
// epilogue's own indices of a tile under ITS first k-steps -- a tile's cell update is followed by the next tile's first MFMA.
// Every wait is a counted s_waitcnt vmcnt(n) derived at compile time from the issue order (pu_* functions, the scheme of
// csrc/jlm_gate_ws.hip); stores are issued by every lane (rows past the edge: a dump page) so that the count is exact.
constexpr int PU_ST = 3;                                  // stores per hypothesis block and lane that every launch issues (c, h hi, h lo)
__device__ float gate_pu_dump_page[4 * 512 + 8];
// -DJLM_PROFILE builds (tools/probes/gate_pu_profile.py): waves 0 and 4 stamp, for a workgroup's first eight tiles, the shader clock
// at the top of the tile's k-steps, behind them, when the epilogue operands are in, and behind the cell update; [4] the wall clock
#ifdef JLM_PROFILE
__device__ unsigned long long jlm_gate_pu_time[256][2][8][5];
#define JLM_PU_T(i) do { if ((threadIdx.x & 255) == 0 && tix < 8) { jlm_gate_pu_time[blockIdx.x & 255][threadIdx.x >> 8][tix][i] = clock64(); \
    if ((i) == 0) jlm_gate_pu_time[blockIdx.x & 255][threadIdx.x >> 8][tix][4] = wall_clock64(); } } while (0)
#else
#define JLM_PU_T(i) (void)0
#endif

// issue order of a tile's vector-memory operations, per wave type (NB hypothesis blocks, NP pieces per stage).  k-step j issues
//   D  the NP pieces of stage j + 3 (of the NEXT tile from j = 13 on);
//   X  j = 0: the epilogue's row ids (NB; first tile: prologue), j = 1: row ids of the next tile's state pieces (NP - 2), j = 2: prev /
//      word of the epilogue rows (2 NB; first tile: prologue), j = 3: prev of the next tile's piece rows (NP - 2), j = 5 + wave: the
//      epilogue operands (5 NB: table line quads + old cell state);
// and the cell update behind k-step 15 issues PU_ST stores per block.
constexpr int pu_x(bool first, int nb, int np, int j) {
    return j == 0 ? (first ? 0 : nb) : j == 1 ? np - 2 : j == 2 ? (first ? 0 : 2 * nb) : j == 3 ? np - 2 : 0;
}
constexpr int pu_issued(bool first, int nb, int np, int j) { return np + pu_x(first, nb, np, j); }
// operations issued behind the D of k-step j0 up to (not including) the top of k-step j1 of the same tile, the epilogue operands
// (5 NB loads at k-step 5 + wave, a run-time position: the waits that can see them exist in two forms) not counted
constexpr int pu_between(bool first, int nb, int np, int j0, int j1) {
    int n = pu_x(first, nb, np, j0);
    for (int j = j0 + 1; j < j1; ++j) n += pu_issued(first, nb, np, j);
    return n;
}
constexpr int pu_clamp(int n) { return n > 63 ? 63 : n; }
// top of k-step kt: stage kt + 1 has landed.  It was requested in k-step kt - 2 -- of the tile before for kt < 2 (the first tile: in
// the prologue, D0 D1 D2 back to back), with that tile's cell-update stores in between
constexpr int pu_top(bool first, int nb, int np, int kt) {
    if (kt >= 2) return pu_between(first, nb, np, kt - 2, kt);
    if (first) return np * (1 - kt) + (kt == 1 ? pu_issued(true, nb, np, 0) : 0);
    // (the tile before is never a "first" tile as far as k-steps 13 .. 15 go: their X is the same in both)
    int n = pu_x(false, nb, np, 14 + kt);
    for (int j = 15 + kt; j < 16; ++j) n += pu_issued(false, nb, np, j);
    n += nb * PU_ST;
    for (int j = 0; j < kt; ++j) n += pu_issued(false, nb, np, j);
    return n;
}
static_assert(pu_top(true, 3, 4, 0) == 4 && pu_top(true, 3, 4, 1) == 4 && pu_top(false, 3, 4, 5) == 2 + 4 && pu_top(false, 2, 5, 0) == 5 + 6 &&
              pu_top(false, 3, 4, 1) == 9 + 4 + 3, "issue-order bookkeeping");

template <int NB, int NP, int BS, bool HF32>
struct GatePu {
    static constexpr int NK = 16, S = GT_STAGES, NS = NP - 2;
    static constexpr int OOB_ROW = 0x7fffffff;
    const GateXgArgs &a;
    float *smem;
    int lane, wave, gb, hb0, li, hf, lrow, lslot, H, ld, M, Q, n0, u0, tiles_m, tm, xg_at;
    __amdgpu_buffer_rsrc_t rs_w, rs_h;
    float *dump;
    int goff[2][2], w_off, h_off;
    int eg[NB], ep[NB], ew[NB];
    bool eok[NB];
    int vidx[NP], vslot[NP], dst[NP];                     // record index of the piece's lane (row x records per row + swizzled slot), the slot alone
    int w16, h16;
    int rn[NS];
#ifdef JLM_PROFILE
    int tix = 0;
#endif
    bool nok[NS];
    f32x16 acc[NB];
    f16x8 A[2][2], B[BS][NB][2];
    f32x4 xg[NB][4], cp[NB];

    __device__ __forceinline__ GatePu(const GateXgArgs &a_, float *smem_) : a(a_), smem(smem_) {}
    __device__ __forceinline__ int piece_index(int i) const { return (NP == 4) ? 2 * wave + i : 8 + 3 * (wave - 4) + i; }

    template <int SG>
    __device__ __forceinline__ void issue() {
        float *base = smem + (SG & (S - 1)) * GT_STAGE_FLOATS;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (i < 2)
                __builtin_amdgcn_struct_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(base + dst[i]), 16, vidx[i], 0,
                                                            (SG & (NK - 1)) * 128, 0, 0);
            else
                __builtin_amdgcn_struct_ptr_buffer_load_lds(rs_h, (__attribute__((address_space(3))) void *)(base + dst[i]), 16, vidx[i], 0,
                                                            (SG & (NK - 1)) * 128, 0, 0);
        }
    }
    __device__ __forceinline__ void issue_at(int sg) {     // the prologue's stages (a run-time stage number: folded after inlining)
        float *base = smem + (sg & (S - 1)) * GT_STAGE_FLOATS;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (i < 2)
                __builtin_amdgcn_struct_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(base + dst[i]), 16, vidx[i], 0,
                                                            sg * 128, 0, 0);
            else
                __builtin_amdgcn_struct_ptr_buffer_load_lds(rs_h, (__attribute__((address_space(3))) void *)(base + dst[i]), 16, vidx[i], 0,
                                                            sg * 128, 0, 0);
        }
    }
    __device__ __forceinline__ void rdA(int stage, int st, int p) {
        A[st][p] = *reinterpret_cast<const f16x8 *>(smem + (stage & (S - 1)) * GT_STAGE_FLOATS + w_off + goff[st][p]);
    }
    __device__ __forceinline__ void rdB(int stage, int st, int nb, int p) {
        B[st % BS][nb][p] = *reinterpret_cast<const f16x8 *>(smem + (stage & (S - 1)) * GT_STAGE_FLOATS + h_off + nb * 1024 + goff[st][p]);
    }
    __device__ __forceinline__ void load_int(int &d, const int *src) { asm volatile("global_load_dword %0, %1, off" : "=&v"(d) : "v"(src) : "memory"); }
    template <int N>
    __device__ __forceinline__ void wait_int(int &v) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N) : "memory"); }
    __device__ __forceinline__ void load_ops() {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float *xrp = a.xg + (size_t)ew[nb] * (size_t)(4 * H) + n0 + 32 * gb + 4 * hf;
            const float *cpp = a.c_in + (size_t)(ep[nb] >= 0 ? ep[nb] : 0) * ld + u0;
            asm volatile("global_load_dwordx4 %0, %5, off\n\t"
                         "global_load_dwordx4 %1, %5, off offset:32\n\t"
                         "global_load_dwordx4 %2, %5, off offset:64\n\t"
                         "global_load_dwordx4 %3, %5, off offset:96\n\t"
                         "global_load_dwordx4 %4, %6, off"
                         : "=&v"(xg[nb][0]), "=&v"(xg[nb][1]), "=&v"(xg[nb][2]), "=&v"(xg[nb][3]), "=&v"(cp[nb])
                         : "v"(xrp), "v"(cpp) : "memory");
        }
    }
    template <int N>
    __device__ __forceinline__ void wait_ops() {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            asm volatile("s_waitcnt vmcnt(%5)" : "+v"(xg[nb][0]), "+v"(xg[nb][1]), "+v"(xg[nb][2]), "+v"(xg[nb][3]), "+v"(cp[nb]) : "n"(N) : "memory");
    }

    // one half step (k-step KT, half ST) of gate_xg_body_u: 3 NB MFMAs, the fragment registers refilled in place behind them -- A with
    // half ST of k-step KT + 1 (KT + 1 = 16: stage 0 of the next tile), B with the same (BS = 2) or with the NEXT half step (BS = 1)
    template <int KT, int ST, int DMA, bool ZERO>
    __device__ __forceinline__ void half() {
        constexpr int BKT = (BS == 2) ? KT + 1 : (ST == 0 ? KT : KT + 1), BST = (BS == 2) ? ST : 1 - ST;
        const f32x16 zf = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[ST][1], B[ST % BS][nb][0], ZERO ? zf : acc[nb], 0, 0, 0);
            if constexpr (DMA >= 0) { if (nb == 0) issue<DMA>(); }
        }
        rdA(KT + 1, ST, 1);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[ST][0], B[ST % BS][nb][0], acc[nb], 0, 0, 0);
            rdB(BKT, BST, nb, 0);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[ST][0], B[ST % BS][nb][1], acc[nb], 0, 0, 0);
            rdB(BKT, BST, nb, 1);
        }
        rdA(KT + 1, ST, 0);
#pragma unroll
        for (int i = 0; i < 3 * NB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (GT_DMA_SPREAD ? (i < NP && DMA >= 0) : (i == 0 && DMA >= 0)) __builtin_amdgcn_sched_group_barrier(0x020, GT_DMA_SPREAD ? 1 : NP, 0);
            if (i == NB - 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (i >= NB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (i == 3 * NB - 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    template <bool FIRST, int KT>
    __device__ __forceinline__ void kstep(int m0, int m0n, bool has_next) {
        // the epilogue operands (5 NB loads behind the D of k-step xg_at) are younger than the stage waited for when they went out in
        // k-step KT - 2 or KT - 1
        constexpr int base = pu_top(FIRST, NB, NP, KT);
        if (KT >= 2 && (unsigned)(KT - 1 - xg_at) < 2u) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(pu_clamp(base + 5 * NB)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(pu_clamp(base)) : "memory");
        if constexpr (KT == 0 && FIRST) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                rdA(0, st, 1); rdA(0, st, 0);
                if (st < BS) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) { rdB(0, st, nb, 0); rdB(0, st, nb, 1); }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KT == 13) {
            // in front of the next tile's first stage: its state pieces' predecessors (X of k-step 3); younger: the rest of that X is
            // this very group, then k-steps 4 .. 12 and, among them, the epilogue operands
#pragma unroll
            for (int i = 0; i < NS; ++i) wait_int<pu_clamp(9 * NP)>(rn[i]);
#pragma unroll
            for (int i = 0; i < NS; ++i) vidx[2 + i] = (nok[i] && rn[i] >= 0) ? rn[i] * h16 + vslot[2 + i] : OOB_ROW;
        }
        half<KT, 0, (KT + 3) & (NK - 1), KT == 0>();
        if constexpr (KT == 0 && !FIRST) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int r = m0 + 32 * (hb0 + nb) + li;
                eok[nb] = r < M;
                load_int(eg[nb], a.rows + (eok[nb] ? r : M - 1));
            }
        }
        if constexpr (KT == 1) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int r = m0n + 8 * piece_index(i) + lrow;
                nok[i] = has_next && r < M;
                load_int(rn[i], a.rows + (nok[i] ? r : M - 1));
            }
        }
        if constexpr (KT == 2 && !FIRST) {
            // behind D of k-step 2: the row ids (X of k-step 0); younger: k-step 1, D of k-step 2
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wait_int<pu_clamp(pu_issued(false, NB, NP, 1) + NP)>(eg[nb]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                load_int(ep[nb], a.prev + eg[nb]);
                load_int(ew[nb], a.word + eg[nb]);
            }
        }
        if constexpr (KT == 3) {
            // behind D of k-step 3: the next tile's piece row ids (X of k-step 1); younger: k-step 2, D of k-step 3
#pragma unroll
            for (int i = 0; i < NS; ++i) wait_int<pu_clamp(pu_issued(FIRST, NB, NP, 2) + NP)>(rn[i]);
#pragma unroll
            for (int i = 0; i < NS; ++i) load_int(rn[i], a.prev + rn[i]);
        }
        if constexpr (KT >= 5 && KT <= 12) {
            if (KT == xg_at) {
                if constexpr (!FIRST) {
                    // prev / word of the epilogue rows (X of k-step 2); younger: k-steps 3 .. KT - 1, D of k-step KT
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        wait_int<pu_clamp(pu_between(false, NB, NP, 2, KT) - 2 * NB + NP)>(ep[nb]);
                        wait_int<pu_clamp(pu_between(false, NB, NP, 2, KT) - 2 * NB + NP)>(ew[nb]);
                    }
                }
                load_ops();
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        half<KT, 1, -1, false>();
    }

    template <bool FIRST>
    __device__ __forceinline__ void tile(bool has_next) {
        const int m0 = tm * GT_BM, m0n = (tm + Q) * GT_BM;
        JLM_PU_T(0);
        gate_for_each_ic([&](auto ktc) { this->template kstep<FIRST, decltype(ktc)::value>(m0, m0n, has_next); },
                         std::make_integer_sequence<int, NK>{});
        JLM_PU_T(1);
        // the epilogue operands: requested at k-step xg_at <= 12 -- at least the three stages of k-steps 13 .. 15 are younger
        wait_ops<3 * NP>();
        JLM_PU_T(2);
        const float ks = a.descale * -1.4426950408889634f, kt = a.descale * -2.8853900817779268f;      // (jlm_common.h: exact for ds = 2^-S)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int g = eok[nb] ? eg[nb] : -1;
            f32x4 cn, hn;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gi = jlm_sigmoid_k(acc[nb][e] + xg[nb][0][e], ks), gf = jlm_sigmoid_k(acc[nb][4 + e] + xg[nb][1][e], ks);
                const float go = jlm_sigmoid_k(acc[nb][8 + e] + xg[nb][2][e], ks), gg = jlm_tanh_k(acc[nb][12 + e] + xg[nb][3][e], kt);
                cn[e] = (ep[nb] >= 0 ? cp[nb][e] : 0.0f) * gf + gg * gi;
                hn[e] = jlm_tanh(cn[e]) * go;
            }
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            f16x4 hi4, lo4;
            jlm_split4(hn, a.h_scale, hi4, lo4);
            float *crow = g >= 0 ? a.c_out + (size_t)g * ld + u0 : dump;
            *reinterpret_cast<f32x4 *>(crow) = cn;
            _Float16 *blk = g >= 0 ? reinterpret_cast<_Float16 *>(a.h_out + (size_t)g * ld + (u0 & ~7)) + (u0 & 7) : reinterpret_cast<_Float16 *>(dump);
            *reinterpret_cast<f16x4 *>(blk) = hi4;
            *reinterpret_cast<f16x4 *>(g >= 0 ? blk + 8 : blk + 4) = lo4;
            if constexpr (HF32) *reinterpret_cast<f32x4 *>(g >= 0 ? a.h_f32 + (size_t)g * ld + u0 : dump) = hn;
        }
        JLM_PU_T(3);
#ifdef JLM_PROFILE
        ++tix;
#endif
    }

    __device__ __forceinline__ void run(int wave_) {
        lane = threadIdx.x & 63; wave = wave_;
        gb = wave & 3; hb0 = (wave >> 2) ? 3 : 0;
        li = lane & 31; hf = lane >> 5; lrow = lane >> 3; lslot = lane & 7;
        H = a.H; ld = a.ld;
        // XCD x keeps tiles_n / 8 gate-column tiles (their slices of the gate matrix stay in its L2: every tile stages them again)
        const int b = blockIdx.x, cpx = a.tiles_n >> 3;
        const int x = b & 7, jb = b >> 3;
        const int tn = x * cpx + jb % cpx, q = jb / cpx;
        Q = ((int)gridDim.x >> 3) / cpx;
        M = a.ndev ? min(*a.ndev, a.nrows) : a.nrows;
        tiles_m = (M + GT_BM - 1) / GT_BM;
        if (q >= tiles_m) return;
        n0 = tn * GT_BN;
        u0 = (n0 >> 2) + 8 * gb + 4 * hf;
        rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wt), (short)16, 0x7ffffff0, 0x00020000);     // (16-byte records, one address
        rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.h), (short)16, 0x7ffffff0, 0x00020000);      //  register: see gate_xg_body_u)
        w16 = H >> 2; h16 = ld >> 2;
        dump = gate_pu_dump_page + 4 * (int)threadIdx.x;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int p = 0; p < 2; ++p) goff[st][p] = li * 32 + (((4 * st + 2 * hf + p) ^ ((li >> 1) & 7)) * 4);
        w_off = gb * 32 * 32;
        h_off = (GT_BN + hb0 * 32) * 32;
        tm = q;
        const int m0 = tm * GT_BM;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int r = m0 + 32 * (hb0 + nb) + li;
            eok[nb] = r < M;
            eg[nb] = a.rows[eok[nb] ? r : M - 1];
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int r = m0 + 8 * piece_index(i) + lrow;
            nok[i] = r < M;
            rn[i] = a.rows[nok[i] ? r : M - 1];
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { ep[nb] = a.prev[eg[nb]]; ew[nb] = a.word[eg[nb]]; }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (i < 2) {
                const int pr = 8 * (2 * wave + i), row = pr + lrow;
                vslot[i] = lslot ^ ((row >> 1) & 7);
                vidx[i] = (n0 + row) * w16 + vslot[i];
                dst[i] = pr * 32;
            } else {
                const int prow = 8 * piece_index(i - 2), row = prow + lrow;
                const int p = a.prev[rn[i - 2]];
                vslot[i] = lslot ^ ((row >> 1) & 7);
                vidx[i] = (nok[i - 2] && p >= 0) ? p * h16 + vslot[i] : OOB_ROW;
                dst[i] = (GT_BN + prow) * 32;
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { asm volatile("" : "+v"(eg[nb])); asm volatile("" : "+v"(ep[nb])); asm volatile("" : "+v"(ew[nb])); }
#pragma unroll
        for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(vidx[i]));
        issue_at(0); issue_at(1); issue_at(2);
        // the epilogue operands go out at k-step 5 + wave: the chip's table lines spread over eight k-steps instead of one burst
        xg_at = 5 + wave;
        tile<true>(tm + Q < tiles_m);
        tm += Q;
        for (; tm < tiles_m; tm += Q) tile<false>(tm + Q < tiles_m);
    }
};

template <bool HF32>
__global__ __launch_bounds__(512, 1) void gate_pu_kernel(GateXgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (wave < 4) { GatePu<3, 4, 1, HF32> k(a, smem); k.run(wave); }
    else { GatePu<2, 5, 2, HF32> k(a, smem); k.run(wave); }
}

__device__ __forceinline__ bool gate_tile_of_block(const GateXgArgs &a, int &m0, int &n0, int &M) {
    const int b = blockIdx.x;
    int tm, tn;
    if ((a.tiles_n & 7) == 0) {
        const int x = b & 7, j = b >> 3, cpx = a.tiles_n >> 3;
        tn = x * cpx + j % cpx;
        tm = j / cpx;
    } else {
        tm = b / a.tiles_n;
        tn = b % a.tiles_n;
    }
    if (tm >= a.tiles_m) return false;
    M = a.ndev ? min(*a.ndev, a.nrows) : a.nrows;
    m0 = tm * GT_BM; n0 = tn * GT_BN;
    return m0 < M;
}

// H = 512 (16 k-steps), the refill-in-place pipeline
template <int ABL>
__global__ __launch_bounds__(512, 1) void gate_xg_u16_kernel(GateXgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int m0, n0, M;
    if (!gate_tile_of_block(a, m0, n0, M)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (wave < 4) gate_xg_body_u<3, 4, 16, 1, ABL>(a, m0, n0, M, wave, lane, smem);
    else gate_xg_body_u<2, 5, 16, 2, ABL>(a, m0, n0, M, wave, lane, smem);
}

__global__ __launch_bounds__(512, 1) void gate_xg_kernel(GateXgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware tile order (block b runs on XCD b % 8, observed; speed only): when the gate tiles divide over the 8
    // XCDs each XCD keeps tiles_n / 8 of them -- its slice of the gate matrix stays in its L2 -- and walks the row tiles
    const int b = blockIdx.x;
    int tm, tn;
    if ((a.tiles_n & 7) == 0) {
        const int x = b & 7, j = b >> 3, cpx = a.tiles_n >> 3;
        tn = x * cpx + j % cpx;
        tm = j / cpx;
    } else {
        tm = b / a.tiles_n;
        tn = b % a.tiles_n;
    }
    if (tm >= a.tiles_m) return;
    const int M = a.ndev ? min(*a.ndev, a.nrows) : a.nrows;
    const int m0 = tm * GT_BM, n0 = tn * GT_BN;
    if (m0 >= M) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (wave < 4) gate_xg_body<3, 4>(a, m0, n0, M, wave, lane, smem);
    else gate_xg_body<2, 5>(a, m0, n0, M, wave, lane, smem);
}

}  // namespace

#ifdef JLM_PROFILE
extern "C" int jlm_prof_read_gate_pu(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(jlm_gate_pu_time), sizeof(jlm_gate_pu_time)) == hipSuccess ? 0 : -1;
}
#endif

// the kernel of the refill-in-place pipeline; measurement builds (-DJLM_GATE_ABLATE) carry ablated copies, picked by JLM_GATE_ABL
static std::vector<const void *> gate_u16_variants() {
    std::vector<const void *> v{reinterpret_cast<const void *>(gate_xg_u16_kernel<0>)};
#ifdef JLM_GATE_ABLATE
#define GV(n) v.push_back(reinterpret_cast<const void *>(gate_xg_u16_kernel<n>));
    GV(1) GV(2) GV(4) GV(8) GV(6) GV(14) GV(5) GV(16) GV(22)
#undef GV
#endif
    return v;
}
static const void *gate_u16_selected() {
#ifdef JLM_GATE_ABLATE
    static const int abl = getenv("JLM_GATE_ABL") ? atoi(getenv("JLM_GATE_ABL")) : 0;
    static const int ids[] = {0, 1, 2, 4, 8, 6, 14, 5, 16, 22};
    static const std::vector<const void *> v = gate_u16_variants();
    for (size_t i = 0; i < v.size(); ++i) if (ids[i] == abl) return v[i];
#endif
    return reinterpret_cast<const void *>(gate_xg_u16_kernel<0>);
}

extern "C" int jlm_lstm_step_xg(const void *h_in, const float *c_in, int ld_state, void *h_out, float *c_out, const int *rows,
                                const int *prev, const int *word, const void *wt8, const float *xgate8, int H, float descale,
                                float h_scale, float *h_f32_out, int n_rows_max, const int *n_dev, void *stream) {
    if (H <= 0 || H % 32 != 0 || ld_state % 16 != 0 || ld_state < H) return -1;
    if (n_rows_max <= 0) return 0;
    static JlmLdsGrant grant_base, grant_u16[16];
    if (int rc = jlm_grant_lds(grant_base, reinterpret_cast<const void *>(gate_xg_kernel), GT_LDS_BYTES)) return rc;
    {
        int vi = 0;
        for (const void *k : gate_u16_variants())
            if (int rc = jlm_grant_lds(grant_u16[vi++ & 15], k, GT_LDS_BYTES)) return rc;
    }
    GateXgArgs a;
    a.h = reinterpret_cast<const float *>(h_in); a.c_in = c_in; a.h_out = reinterpret_cast<float *>(h_out); a.c_out = c_out;
    a.h_f32 = h_f32_out;
    a.ld = ld_state; a.rows = rows; a.prev = prev; a.word = word;
    a.wt = reinterpret_cast<const float *>(wt8); a.xg = xgate8; a.H = H; a.descale = descale; a.h_scale = h_scale;
    a.nrows = n_rows_max; a.ndev = n_dev;
    a.tiles_m = (n_rows_max + GT_BM - 1) / GT_BM;
    a.tiles_n = 4 * H / GT_BN;
    a.cx = 0;
    // JLM_GATE_V: 3 the persistent form of the one-tile kernel (gate_pu_kernel above); 2 the W-stationary persistent kernel
    // (csrc/jlm_gate_ws.hip) -- both for H = 512, a row list, gate-column tiles divisible over the 8 XCDs; 1 one tile per workgroup,
    // refill-in-place pipeline (round 3); 0 the round-2 loop (every H).  Default by the launch's row bound (tools/gpu_gate_pu.sh,
    // tools/gpu_gate_ws.sh; us per launch for 1 / 3 / 2 on one box: 2 560 rows 23.2 / 24.1 / 26.4, 5 120: 42.1 / 40.5 / 43.1,
    // 10 240: 82.7 / 75.0 / 80.2, 20 480: 174 / 160 / 156): one tile per CU -> 1; two to a few tiles per CU -> 3; more -> 2
    // (JLM_GATE_V = 1 / 2 / 3 forces one of the three H = 512 forms for A/B; the round-2 loop serves the other H only -- forcing it
    //  at H = 512, JLM_GATE_V=0, left with ABI 9)
    static const int variant_env = getenv("JLM_GATE_V") ? atoi(getenv("JLM_GATE_V")) : -1;
    const int variant = variant_env >= 1 ? variant_env : (n_rows_max >= 16384 ? 2 : n_rows_max >= 4096 ? 3 : 1);
    static const int ws_l = getenv("JLM_GATE_WS_L") ? atoi(getenv("JLM_GATE_WS_L")) : 3;
    if (variant == 4 && H == 512 && rows && !h_f32_out) {
        // round 6: 128 x 256 tiles, a 2 x 2 register block per wave, persistent (csrc/jlm_gate_p2.hip)
        if (int rc = jlm_gate::p2_launch(a, (hipStream_t)stream)) return rc;
    } else if (variant == 3 && H == 512 && rows && (a.tiles_n & 7) == 0) {
        // the persistent form of the one-tile kernel (gate_pu_kernel): tiles_n column tiles x Q row-tile sequences
        const int per_col = 256 / a.tiles_n;
        const int Q = a.tiles_m < per_col ? a.tiles_m : per_col;
        const bool hf32 = h_f32_out != nullptr;
        const void *fn = hf32 ? reinterpret_cast<const void *>(gate_pu_kernel<true>) : reinterpret_cast<const void *>(gate_pu_kernel<false>);
        static JlmLdsGrant grant_pu[2];
        if (int rc = jlm_grant_lds(grant_pu[hf32 ? 1 : 0], fn, GT_LDS_BYTES)) return rc;
        void *params[] = {&a};
        hipError_t e = hipLaunchKernel(fn, dim3(a.tiles_n * Q), dim3(512), params, GT_LDS_BYTES, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    } else if (variant == 2 && H == 512 && rows && (a.tiles_n & 7) == 0) {
        const int per_col = 256 / a.tiles_n;                       // row-tile sequences per gate-column tile: one resident workgroup per CU
        const int Q = a.tiles_m < per_col ? a.tiles_m : per_col;
        // gate-column tiles per XCD: fabric bytes ~ (16 / cx) x rows x 2 KB of state + (cx / 16) x 8 XCDs x 4 MB of gate matrix
        static const int ws_cx = getenv("JLM_GATE_WS_CX") ? atoi(getenv("JLM_GATE_WS_CX")) : 0;
        a.cx = ws_cx ? ws_cx : 4;                                  // (measured 2 / 4 / 8 / 16 at 10 240 and 20 480 rows: 4 by 2-4 %)
        if (a.cx > a.tiles_n || a.tiles_n % a.cx || 32 % a.cx) a.cx = 2;
        if (int rc = jlm_gate::ws_launch(a, ws_l == 3 ? 3 : 7, Q, (hipStream_t)stream)) return rc;
    } else if (variant >= 1 && H == 512) {
        void *params[] = {&a};
        hipError_t e = hipLaunchKernel(gate_u16_selected(), dim3(a.tiles_m * a.tiles_n), dim3(512), params, GT_LDS_BYTES, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    } else
        hipLaunchKernelGGL(gate_xg_kernel, dim3(a.tiles_m * a.tiles_n), dim3(512), GT_LDS_BYTES, (hipStream_t)stream, a);
    JLM_LAUNCH_CHECK();
    return 0;
}
