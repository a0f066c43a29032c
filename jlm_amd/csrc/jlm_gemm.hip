// f32 MFMA "NT" GEMM family for gfx950 (MI355X).
//
//   C[m][n] = sum_k Arow(m)[k] * Brow(n)[k]        both operands K-contiguous
//
// One mainloop, three epilogues:
//   EpiStore  plain store + column bias            (K4, V_table projections, debug logits)
//   EpiGate   LSTM gate nonlinearity + state update (K2+K3; operand A is the
//             gathered [h_prev | embedding] row, K1+K9 fused into the tile load)
//   EpiLse    per-tile online log-sum-exp          (K5+K6; logits never stored)
//
// Precision: the reference computes in float64 over float32 weights and the
// parity bar is <=1e-4 relative on logits, so the contraction runs on the
// exact-f32 matrix pipe (v_mfma_f32_32x32x2_f32, 157 TFLOP/s dense peak), not
// on single-pass bf16/fp16 MFMA (which misses the bar at K~768).
//
// Tiling: block = WAVES_M x WAVES_N waves, wave tile = (MT*32) x (NT*32),
// BK = 32 floats per k-step, tiles staged global -> LDS by the DMA one k-step
// ahead of the MFMAs (see "mainloop, LDS-DMA staging" below); two LDS buffers,
// one barrier per k-step.  f32 MFMA retires 4096 FLOP per 64 cycles per SIMD, so
// a 64x64 wave tile spends 4096 cycles per k-step against 16 ds_read_b128 and 8
// DMA issues: the kernel is MFMA-issue bound by design.
//
// MFMA operand mapping (32x32x2): lane l supplies A[i = l&31][kk = l>>5] and
// B[kk = l>>5][j = l&31].  Within a k-step lane-half h owns k = 16h .. 16h+15
// (one ds_read_b128 per 4 k), so MFMA (q, e) contracts k = 4q+e and 16+4q+e;
// A and B use the same assignment, hence the sum over k is complete.
// Accumulator: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#include "jlm_common.h"

// Per-workgroup timeline of the tile GEMMs (-DJLM_PROFILE builds only, tools/probes/gate_profile.py): thread 0 of a
// workgroup stamps the 100 MHz wall clock at kernel start, after the prologue, after the main loop and at the end.
#ifdef JLM_PROFILE
static __device__ unsigned long long jlm_wg_time[4096][4];
#define JLM_WG_T(i) do { if (threadIdx.x == 0) jlm_wg_time[blockIdx.x & 4095][i] = wall_clock64(); } while (0)
extern "C" int jlm_prof_read_wg_gemm(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(jlm_wg_time), sizeof(jlm_wg_time)) == hipSuccess ? 0 : -1;
}
#else
#define JLM_WG_T(i) (void)0
#endif
#include <stdlib.h>

#define BK 32

template <int WM_, int WN_, int MT_, int NT_>
struct TileCfg {
    static constexpr int WAVES_M = WM_, WAVES_N = WN_, MT = MT_, NT = NT_;
    static constexpr int BM = WM_ * MT_ * 32;
    static constexpr int BN = WN_ * NT_ * 32;
    static constexpr int NTHREADS = WM_ * WN_ * 64;
};

// ---------------------------------------------------------------- row sources
// Row sources hand out an always-dereferenceable pointer plus a validity flag: the
// tile loads are then unconditional (a select zeroes invalid data afterwards) and the
// compiler does not wrap every load in its own branch.
struct PlainRows {
    const float *base;
    const int *map;      // row -> storage row (NULL = identity, <0 = zero row)
    int ld;
    int nrows;           // static upper bound
    const int *ndev;     // optional device-side count
    struct St { const float *p; bool ok; };
    __device__ int count() const { return ndev ? min(*ndev, nrows) : nrows; }
    __device__ St init(int row, int n) const {
        St s; s.p = base; s.ok = false;
        if (row < n) {
            int r = map ? map[row] : row;
            if (r >= 0) { s.p = base + (size_t)r * ld; s.ok = true; }
        }
        return s;
    }
    __device__ const float *ptr(const St &s, int) const { return s.p; }
    __device__ bool valid(const St &s, int) const { return s.ok; }
};

// A operand of the gate GEMM: row r -> g = rows[r]; [ h[prev[g]] (k < H) | emb[word[g]] (k >= H) ]
struct GateRows {
    const float *h; int ldh;
    const int *rows; const int *prev; const int *word;
    const float *emb; int lde;
    int H;
    int nrows; const int *ndev;
    struct St { const float *ph; const float *pe; bool okh, oke; };
    __device__ int count() const { return ndev ? min(*ndev, nrows) : nrows; }
    __device__ St init(int row, int n) const {
        St s; s.ph = h; s.pe = emb; s.okh = false; s.oke = false;
        if (row < n) {
            int g = rows ? rows[row] : row;
            int p = prev[g];
            if (p >= 0) { s.ph = h + (size_t)p * ldh; s.okh = true; }
            s.pe = emb + (size_t)word[g] * lde;
            s.oke = true;
        }
        return s;
    }
    // k-steps never straddle H (H % 32 == 0): a step reads either the state or the embedding
    __device__ const float *ptr(const St &s, int k0) const { return k0 < H ? s.ph : s.pe - H; }
    __device__ bool valid(const St &s, int k0) const { return k0 < H ? s.okh : s.oke; }
};

// ------------------------------------------------------------------ tile map
// XCD-aware mode: block b runs on XCD b%8 (observed placement, speed only).  All
// N tiles of one M tile are walked consecutively by ONE XCD, so the M-side panel
// (the vocabulary block, the big operand) is fetched from HBM once and re-read
// from that XCD's L2; the N-side operand is small and lives in every L2.
struct TileMap {
    int tiles_m, tiles_n, xcd;
    __device__ bool get(int b, int &tm, int &tn) const {
        if (xcd == 0) { tm = b / tiles_n; tn = b % tiles_n; return tm < tiles_m; }
        int x = b & 7, j = b >> 3;
        if (xcd == 1) {                       // each XCD owns every 8th M tile, walks all N tiles of it
            tm = (j / tiles_n) * 8 + x; tn = j % tiles_n;
        } else {                              // xcd == 2: each XCD owns tiles_n/8 N tiles (weights stay in
            int cpx = tiles_n >> 3;           // its L2), walks the M tiles; the M-side tile is re-read by
            tn = x * cpx + j % cpx;           // the cpx consecutive workgroups of the same XCD
            tm = j / cpx;
        }
        return tm < tiles_m;
    }
    int grid() const { return xcd == 1 ? ((tiles_m + 7) / 8) * 8 * tiles_n : tiles_m * tiles_n; }
};

// ------------------------------------------------------------------ epilogues
struct EpiStore {
    float *C; const int *c_map; int ldc; const float *bias;
    float scale = 1.0f;                    // split-f16 mainloop: 2^-(eA + eB); 1 (exact) for the f32 mainloop
    // The storage rows of the lane's 16 x MT output rows, loaded at kernel start: behind the k-loop the index load is a
    // memory round trip in front of every store (the T projection's epilogue: 3.3 of its 10 us, tools/probes/tproj_profile.py).
    template <class Cfg> struct Pre { int r[Cfg::MT][16]; };
    template <class Cfg> __device__ void prepare(Pre<Cfg> &pre, int m0, int M) const {
        const int lane = threadIdx.x & 63, wm = (int)(threadIdx.x >> 6) / Cfg::WAVES_N;
#pragma unroll
        for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = m0 + (wm * Cfg::MT + mt) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                pre.r[mt][reg] = row < M ? (c_map ? c_map[row] : row) : -1;
            }
    }
    template <class Cfg>
    __device__ void run(f32x16 (&acc)[Cfg::MT][Cfg::NT], int m0, int n0, int wm, int wn, int lane,
                        int M, int N, float *, const Pre<Cfg> &pre) const {
#pragma unroll
        for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = pre.r[mt][reg];
                if (r < 0) continue;
                float *crow = C + (size_t)r * ldc;
#pragma unroll
                for (int nt = 0; nt < Cfg::NT; ++nt) {
                    int col = n0 + (wn * Cfg::NT + nt) * 32 + (lane & 31);
                    if (col < N) crow[col] = acc[mt][nt][reg] * scale + (bias ? bias[col] : 0.0f);
                }
            }
    }
};

// Gate epilogue for 64 x 64 tiles.  The 64 columns of a tile are the i, f, o, g
// pre-activations of the same 16 hidden units (packed weight layout,
// include/jlm_hip.h).  The accumulators go through LDS once so that one thread
// owns (row, 4 consecutive units) with all four gates: the LSTM cell update is
// then 4-wide, and c_prev / c' / h' move as 16-byte accesses.  Small tiles on
// purpose: at R = 2560 rows the gate GEMM is 40 x 32 = 1280 workgroups = exactly
// 5 per CU; 128 x 128 tiles (320 workgroups on 256 CUs) cap MFMA use at 62.5 %.
struct EpiGate {
    const float *c_in; float *h_out; float *c_out; int ld;
    const int *rows; const int *prev; const float *bias;
    float scale = 1.0f;                    // split-f16 mainloop: 2^-S; 1 (exact) for the f32 mainloop
    float *h_split = nullptr;              // optional: h' also (or only, h_out == NULL) as split rows, stride ld
    float h_scale = 1.0f;                  //           scaled by this power of two
    const float *xgate = nullptr;          // optional [V, 4H] table  emb . W_x^T + bias  (packed column order):
    const int *word = nullptr;             //   the GEMM then contracts over the state only and bias is unused
    int ld_gate = 0;                       //   = H (a table row has 4 H floats)
    // Row indices of the thread's epilogue rows, loaded at kernel start: the chain rows[] -> prev[] / word[] is two
    // dependent global loads, which otherwise sit between the last MFMA and the first gate (once per 32/64-row pass).
    template <class Cfg> struct Pre {
        static constexpr int RPP = 256 / (Cfg::BN / 16);
        static constexpr int N = Cfg::BM / RPP;
        int g[N], p[N], w[N];
    };
    template <class Cfg> __device__ void prepare(Pre<Cfg> &pre, int m0, int M) const {
        constexpr int TPR = Cfg::BN / 16;
#pragma unroll
        for (int i = 0; i < Pre<Cfg>::N; ++i) {
            const int row = m0 + i * Pre<Cfg>::RPP + (int)threadIdx.x / TPR;
            int g = -1, p = -1, w = 0;
            if (row < M) {
                g = rows ? rows[row] : row;
                p = prev[g];
                if (xgate) w = word[g];
            }
            pre.g[i] = g; pre.p[i] = p; pre.w[i] = w;
        }
    }
    template <class Cfg>
    __device__ void run(f32x16 (&acc)[Cfg::MT][Cfg::NT], int m0, int n0, int wm, int wn, int lane,
                        int M, int, float *smem, const Pre<Cfg> &pre) const {
        constexpr int BN = Cfg::BN, NT = Cfg::NT;
        // STRIP: one row of four waves, each owning a 32-column strip of all MT row blocks (the one-tile-per-CU
        // form of the LSTM step, TileCfg<1, 4, MT, 1>): the tile passes through LDS 32 rows at a time
        constexpr bool STRIP = Cfg::WAVES_M == 1 && Cfg::WAVES_N == 4 && NT == 1;
        static_assert((BN == 64 || BN == 128) && Cfg::NTHREADS == 256 &&
                          (STRIP ? BN == 128 : (Cfg::WAVES_N == 2 && Cfg::BM % 64 == 0)),
                      "gate epilogue: tiles of 64-column groups (4 gates x 16 units), 64 (strip form: 32) rows at a time");
        constexpr int PR = STRIP ? 32 : 64;    // rows of the tile per pass
        constexpr int CT_LD = BN + 4;          // transposition buffer [PR][CT_LD]
        constexpr int TPR = BN / 16;           // threads per row, 4 units (x 4 gates) each
        constexpr int RPP = 256 / TPR;         // rows per pass
        float *ct = smem;
        const int tid = threadIdx.x;
        const int q = tid % TPR;
        const int cb = (q >> 2) * 64 + (q & 3) * 4;          // column of the thread's i-gate quad inside the tile
        const int u0 = (n0 >> 2) + (q >> 2) * 16 + (q & 3) * 4;
        f32x4 bi = {0.f, 0.f, 0.f, 0.f}, bf = bi, bo = bi, bg = bi;
        if (!xgate) {
            bi = *reinterpret_cast<const f32x4 *>(bias + n0 + cb);
            bf = *reinterpret_cast<const f32x4 *>(bias + n0 + cb + 16);
            bo = *reinterpret_cast<const f32x4 *>(bias + n0 + cb + 32);
            bg = *reinterpret_cast<const f32x4 *>(bias + n0 + cb + 48);
        }
        // Strip form (one workgroup per CU, nobody else to hide behind): the table rows and old cell states of ALL the
        // thread's rows are requested up front, so the passes below wait for one memory round trip, not one each.
        constexpr int NPRE = Pre<Cfg>::N;
        constexpr bool PF_ALL = STRIP;
        f32x4 xi[PF_ALL ? NPRE : 1], xf[PF_ALL ? NPRE : 1], xo[PF_ALL ? NPRE : 1], xg[PF_ALL ? NPRE : 1], cpv[PF_ALL ? NPRE : 1];
        if (PF_ALL) {
#pragma unroll
            for (int pi = 0; pi < NPRE; ++pi) {
                xi[pi] = bi; xf[pi] = bf; xo[pi] = bo; xg[pi] = bg;
                cpv[pi] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (pre.g[pi] < 0) continue;
                if (xgate) {
                    const float *xr = xgate + (size_t)pre.w[pi] * (size_t)(4 * ld_gate) + n0 + cb;
                    xi[pi] = *reinterpret_cast<const f32x4 *>(xr);
                    xf[pi] = *reinterpret_cast<const f32x4 *>(xr + 16);
                    xo[pi] = *reinterpret_cast<const f32x4 *>(xr + 32);
                    xg[pi] = *reinterpret_cast<const f32x4 *>(xr + 48);
                }
                if (pre.p[pi] >= 0) cpv[pi] = *reinterpret_cast<const f32x4 *>(c_in + (size_t)pre.p[pi] * ld + u0);
            }
        }
#pragma unroll
        for (int rh = 0; rh < Cfg::BM / PR; ++rh) {
            if (rh > 0) __syncthreads();       // the previous rows have been consumed
#pragma unroll
            for (int mt = 0; mt < Cfg::MT; ++mt) {
                const int blk = wm * Cfg::MT + mt;           // 32-row block of the tile held by this wave
                if (blk / (PR / 32) != rh) continue;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int r = (blk % (PR / 32)) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                        ct[r * CT_LD + (wn * NT + nt) * 32 + (lane & 31)] = acc[mt][nt][reg] * scale;
                    }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < PR / RPP; ++it) {
                const int r = tid / TPR + it * RPP;
                const int row = m0 + rh * PR + r;
                if (row >= M) continue;
                const int pi = rh * (PR / RPP) + it;
                const int g = pre.g[pi];
                const int p = pre.p[pi];
                if (PF_ALL) {
                    bi = xi[pi]; bf = xf[pi]; bo = xo[pi]; bg = xg[pi];
                } else if (xgate) {        // 4 x 16 B of the word's precomputed input-side pre-activations
                    const float *xr = xgate + (size_t)pre.w[pi] * (size_t)(4 * ld_gate) + n0 + cb;
                    bi = *reinterpret_cast<const f32x4 *>(xr);
                    bf = *reinterpret_cast<const f32x4 *>(xr + 16);
                    bo = *reinterpret_cast<const f32x4 *>(xr + 32);
                    bg = *reinterpret_cast<const f32x4 *>(xr + 48);
                }
                const f32x4 zi = *reinterpret_cast<const f32x4 *>(ct + r * CT_LD + cb) + bi;
                const f32x4 zf = *reinterpret_cast<const f32x4 *>(ct + r * CT_LD + cb + 16) + bf;
                const f32x4 zo = *reinterpret_cast<const f32x4 *>(ct + r * CT_LD + cb + 32) + bo;
                const f32x4 zg = *reinterpret_cast<const f32x4 *>(ct + r * CT_LD + cb + 48) + bg;
                f32x4 cp = {0.f, 0.f, 0.f, 0.f};
                if (PF_ALL) cp = cpv[pi];
                else if (p >= 0) cp = *reinterpret_cast<const f32x4 *>(c_in + (size_t)p * ld + u0);
                f32x4 cn, hn;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gi = jlm_sigmoid(zi[e]), gf = jlm_sigmoid(zf[e]), go = jlm_sigmoid(zo[e]);
                    const float gg = jlm_tanh(zg[e]);
                    cn[e] = cp[e] * gf + gg * gi;
                    hn[e] = jlm_tanh(cn[e]) * go;
                }
                *reinterpret_cast<f32x4 *>(c_out + (size_t)g * ld + u0) = cn;
                if (h_out) *reinterpret_cast<f32x4 *>(h_out + (size_t)g * ld + u0) = hn;
                if (h_split) {             // units u0 .. u0+3 = half of an 8-value block [8 x f16 hi][8 x f16 lo]
                    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                    f16x4 hi4, lo4;
                    jlm_split4(hn, h_scale, hi4, lo4);
                    _Float16 *blk = reinterpret_cast<_Float16 *>(h_split + (size_t)g * ld + (u0 & ~7)) + (u0 & 7);
                    *reinterpret_cast<f16x4 *>(blk) = hi4;
                    *reinterpret_cast<f16x4 *>(blk + 8) = lo4;
                }
            }
        }
    }
};

// LSE epilogue.  GEMM M = vocabulary rows, N = hypothesis rows ("swapped"
// orientation): a lane then holds 16*MT vocabulary logits of ONE hypothesis in
// registers, so max / sum-exp are in-lane chains plus one cross-half exchange
// and one 2-wave combine through LDS.
struct EpiLse {
    const float *bias;     // per vocabulary row of this segment
    float *part;           // float2 [tiles][ld_part]
    int ld_part, tile0;
    float scale = 1.0f;    // split-f16 mainloop: 2^-(eA + eB); 1 (exact) for the f32 mainloop
    template <class Cfg> struct Pre {};
    template <class Cfg> __device__ void prepare(Pre<Cfg> &, int, int) const {}
    template <class Cfg>
    __device__ void run(f32x16 (&acc)[Cfg::MT][Cfg::NT], int m0, int n0, int wm, int wn, int lane,
                        int M, int N, float *smem, const Pre<Cfg> &) const {
        float2 *red = reinterpret_cast<float2 *>(smem);      // [WAVES_M][BN]
#pragma unroll
        for (int nt = 0; nt < Cfg::NT; ++nt) {
            float m = JLM_NEG_BIG;
            float v[Cfg::MT][16];
#pragma unroll
            for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    int row = m0 + (wm * Cfg::MT + mt) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                    bool ok = row < M;
                    float x = ok ? fmaf(acc[mt][nt][reg], scale, bias[row]) : JLM_NEG_BIG;
                    v[mt][reg] = x;
                    m = fmaxf(m, x);
                }
            float s = 0.0f;
#pragma unroll
            for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    s += (v[mt][reg] > JLM_NEG_BIG) ? expf(v[mt][reg] - m) : 0.0f;
            float m2 = __shfl_xor(m, 32), s2 = __shfl_xor(s, 32);
            lse_merge(m, s, m2, s2);
            if (lane < 32) red[wm * Cfg::BN + (wn * Cfg::NT + nt) * 32 + lane] = make_float2(m, s);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < Cfg::BN; c += Cfg::NTHREADS) {
            float2 a = red[c];
#pragma unroll
            for (int w = 1; w < Cfg::WAVES_M; ++w) {
                float2 b = red[w * Cfg::BN + c];
                lse_merge(a.x, a.y, b.x, b.y);
            }
            int col = n0 + c;
            if (col < N) reinterpret_cast<float2 *>(part)[(size_t)(tile0 + m0 / Cfg::BM) * ld_part + col] = a;
        }
    }
};

// ------------------------------------------------------ mainloop, LDS-DMA staging
// Same tiling and MFMA mapping as above, but the tiles go global -> LDS directly
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass, nothing for the wave
// to wait on until the end of the k-step).  The DMA writes lane-linear (wave base +
// 16 B x lane), so an operand tile is stored UNPADDED as [row][8 x 16 B] and the bank
// spread comes from a swizzle applied to the per-lane SOURCE address and, identically,
// to the fragment read: 16-B slot = chunk ^ ((row >> 1) & 7).  For ds_read_b128 (64
// banks, 16-lane service groups) rows 2j, 2j+1 then land on slots c^j and 8+(c^j):
// 16 distinct slots, conflict free, for every service group of a 32-row fragment.
// Rows past the edge / chunks past K read a zero page instead of being masked.
__device__ float jlm_zero_page[64];

#define GLDS16(gp, lp)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp),      \
                                     (__attribute__((address_space(3))) void *)(lp), 16, 0, 0)

template <class Cfg>
struct Lds2 {
    static constexpr int NW = Cfg::NTHREADS / 64;
    static constexpr int A_INST = Cfg::BM / 8 / NW;
    static constexpr int B_INST = Cfg::BN / 8 / NW;
    static constexpr int BYTES = 2 * (Cfg::BM + Cfg::BN) * 32 * 4;
    static_assert(Cfg::BM % (8 * NW) == 0 && Cfg::BN % (8 * NW) == 0, "tile rows must split over the waves");
};

template <class Cfg, class ARows, class BRows, class Epi>
__global__ __launch_bounds__(Cfg::NTHREADS) void gemm2_kernel(ARows A, BRows B, int K, Epi epi, TileMap tmap) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = Cfg::BM, BN = Cfg::BN, MT = Cfg::MT, NT = Cfg::NT;
    using L2 = Lds2<Cfg>;
    int tile_m, tile_n;
    if (!tmap.get(blockIdx.x, tile_m, tile_n)) return;
    const int M = A.count(), N = B.count();
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    if (m0 >= M || n0 >= N) return;
    typename Epi::template Pre<Cfg> pre;
    epi.template prepare<Cfg>(pre, m0, M);      // index loads of the epilogue, in flight under the mainloop
    float *As = smem;                         // [2][BM][32]
    float *Bs = smem + 2 * BM * 32;           // [2][BN][32]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
    const int lrow = lane >> 3, lslot = lane & 7;

    typename ARows::St sa[L2::A_INST];
    typename BRows::St sb[L2::B_INST];
    int ka[L2::A_INST], kb[L2::B_INST];       // source chunk (in floats) of this lane after the swizzle
#pragma unroll
    for (int i = 0; i < L2::A_INST; ++i) {
        const int r = (wave * L2::A_INST + i) * 8 + lrow;
        sa[i] = A.init(m0 + r, M);
        ka[i] = (lslot ^ ((r >> 1) & 7)) * 4;
    }
#pragma unroll
    for (int i = 0; i < L2::B_INST; ++i) {
        const int r = (wave * L2::B_INST + i) * 8 + lrow;
        sb[i] = B.init(n0 + r, N);
        kb[i] = (lslot ^ ((r >> 1) & 7)) * 4;
    }
    auto issue = [&](int k0, int buf) {
#pragma unroll
        for (int i = 0; i < L2::A_INST; ++i) {
            const int k = k0 + ka[i];
            const float *p = (k < K && A.valid(sa[i], k0)) ? A.ptr(sa[i], k0) + k : jlm_zero_page;
            GLDS16(p, As + (buf * BM + (wave * L2::A_INST + i) * 8) * 32);
        }
#pragma unroll
        for (int i = 0; i < L2::B_INST; ++i) {
            const int k = k0 + kb[i];
            const float *p = (k < K && B.valid(sb[i], k0)) ? B.ptr(sb[i], k0) + k : jlm_zero_page;
            GLDS16(p, Bs + (buf * BN + (wave * L2::B_INST + i) * 8) * 32);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    const int nk = (K + BK - 1) / BK;
    const int li = lane & 31, h = lane >> 5;
    int qoff[4];                               // float offset of quad q inside the lane's row
#pragma unroll
    for (int q = 0; q < 4; ++q) qoff[q] = li * 32 + (((h * 4 + q) ^ ((li >> 1) & 7)) * 4);
    issue(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue((kt + 1) * BK, cur ^ 1);
        const float *as = As + (cur * BM + wm * MT * 32) * 32;
        const float *bs = Bs + (cur * BN + wn * NT * 32) * 32;
        f32x4 a[2][MT], b[2][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[0][mt] = *reinterpret_cast<const f32x4 *>(as + mt * 1024 + qoff[0]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[0][nt] = *reinterpret_cast<const f32x4 *>(bs + nt * 1024 + qoff[0]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < 3) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    a[(q + 1) & 1][mt] = *reinterpret_cast<const f32x4 *>(as + mt * 1024 + qoff[q + 1]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    b[(q + 1) & 1][nt] = *reinterpret_cast<const f32x4 *>(bs + nt * 1024 + qoff[q + 1]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][mt][e], b[q & 1][nt][e], acc[mt][nt], 0, 0, 0);
        }
        __syncthreads();          // hipcc drains the DMA (vmcnt(0)) here: the next buffer is complete
    }
    epi.template run<Cfg>(acc, m0, n0, wm, wn, lane, M, N, smem, pre);
}

template <class Cfg, class ARows, class BRows, class Epi>
static int launch_gemm2(const ARows &A, const BRows &B, int K, const Epi &epi, int xcd, hipStream_t st) {
    auto kern = gemm2_kernel<Cfg, ARows, BRows, Epi>;
    constexpr int lds = Lds2<Cfg>::BYTES;
    static JlmLdsGrant grant;
    if (int rc = jlm_grant_lds(grant, reinterpret_cast<const void *>(kern), lds)) return rc;
    TileMap tm;
    tm.tiles_m = (A.nrows + Cfg::BM - 1) / Cfg::BM;
    tm.tiles_n = (B.nrows + Cfg::BN - 1) / Cfg::BN;
    tm.xcd = xcd;
    if (tm.tiles_m == 0 || tm.tiles_n == 0) return 0;
    hipLaunchKernelGGL(kern, dim3(tm.grid()), dim3(Cfg::NTHREADS), lds, st, A, B, K, epi, tm);
    JLM_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------ split-f16 mainloop ("f16x3")
// Same tiles, same LDS-DMA staging, same row sources and epilogues as gemm2_kernel, but both
// operands are SPLIT ROWS (include/jlm_hip.h): a 32-value k-step of a row is still 128 bytes =
// 8 granules, granule (2 kb + p) = plane p (0 hi, 1 lo) of the 8 values kb.  A k-step is two
// v_mfma_f32_32x32x16_f16 steps; lane half h of step s reads granules 4 s + 2 h + p, and each 32x32
// block of the wave tile takes 3 MFMAs per step (lo.hi, hi.lo, hi.hi) -- 6 x 32 cycles per k-step and
// block instead of the 16 x 64 of the f32 pipe.  The epilogue multiplies the accumulators by
// 2^-(eA + eB) (Epi::scale).
template <class Cfg, class ARows, class BRows, class Epi>
__global__ __launch_bounds__(Cfg::NTHREADS) void gemm_split_kernel(ARows A, BRows B, int K, Epi epi, TileMap tmap) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = Cfg::BM, BN = Cfg::BN, MT = Cfg::MT, NT = Cfg::NT;
    using L2 = Lds2<Cfg>;
    int tile_m, tile_n;
    if (!tmap.get(blockIdx.x, tile_m, tile_n)) return;
    const int M = A.count(), N = B.count();
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    if (m0 >= M || n0 >= N) return;
    JLM_WG_T(0);
    typename Epi::template Pre<Cfg> pre;
    epi.template prepare<Cfg>(pre, m0, M);      // index loads of the epilogue, in flight under the mainloop
    float *As = smem;                         // [2][BM][32]
    float *Bs = smem + 2 * BM * 32;           // [2][BN][32]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
    const int lrow = lane >> 3, lslot = lane & 7;
    typename ARows::St sa[L2::A_INST];
    typename BRows::St sb[L2::B_INST];
    int ka[L2::A_INST], kb[L2::B_INST];
#pragma unroll
    for (int i = 0; i < L2::A_INST; ++i) {
        const int r = (wave * L2::A_INST + i) * 8 + lrow;
        sa[i] = A.init(m0 + r, M);
        ka[i] = (lslot ^ ((r >> 1) & 7)) * 4;
    }
#pragma unroll
    for (int i = 0; i < L2::B_INST; ++i) {
        const int r = (wave * L2::B_INST + i) * 8 + lrow;
        sb[i] = B.init(n0 + r, N);
        kb[i] = (lslot ^ ((r >> 1) & 7)) * 4;
    }
    auto issue = [&](int k0, int buf) {
#pragma unroll
        for (int i = 0; i < L2::A_INST; ++i) {
            const int k = k0 + ka[i];
            const float *p = (k < K && A.valid(sa[i], k0)) ? A.ptr(sa[i], k0) + k : jlm_zero_page;
            GLDS16(p, As + (buf * BM + (wave * L2::A_INST + i) * 8) * 32);
        }
#pragma unroll
        for (int i = 0; i < L2::B_INST; ++i) {
            const int k = k0 + kb[i];
            const float *p = (k < K && B.valid(sb[i], k0)) ? B.ptr(sb[i], k0) + k : jlm_zero_page;
            GLDS16(p, Bs + (buf * BN + (wave * L2::B_INST + i) * 8) * 32);
        }
    };
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    const int nk = (K + BK - 1) / BK;
    const int li = lane & 31, h = lane >> 5;
    int goff[2][2];                            // float offset of (step s, plane p) inside the lane's row
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int p = 0; p < 2; ++p) goff[st][p] = li * 32 + (((4 * st + 2 * h + p) ^ ((li >> 1) & 7)) * 4);
    JLM_WG_T(1);
    issue(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue((kt + 1) * BK, cur ^ 1);
        const float *as = As + (cur * BM + wm * MT * 32) * 32;
        const float *bs = Bs + (cur * BN + wn * NT * 32) * 32;
        f16x8 a[2][MT][2], b[2][NT][2];
        auto load = [&](int st) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[st][mt][p] = *reinterpret_cast<const f16x8 *>(as + mt * 1024 + goff[st][p]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[st][nt][p] = *reinterpret_cast<const f16x8 *>(bs + nt * 1024 + goff[st][p]);
            }
        };
        load(0);
        load(1);
        // all fragment reads of the k-step in flight before the first MFMA: left alone the scheduler
        // sinks each read next to its use to save registers and every MFMA then waits an LDS round trip
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            // products outer, blocks inner: consecutive MFMAs never share an accumulator (MT * NT > 1)
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[st][mt][pr == 0 ? 1 : 0], b[st][nt][pr == 1 ? 1 : 0],
                                                                             acc[mt][nt], 0, 0, 0);
        }
        __syncthreads();
    }
    JLM_WG_T(2);
    epi.template run<Cfg>(acc, m0, n0, wm, wn, lane, M, N, smem, pre);
    JLM_WG_T(3);
}

// NST-stage form (NST = 3 is what is instantiated) for grids that do not fill the chip (the T projection: 240 workgroups, one per CU, so the
// extra stage costs no residency).  Per k-step:  wait(step kt landed) ; barrier ; request step kt + 2 ; MFMAs(kt):
// a DMA piece has two k-steps to arrive and the wait is s_waitcnt vmcnt(pieces of one stage), not vmcnt(0).
template <class Cfg, class ARows, class BRows, class Epi, int NST>
__global__ __launch_bounds__(Cfg::NTHREADS) void gemm_split3_kernel(ARows A, BRows B, int K, Epi epi, TileMap tmap) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = Cfg::BM, BN = Cfg::BN, MT = Cfg::MT, NT = Cfg::NT;
    using L2 = Lds2<Cfg>;
    constexpr int PER = L2::A_INST + L2::B_INST;      // DMA pieces of one stage issued by this wave
    int tile_m, tile_n;
    if (!tmap.get(blockIdx.x, tile_m, tile_n)) return;
    const int M = A.count(), N = B.count();
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    if (m0 >= M || n0 >= N) return;
    JLM_WG_T(0);
    typename Epi::template Pre<Cfg> pre;
    epi.template prepare<Cfg>(pre, m0, M);      // index loads of the epilogue, in flight under the mainloop
    float *As = smem;                         // [NST][BM][32]
    float *Bs = smem + NST * BM * 32;         // [NST][BN][32]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
    const int lrow = lane >> 3, lslot = lane & 7;
    typename ARows::St sa[L2::A_INST];
    typename BRows::St sb[L2::B_INST];
    int ka[L2::A_INST], kb[L2::B_INST];
#pragma unroll
    for (int i = 0; i < L2::A_INST; ++i) {
        const int r = (wave * L2::A_INST + i) * 8 + lrow;
        sa[i] = A.init(m0 + r, M);
        ka[i] = (lslot ^ ((r >> 1) & 7)) * 4;
    }
#pragma unroll
    for (int i = 0; i < L2::B_INST; ++i) {
        const int r = (wave * L2::B_INST + i) * 8 + lrow;
        sb[i] = B.init(n0 + r, N);
        kb[i] = (lslot ^ ((r >> 1) & 7)) * 4;
    }
    auto issue = [&](int k0, int stg) {
#pragma unroll
        for (int i = 0; i < L2::A_INST; ++i) {
            const int k = k0 + ka[i];
            const float *p = (k < K && A.valid(sa[i], k0)) ? A.ptr(sa[i], k0) + k : jlm_zero_page;
            GLDS16(p, As + (stg * BM + (wave * L2::A_INST + i) * 8) * 32);
        }
#pragma unroll
        for (int i = 0; i < L2::B_INST; ++i) {
            const int k = k0 + kb[i];
            const float *p = (k < K && B.valid(sb[i], k0)) ? B.ptr(sb[i], k0) + k : jlm_zero_page;
            GLDS16(p, Bs + (stg * BN + (wave * L2::B_INST + i) * 8) * 32);
        }
    };
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    const int nk = (K + BK - 1) / BK;
    const int li = lane & 31, h = lane >> 5;
    int goff[2][2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int p = 0; p < 2; ++p) goff[st][p] = li * 32 + (((4 * st + 2 * h + p) ^ ((li >> 1) & 7)) * 4);
    JLM_WG_T(1);
#pragma unroll
    for (int q = 0; q < NST - 1; ++q)
        if (q < nk) issue(q * BK, q);
    int stg = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // steps kt+1 .. kt+NST-2 may still be in flight (fewer at the end: wait for everything then)
        if (kt + NST - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NST - 2) * PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + NST - 1 < nk) issue((kt + NST - 1) * BK, stg == 0 ? NST - 1 : stg - 1);
        const float *as = As + (stg * BM + wm * MT * 32) * 32;
        const float *bs = Bs + (stg * BN + wn * NT * 32) * 32;
        f16x8 a[2][MT][2], b[2][NT][2];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[st][mt][p] = *reinterpret_cast<const f16x8 *>(as + mt * 1024 + goff[st][p]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[st][nt][p] = *reinterpret_cast<const f16x8 *>(bs + nt * 1024 + goff[st][p]);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[st][mt][pr == 0 ? 1 : 0], b[st][nt][pr == 1 ? 1 : 0],
                                                                             acc[mt][nt], 0, 0, 0);
        stg = stg == NST - 1 ? 0 : stg + 1;
    }
    JLM_WG_T(2);
    __syncthreads();                          // the epilogues reuse the staging memory
    epi.template run<Cfg>(acc, m0, n0, wm, wn, lane, M, N, smem, pre);
    JLM_WG_T(3);
}

template <class Cfg, class ARows, class BRows, class Epi, int NST>
static int launch_gemm_split3(const ARows &A, const BRows &B, int K, const Epi &epi, int xcd, hipStream_t st) {
    auto kern = gemm_split3_kernel<Cfg, ARows, BRows, Epi, NST>;
    constexpr int lds = Lds2<Cfg>::BYTES / 2 * NST;
    static JlmLdsGrant grant;
    if (int rc = jlm_grant_lds(grant, reinterpret_cast<const void *>(kern), lds)) return rc;
    TileMap tm;
    tm.tiles_m = (A.nrows + Cfg::BM - 1) / Cfg::BM;
    tm.tiles_n = (B.nrows + Cfg::BN - 1) / Cfg::BN;
    tm.xcd = xcd;
    if (tm.tiles_m == 0 || tm.tiles_n == 0) return 0;
    hipLaunchKernelGGL(kern, dim3(tm.grid()), dim3(Cfg::NTHREADS), lds, st, A, B, K, epi, tm);
    JLM_LAUNCH_CHECK();
    return 0;
}

template <class Cfg, class ARows, class BRows, class Epi>
static int launch_gemm_split(const ARows &A, const BRows &B, int K, const Epi &epi, int xcd, hipStream_t st) {
    auto kern = gemm_split_kernel<Cfg, ARows, BRows, Epi>;
    constexpr int lds = Lds2<Cfg>::BYTES;
    static JlmLdsGrant grant;
    if (int rc = jlm_grant_lds(grant, reinterpret_cast<const void *>(kern), lds)) return rc;
    TileMap tm;
    tm.tiles_m = (A.nrows + Cfg::BM - 1) / Cfg::BM;
    tm.tiles_n = (B.nrows + Cfg::BN - 1) / Cfg::BN;
    tm.xcd = xcd;
    if (tm.tiles_m == 0 || tm.tiles_n == 0) return 0;
    hipLaunchKernelGGL(kern, dim3(tm.grid()), dim3(Cfg::NTHREADS), lds, st, A, B, K, epi, tm);
    JLM_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ launchers
typedef TileCfg<2, 2, 2, 2> Cfg128;     // 128 x 128, wave 64 x 64
typedef TileCfg<2, 2, 1, 1> Cfg64;      //  64 x  64, wave 32 x 32

extern "C" int jlm_lstm_step(const float *h_in, const float *c_in, int ld_state, float *h_out, float *c_out,
                             const int *rows, const int *prev, const int *word, const float *emb, int ld_emb,
                             const float *wt, const float *bias, int kpad, int H, int E, int n_rows_max,
                             const int *n_dev, void *stream) {
    if (H % 32 != 0 || E % 4 != 0 || kpad % BK != 0 || kpad < H + E || ld_state % 4 || ld_emb % 4) return -1;
    // (H % 32: the [h | emb] switch of the A loader must fall on a k-step boundary)
    GateRows A;
    A.h = h_in; A.ldh = ld_state; A.rows = rows; A.prev = prev; A.word = word;
    A.emb = emb; A.lde = ld_emb; A.H = H; A.nrows = n_rows_max; A.ndev = n_dev;
    PlainRows B;
    B.base = wt; B.map = nullptr; B.ld = kpad; B.nrows = 4 * H; B.ndev = nullptr;
    EpiGate epi;
    epi.c_in = c_in; epi.h_out = h_out; epi.c_out = c_out; epi.ld = ld_state;
    epi.rows = rows; epi.prev = prev; epi.bias = bias;
    // K = H + E: chunks past it are zero filled, the packed weights are zero padded
    const int tiles_n = 4 * H / Cfg64::BN;
    // 64 x 64 tiles balance the CUs at a few thousand rows (1 280 workgroups = 5 per CU at R = 2 560);
    // from ~8 k rows on the balance is given and 128 x 64 tiles (fewer barriers per MFMA) are 4 % faster
    if (n_rows_max >= 8192) return launch_gemm2<TileCfg<2, 2, 2, 1>>(A, B, H + E, epi, (tiles_n % 8 == 0) ? 2 : 0, (hipStream_t)stream);
    return launch_gemm2<Cfg64>(A, B, H + E, epi, (tiles_n % 8 == 0) ? 2 : 0, (hipStream_t)stream);
}

extern "C" int jlm_gemm_nt(const float *Ap, int lda, const int *a_rows, const float *Bp, int ldb, const int *b_rows,
                           float *C, int ldc, const int *c_rows, const float *bias, int M, int N, int K,
                           const int *m_dev, void *stream) {
    if (K % 4 != 0 || lda % 4 != 0 || ldb % 4 != 0) return -1;
    PlainRows A, B;
    A.base = Ap; A.map = a_rows; A.ld = lda; A.nrows = M; A.ndev = m_dev;
    B.base = Bp; B.map = b_rows; B.ld = ldb; B.nrows = N; B.ndev = nullptr;
    EpiStore epi;
    epi.C = C; epi.c_map = c_rows; epi.ldc = ldc; epi.bias = bias;
    // small problems: 64 x 64 tiles give 4x the workgroups (K4: [R,512]x[512,256] is only 40 tiles of 128^2)
    long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (tiles128 < 512) return launch_gemm2<Cfg64>(A, B, K, epi, 0, (hipStream_t)stream);
    return launch_gemm2<Cfg128>(A, B, K, epi, 0, (hipStream_t)stream);
}

// Split-f16 form of jlm_gemm_nt (operands = split rows, strides in 4-byte units).  (The round-1 split LSTM step that lived here,
// jlm_lstm_step_split, left with ABI 9: the decode has used jlm_lstm_step_xg since round 2 -- HISTORY.md.)
extern "C" int jlm_gemm_nt_split(const void *Ap, int lda, const int *a_rows, const void *Bp, int ldb, const int *b_rows,
                                 float *C, int ldc, const int *c_rows, const float *bias, float descale, int M, int N,
                                 int K, const int *m_dev, void *stream) {
    if (K % 16 != 0 || lda % 16 != 0 || ldb % 16 != 0) return -1;
    PlainRows A, B;
    A.base = reinterpret_cast<const float *>(Ap); A.map = a_rows; A.ld = lda; A.nrows = M; A.ndev = m_dev;
    B.base = reinterpret_cast<const float *>(Bp); B.map = b_rows; B.ld = ldb; B.nrows = N; B.ndev = nullptr;
    EpiStore epi;
    epi.C = C; epi.c_map = c_rows; epi.ldc = ldc; epi.bias = bias; epi.scale = descale;
    long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    const long tiles64 = (long)((M + 63) / 64) * ((N + 63) / 64);
    static int stages = -1;                    // 3 stages: 17.4 -> 13.8 us on the T projection; 4 and 6 are no better
    if (stages < 0) { const char *e = getenv("JLM_T_STAGES"); stages = e ? atoi(e) : 3; }
    // XCD map 1: the column tiles of one row tile run on the same XCD, so a gathered A row crosses the fabric into ONE L2
    // instead of up to tiles_n of them (T projection: 36.6 MB fetched per launch for 9.4 MB of operands with the linear map)
    static int txcd = -1;
    if (txcd < 0) { const char *e = getenv("JLM_T_XCD"); txcd = e ? atoi(e) : 1; }
    if (tiles64 <= 256 && stages == 3) return launch_gemm_split3<Cfg64, PlainRows, PlainRows, EpiStore, 3>(A, B, K, epi, txcd, (hipStream_t)stream);
    if (tiles128 < 512) return launch_gemm_split<Cfg64>(A, B, K, epi, 0, (hipStream_t)stream);
    return launch_gemm_split<Cfg128>(A, B, K, epi, 0, (hipStream_t)stream);
}

extern "C" int jlm_vocab_lse_partials(const float *Bseg, int ldb, int n_vocab, int K, const float *T, int ldt,
                                      const int *rows, const float *bias, float *part, int ld_part, int tile0,
                                      int n_rows_max, const int *n_dev, void *stream) {
    if (K % 4 != 0 || ldb % 4 != 0 || ldt % 4 != 0) return -1;
    PlainRows A, B;
    A.base = Bseg; A.map = nullptr; A.ld = ldb; A.nrows = n_vocab; A.ndev = nullptr;
    B.base = T; B.map = rows; B.ld = ldt; B.nrows = n_rows_max; B.ndev = n_dev;
    EpiLse epi;
    epi.bias = bias; epi.part = part; epi.ld_part = ld_part; epi.tile0 = tile0;
    int r = launch_gemm2<Cfg128>(A, B, K, epi, 1, (hipStream_t)stream);
    if (r != 0) return r > 0 ? -r : r;
    return (n_vocab + Cfg128::BM - 1) / Cfg128::BM;
}

extern "C" int jlm_vocab_lse_partials_split(const void *Bsplit, int ldb, int n_vocab, int K, const void *Tsplit, int ldt,
                                            const int *rows, const float *bias, float descale, float *part, int ld_part,
                                            int tile0, int n_rows_max, const int *n_dev, void *stream) {
    if (K % 16 != 0 || ldb % 16 != 0 || ldt % 16 != 0) return -1;
    PlainRows A, B;
    A.base = reinterpret_cast<const float *>(Bsplit); A.map = nullptr; A.ld = ldb; A.nrows = n_vocab; A.ndev = nullptr;
    B.base = reinterpret_cast<const float *>(Tsplit); B.map = rows; B.ld = ldt; B.nrows = n_rows_max; B.ndev = n_dev;
    EpiLse epi;
    epi.bias = bias; epi.part = part; epi.ld_part = ld_part; epi.tile0 = tile0; epi.scale = descale;
    int r = launch_gemm_split<Cfg128>(A, B, K, epi, 1, (hipStream_t)stream);
    if (r != 0) return r > 0 ? -r : r;
    return (n_vocab + Cfg128::BM - 1) / Cfg128::BM;
}

// ------------------------------------------------- rows-stationary vocabulary LSE
// The D-softmax* segments have short contractions (k = 200 / 100 / 50): one output
// tile per workgroup pays a cold prologue and an epilogue per 7 / 4 / 2 k-steps.
// Here a workgroup keeps its 128 hypothesis rows STATIONARY -- each wave holds the
// MFMA B-fragments of its 32 rows for the whole contraction in registers (k/2
// floats per lane) -- and streams a contiguous range of vocabulary tiles (32*MT
// words each) through a double-buffered LDS ring in one continuous (tile, k-step)
// pipeline.  Each lane owns the running (max, sum exp) of one row; a vocabulary
// tile ends with 16*MT in-register updates and no cross-lane traffic.  One
// (max, sum) pair per (row, vocabulary range) is written at the very end.
// One kernel instantiation per number of k-steps NK (its own register budget);
// the grid of a launch is sized to ONE resident round of workgroups.
JLM_PROF_READER(jlm_prof_read)

template <int NK, int MT>
__device__ __forceinline__ void lse_stat_body(
    const jlm_segment &sg, const float *__restrict__ bias, int p_in_seg, int parts_in_seg, int pt, int n_paths,
    const float *__restrict__ T, int ldt, const int *__restrict__ rows, float2 *__restrict__ part_row, float *smem) {
    constexpr int BMV = 32 * MT;                   // vocabulary rows per tile
    constexpr int NINST = BMV / 32;                // LDS-DMA instructions per wave per k-step (8 rows each, 4 waves)
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    JLM_PROF_DECL();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, li = lane & 31;
    const int K = sg.k, n_vocab = sg.v_end - sg.v_start, ldb = sg.ldb;
    const int ntiles = (n_vocab + BMV - 1) / BMV;
    const int vt0 = (int)((long)ntiles * p_in_seg / parts_in_seg), vt1 = (int)((long)ntiles * (p_in_seg + 1) / parts_in_seg);
    const float *__restrict__ Bp = sg.B;
    // 1. this lane's row fragments for the whole contraction, pre-scaled by log2(e): the logits come
    //    out of the matrix pipe in base-2 units and the fold uses the bare v_exp_f32
    const int prow = pt * 128 + wave * 32 + li;
    const bool row_ok = prow < n_paths;
    const float *trow = T + (size_t)(row_ok ? (rows ? rows[prow] : prow) : 0) * ldt + sg.t_off;
    // The last k-step may be short: with kv = valid k rounded up to 8, lane-half h owns k = h*kv/2 ..
    // and only nq_last = kv/8 of the four MFMA quads are issued (k = 200 / 104 / 56 cost 6.25 / 3.25 /
    // 1.75 k-steps of MFMAs instead of 7 / 4 / 2).  Full k-steps are the nq = 4 case of the same map.
    const int nq_last = (K - 32 * (NK - 1) + 7) >> 3;
    f32x4 tf[NK][4];
#pragma unroll
    for (int kt = 0; kt < NK; ++kt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nq = (kt == NK - 1) ? nq_last : 4;
            const int k = kt * 32 + (h * nq + q) * 4;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(trow + (k < K ? k : 0));
            tf[kt][q] = (row_ok && q < nq && k < K) ? v * LOG2E : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    // 2. stream vocabulary tiles: LDS-DMA, unpadded [row][8 x 16 B] tiles, source-side swizzle (see gemm2_kernel)
    float *Bs = smem;                              // [2][BMV][32]
    float *bias_s = smem + 2 * BMV * 32;           // [3][BMV], base-2 units
    const int lrow = lane >> 3, lslot = lane & 7;
    int srow[NINST], skk[NINST];
#pragma unroll
    for (int i = 0; i < NINST; ++i) {
        srow[i] = (wave * NINST + i) * 8 + lrow;
        skk[i] = (lslot ^ ((srow[i] >> 1) & 7)) * 4;
    }
    auto issue = [&](int t, int kt, int buf) {
#pragma unroll
        for (int i = 0; i < NINST; ++i) {
            const int vrow = t * BMV + srow[i], k = kt * 32 + skk[i];
            const float *src = (vrow < n_vocab && k < K) ? Bp + (size_t)vrow * ldb + k : jlm_zero_page;
            GLDS16(src, Bs + (buf * BMV + (wave * NINST + i) * 8) * 32);
        }
    };
    // bias of rows past the vocabulary is a huge negative number: padded rows (zero-page data,
    // accumulator 0) then fold to 2^(-inf) = 0 with no masking code in the fold
    auto bias_stage = [&](int t) {
        if (tid < BMV) {
            const int vrow = t * BMV + tid;
            bias_s[((t - vt0) % 3) * BMV + tid] = vrow < n_vocab ? bias[vrow] * LOG2E : JLM_NEG_BIG;
        }
    };
    float m = JLM_NEG_BIG, s = 0.0f;
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
    int qoff[4], qoff_last[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        qoff[q] = li * 32 + (((h * 4 + q) ^ ((li >> 1) & 7)) * 4);
        qoff_last[q] = li * 32 + ((((h * nq_last + q) & 7) ^ ((li >> 1) & 7)) * 4);
    }
    JLM_PROF_MARK(p_t1);
    issue(vt0, 0, 0);
    bias_stage(vt0);
    __syncthreads();
    JLM_PROF_MARK(p_t2);
    int buf = 0;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = vt0; t < vt1; ++t) {
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            const bool last_k = (kt == NK - 1);
            issue(last_k ? t + 1 : t, last_k ? 0 : kt + 1, buf ^ 1);   // past the range: harmless (zero page / next range)
            const float *bs = Bs + buf * BMV * 32;
            f32x4 a[2][MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                a[0][mt] = *reinterpret_cast<const f32x4 *>(bs + mt * 1024 + (last_k ? qoff_last[0] : qoff[0]));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (last_k && q >= nq_last) break;       // short last k-step (uniform)
                if (q < 3) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        a[(q + 1) & 1][mt] =
                            *reinterpret_cast<const f32x4 *>(bs + mt * 1024 + (last_k ? qoff_last[q + 1] : qoff[q + 1]));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)      // a tile's first MFMA starts from C = 0: no accumulator reset pass
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][mt][e], tf[kt][q][e],
                                                                       (kt == 0 && q == 0 && e == 0) ? zero16 : acc[mt], 0, 0, 0);
            }
            // bias of the next tile: its global load is waited for together with the DMA at the barrier
            // (a wait placed before the MFMAs would also drain the DMA just issued: vmcnt is in-order)
            if (last_k) bias_stage(t + 1);
            JLM_PROF_MARK(p_x);
            __syncthreads();
            JLM_PROF_ADD(p_bar, p_x);
            buf ^= 1;
        }
        JLM_PROF_MARK(p_x);
        // 3. fold this tile's 16*MT base-2 logits of the lane's row into (m, s): branch free,
        //    ~4 VALU + 1 v_exp per logit
        const float *bt = bias_s + ((t - vt0) % 3) * BMV + 4 * h;
        float tmax = JLM_NEG_BIG;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bt + mt * 32 + 8 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[mt][4 * j + e] + b4[e];
                    acc[mt][4 * j + e] = v;
                    tmax = fmaxf(tmax, v);
                }
            }
        const float mn = fmaxf(m, tmax);
        float add0 = 0.0f, add1 = 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                add0 += __builtin_amdgcn_exp2f(acc[mt][r] - mn);
                add1 += __builtin_amdgcn_exp2f(acc[mt][r + 1] - mn);
            }
        s = s * __builtin_amdgcn_exp2f(m - mn) + (add0 + add1);
        m = mn;
        JLM_PROF_ADD(p_fold, p_x);
    }
    JLM_PROF_FLUSH();
    const float m2 = __shfl_xor(m, 32), s2 = __shfl_xor(s, 32);
    {   // merge the two lane halves (base-2 units), then hand out natural-log units
        const float mm = fmaxf(m, m2);
        s = s * __builtin_amdgcn_exp2f(m - mm) + s2 * __builtin_amdgcn_exp2f(m2 - mm);
        m = mm * LN2;
    }
    if (h == 0 && row_ok) part_row[prow] = make_float2(m, s);
}

// All segments of a model in ONE launch (one resident round of workgroups): range p belongs to
// segment part_seg[p]; ranges are handed to segments in proportion to their MFMA work.
#define LSES_MAX_PARTS 96
struct LseStatArgs {
    int n_parts, n_segs;
    jlm_segment seg[JLM_MAX_SEGMENTS];
    const float *bias[JLM_MAX_SEGMENTS];
    short part_first[JLM_MAX_SEGMENTS + 1];     // ranges [part_first[i], part_first[i+1]) belong to segment i
};

__global__ __launch_bounds__(256, 2) void vocab_lse_stationary_kernel(LseStatArgs a, const float *__restrict__ T, int ldt,
                                                                       const int *__restrict__ rows, float2 *__restrict__ part,
                                                                       int ld_part, int n_rows_max, const int *n_dev,
                                                                       int n_ptiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n_paths = n_dev ? min(*n_dev, n_rows_max) : n_rows_max;
    // XCD-aware order (n_parts is a multiple of 8 then): one XCD walks all row tiles of a vocabulary
    // range, which stays in its L2, and every XCD gets the same number of ranges.  With fewer than 8
    // ranges the workgroups of a range are simply dealt round-robin over the XCDs.
    const int b = blockIdx.x;
    int p, pt;
    if ((a.n_parts & 7) == 0) { const int x = b & 7, jb = b >> 3; p = (jb / n_ptiles) * 8 + x; pt = jb % n_ptiles; }
    else { p = b / n_ptiles; pt = b % n_ptiles; }
    if (p >= a.n_parts || pt * 128 >= n_paths) return;
    int si = 0;
    while (si + 1 < a.n_segs && p >= a.part_first[si + 1]) ++si;
    const jlm_segment sg = a.seg[si];
    const float *bias = a.bias[si];
    const int pis = p - a.part_first[si], npis = a.part_first[si + 1] - a.part_first[si];
    float2 *prow = part + (size_t)p * ld_part;
    switch ((sg.k + BK - 1) / BK) {
        case 1:
        case 2: lse_stat_body<2, 4>(sg, bias, pis, npis, pt, n_paths, T, ldt, rows, prow, smem); break;
        case 3: lse_stat_body<3, 4>(sg, bias, pis, npis, pt, n_paths, T, ldt, rows, prow, smem); break;
        case 4: lse_stat_body<4, 4>(sg, bias, pis, npis, pt, n_paths, T, ldt, rows, prow, smem); break;
        case 5: lse_stat_body<5, 4>(sg, bias, pis, npis, pt, n_paths, T, ldt, rows, prow, smem); break;
        case 6: lse_stat_body<6, 4>(sg, bias, pis, npis, pt, n_paths, T, ldt, rows, prow, smem); break;
        case 7: lse_stat_body<7, 4>(sg, bias, pis, npis, pt, n_paths, T, ldt, rows, prow, smem); break;
        default: lse_stat_body<8, 2>(sg, bias, pis, npis, pt, n_paths, T, ldt, rows, prow, smem); break;
    }
}

// Returns the number of partial slices written (to be folded by jlm_lse_combine), or <0.
extern "C" int jlm_vocab_lse_stationary(const jlm_segment *segs_host, int n_segs, const float *b2, const float *T, int ldt,
                                        const int *rows, float *part, int ld_part, int max_parts, int n_rows_max,
                                        const int *n_dev, void *stream) {
    if (n_segs < 1 || n_segs > JLM_MAX_SEGMENTS || ldt % 4 || n_rows_max <= 0) return -1;
    LseStatArgs a;
    a.n_segs = n_segs;
    long work[JLM_MAX_SEGMENTS], total = 0;
    int ntiles[JLM_MAX_SEGMENTS];
    for (int i = 0; i < n_segs; ++i) {
        const jlm_segment &sg = segs_host[i];
        const int nk = (sg.k + BK - 1) / BK;
        if (nk > 8 || sg.k % 4 || sg.ldb % 4 || sg.t_off % 4) return -2;   // caller falls back to the tile form
        a.seg[i] = sg;
        a.bias[i] = b2 + sg.v_start;
        const int bmv = nk >= 8 ? 64 : 128;
        ntiles[i] = (sg.v_end - sg.v_start + bmv - 1) / bmv;
        work[i] = (long)(sg.v_end - sg.v_start) * ((sg.k + 7) / 8);          // MFMA quads actually issued
        total += work[i];
    }
    const int n_ptiles = (n_rows_max + 127) / 128;
    int cap = max_parts < LSES_MAX_PARTS ? max_parts : LSES_MAX_PARTS;
    if (cap < n_segs) return -1;
    int np = (2 * 256) / n_ptiles;                 // one resident round: 2 workgroups per CU (register bound)
    if (np < n_segs) np = n_segs;
    if (np > cap) np = cap;
    if (np >= 8) np &= ~7;                         // whole ranges per XCD, the same number on each (see kernel)
    if (np < n_segs) np = n_segs;
    // ranges per segment in proportion to its work, at least one, at most one per tile
    int given = 0, k[JLM_MAX_SEGMENTS];
    for (int i = 0; i < n_segs; ++i) {
        k[i] = (int)((work[i] * np + total / 2) / total);
        if (k[i] < 1) k[i] = 1;
        if (k[i] > ntiles[i]) k[i] = ntiles[i];
        given += k[i];
    }
    for (int guard = 0; given != np && guard < 4 * LSES_MAX_PARTS; ++guard) {   // settle the rounding on the fattest / leanest ranges
        int best = -1;
        for (int i = 0; i < n_segs; ++i) {
            if (given < np) { if (k[i] < ntiles[i] && (best < 0 || work[i] * k[best] > work[best] * k[i])) best = i; }
            else { if (k[i] > 1 && (best < 0 || work[i] * k[best] < work[best] * k[i])) best = i; }
        }
        if (best < 0) break;
        if (given < np) { ++k[best]; ++given; } else { --k[best]; --given; }
    }
    a.part_first[0] = 0;
    for (int i = 0; i < n_segs; ++i) a.part_first[i + 1] = (short)(a.part_first[i] + k[i]);
    a.n_parts = given;
    const int lds = (2 * 128 * 32 + 3 * 128) * 4;
    const int grid = given * n_ptiles;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(vocab_lse_stationary_kernel, dim3(grid), dim3(256), lds, st, a, T, ldt, rows,
                       reinterpret_cast<float2 *>(part), ld_part, n_rows_max, n_dev, n_ptiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(int)e - 100;
    return given;
}

// ------------------------------------------------------- word-list LSE on the matrix pipe
// Selected-vocabulary / incremental-vocabulary normaliser (jlm_wordlist_lse) for single-segment
// models: one workgroup per (sentence, frame) group, its <= 32 hypothesis rows stationary in
// registers (as in vocab_lse_stationary_kernel), the group's word list walked in 32-word tiles that
// are GATHERED row by row straight into LDS by the DMA (the per-lane source address is the word's
// weight row).  The four waves take every fourth tile, each with a private double buffer -- no
// workgroup barrier in the loop, only the wave's own vmcnt -- and meet once at the end.
template <int NK>
__global__ __launch_bounds__(256) void wordlist_lse_mfma_kernel(
    jlm_segment sg, const float *__restrict__ b2, const float *__restrict__ T, int ldt,
    const int *__restrict__ g0v, const int *__restrict__ cnt, const int *__restrict__ cnt_idx,
    const int *__restrict__ wl, const int *__restrict__ wl_off, const int *__restrict__ wl_idx, int wl_base,
    float *__restrict__ run_max, double *__restrict__ run_sum, double *__restrict__ lse, int merge, int beam) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    const int j = blockIdx.x;
    // beams above 32: a group's rows are taken by ceil(beam / 32) workgroups (blockIdx.y), 32 rows each
    const int nrows = min(min(cnt[cnt_idx[j]], beam) - 32 * (int)blockIdx.y, 32);
    if (nrows <= 0) return;
    const int gbase = g0v[j] + 32 * (int)blockIdx.y;
    const int lid = wl_base + wl_idx[j];
    const int w0 = wl_off[lid], nw = wl_off[lid + 1] - w0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, li = lane & 31;
    const int K = sg.k, ldb = sg.ldb;
    const float *__restrict__ Bp = sg.B;
    const int ntiles = (nw + 31) >> 5;
    float *Bs = smem + wave * (2 * 32 * 32 + 2 * 32);      // this wave's [2][32][32] ring + [2][32] bias
    float *bias_s = Bs + 2 * 32 * 32;
    float m = JLM_NEG_BIG, s = 0.0f;
    if (wave < ntiles) {
        const bool row_ok = li < nrows;
        const float *trow = T + (size_t)(gbase + (row_ok ? li : 0)) * ldt + sg.t_off;
        const int nq_last = (K - 32 * (NK - 1) + 7) >> 3;
        f32x4 tf[NK][4];
#pragma unroll
        for (int kt = 0; kt < NK; ++kt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nq = (kt == NK - 1) ? nq_last : 4;
                const int k = kt * 32 + (h * nq + q) * 4;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(trow + (k < K ? k : 0));
                tf[kt][q] = (row_ok && q < nq && k < K) ? v * LOG2E : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        const int lrow = lane >> 3, lslot = lane & 7;
        int skk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) skk[i] = (lslot ^ (((i * 8 + lrow) >> 1) & 7)) * 4;
        int qoff[4], qoff_last[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            qoff[q] = li * 32 + (((h * 4 + q) ^ ((li >> 1) & 7)) * 4);
            qoff_last[q] = li * 32 + ((((h * nq_last + q) & 7) ^ ((li >> 1) & 7)) * 4);
        }
        // word ids of a tile: rows i*8 + lrow for the DMA, row li for the bias
        auto load_wids = [&](int t, int (&wd)[4], int &wb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = t * 32 + i * 8 + lrow;
                wd[i] = r < nw ? wl[w0 + r] : -1;
            }
            const int r = t * 32 + li;
            wb = r < nw ? wl[w0 + r] : -1;
        };
        auto issue = [&](const int (&wd)[4], int kt, int buf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = kt * 32 + skk[i];
                const float *src = (wd[i] >= 0 && k < K) ? Bp + (size_t)(wd[i] - sg.v_start) * ldb + k : jlm_zero_page;
                GLDS16(src, Bs + (buf * 32 + i * 8) * 32);
            }
        };
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int wd[4], wb, wdn[4], wbn;
        load_wids(wave, wd, wb);
        issue(wd, 0, 0);
        int buf = 0;
        f32x16 acc = zero16;
        for (int t = wave; t < ntiles; t += 4) {
            const bool more_tiles = t + 4 < ntiles;
            if (more_tiles) load_wids(t + 4, wdn, wbn);
            if (h == 0) bias_s[((t >> 2) & 1) * 32 + li] = wb >= 0 ? b2[wb] * LOG2E : JLM_NEG_BIG;
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                const bool last_k = (kt == NK - 1);
                // this chunk's DMA (and everything older) has landed; same-wave visibility needs only the count
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!last_k) issue(wd, kt + 1, buf ^ 1);
                else if (more_tiles) issue(wdn, 0, buf ^ 1);
                const float *bs = Bs + buf * 32 * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (last_k && q >= nq_last) break;
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(bs + (last_k ? qoff_last[q] : qoff[q]));
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], tf[kt][q][e],
                                                                   (kt == 0 && q == 0 && e == 0) ? zero16 : acc, 0, 0, 0);
                }
                buf ^= 1;
            }
            // fold the tile's 16 base-2 logits of this lane's row
            const float *bt = bias_s + ((t >> 2) & 1) * 32 + 4 * h;
            float tmax = JLM_NEG_BIG;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bt + 8 * jj);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[4 * jj + e] + b4[e];
                    acc[4 * jj + e] = v;
                    tmax = fmaxf(tmax, v);
                }
            }
            const float mn = fmaxf(m, tmax);
            float add = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) add += __builtin_amdgcn_exp2f(acc[r] - mn);
            s = s * __builtin_amdgcn_exp2f(m - mn) + add;
            m = mn;
            if (more_tiles) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wd[i] = wdn[i];
                wb = wbn;
            }
        }
        const float m2 = __shfl_xor(m, 32), s2 = __shfl_xor(s, 32);
        const float mm = fmaxf(m, m2);
        s = s * __builtin_amdgcn_exp2f(m - mm) + s2 * __builtin_amdgcn_exp2f(m2 - mm);
        m = mm;
    }
    // the four waves' partial (max, sum) per row meet in LDS (base-2 units)
    __syncthreads();
    float *red = smem;                     // [4][32][2], reuses wave 0's ring
    if (h == 0) { red[(wave * 32 + li) * 2] = m; red[(wave * 32 + li) * 2 + 1] = s; }
    __syncthreads();
    if (tid < nrows) {
        float M = red[tid * 2], S = red[tid * 2 + 1];
        for (int w = 1; w < 4; ++w) {
            const float m2 = red[(w * 32 + tid) * 2], s2 = red[(w * 32 + tid) * 2 + 1];
            const float mm = fmaxf(M, m2);
            S = S * __builtin_amdgcn_exp2f(M - mm) + s2 * __builtin_amdgcn_exp2f(m2 - mm);
            M = mm;
        }
        const int g = gbase + tid;
        float Mn = M * LN2;                // natural-log units from here on
        double Sd = (double)S;
        if (merge) {
            const float pm = run_max[g];
            const double ps = run_sum[g];
            const float mm = fmaxf(pm, Mn);
            Sd = ps * exp((double)pm - (double)mm) + Sd * exp((double)Mn - (double)mm);
            Mn = mm;
        }
        run_max[g] = Mn;
        run_sum[g] = Sd;
        lse[g] = (double)Mn + log(Sd);
    }
}

// Single-segment fast path of jlm_wordlist_lse (declared in jlm_beam.hip's launcher).
extern "C" int jlm_wordlist_lse_mfma(const jlm_segment *seg_host, const float *b2, const float *T, int ldt, const int *g0,
                                     const int *cnt, const int *cnt_idx, const int *wl, const int *wl_off,
                                     const int *wl_idx, int wl_base, float *run_max, double *run_sum, double *lse,
                                     int merge, int beam, int n_groups, void *stream) {
    const jlm_segment sg = *seg_host;
    const int nk = (sg.k + BK - 1) / BK;
    if (nk < 1 || nk > 8 || sg.k % 4 || sg.ldb % 4 || sg.t_off % 4 || ldt % 4 || beam > 64) return -2;
    if (n_groups <= 0) return 0;
    const int lds = 4 * (2 * 32 * 32 + 2 * 32) * 4;
    hipStream_t st = (hipStream_t)stream;
#define JLM_WL_LAUNCH(N)                                                                                              \
    hipLaunchKernelGGL(wordlist_lse_mfma_kernel<N>, dim3(n_groups, (beam + 31) / 32), dim3(256), lds, st, sg, b2, T, ldt, g0, cnt, cnt_idx, \
                       wl, wl_off, wl_idx, wl_base, run_max, run_sum, lse, merge, beam)
    switch (nk) {
        case 1: JLM_WL_LAUNCH(1); break;
        case 2: JLM_WL_LAUNCH(2); break;
        case 3: JLM_WL_LAUNCH(3); break;
        case 4: JLM_WL_LAUNCH(4); break;
        case 5: JLM_WL_LAUNCH(5); break;
        case 6: JLM_WL_LAUNCH(6); break;
        case 7: JLM_WL_LAUNCH(7); break;
        default: JLM_WL_LAUNCH(8); break;
    }
#undef JLM_WL_LAUNCH
    JLM_LAUNCH_CHECK();
    return 0;
}

// lse[g] = log sum exp over the tile partials of one row.  One lane per row (coalesced
// float2 reads along the row index), the tiles of a row are split over the 16 waves of
// the workgroup and merged online, then the 16 partial (max, sum) pairs meet in LDS.
#define LSEC_WAVES 16
__global__ __launch_bounds__(LSEC_WAVES * 64) void lse_combine_kernel(const float2 *part, int ld_part, int n_tiles,
                                                                      const int *rows, double *lse, int n_rows_max,
                                                                      const int *n_dev) {
    __shared__ float sm_m[LSEC_WAVES][64];
    __shared__ double sm_s[LSEC_WAVES][64];
    const int n = n_dev ? min(*n_dev, n_rows_max) : n_rows_max;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= n) return;
    float m = JLM_NEG_BIG;
    double s = 0.0;
    if (r < n) {
        for (int t = wave; t < n_tiles; t += LSEC_WAVES) {
            const float2 p = part[(size_t)t * ld_part + r];
            const float mm = fmaxf(m, p.x);
            s = s * (double)expf(m - mm) + (double)p.y * (double)expf(p.x - mm);
            m = mm;
        }
    }
    sm_m[wave][lane] = m;
    sm_s[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && r < n) {
        float M = sm_m[0][lane];
        for (int w = 1; w < LSEC_WAVES; ++w) M = fmaxf(M, sm_m[w][lane]);
        double S = 0.0;
        for (int w = 0; w < LSEC_WAVES; ++w) S += sm_s[w][lane] * exp((double)sm_m[w][lane] - (double)M);
        const int g = rows ? rows[r] : r;
        lse[g] = (double)M + log(S);
    }
}

extern "C" int jlm_lse_combine(const float *part, int ld_part, int n_tiles, const int *rows, double *lse,
                               int n_rows_max, const int *n_dev, void *stream) {
    if (n_rows_max <= 0) return 0;
    const int grid = (n_rows_max + 63) / 64;
    hipLaunchKernelGGL(lse_combine_kernel, dim3(grid), dim3(LSEC_WAVES * 64), 0, (hipStream_t)stream,
                       reinterpret_cast<const float2 *>(part), ld_part, n_tiles, rows, lse, n_rows_max, n_dev);
    JLM_LAUNCH_CHECK();
    return 0;
}
