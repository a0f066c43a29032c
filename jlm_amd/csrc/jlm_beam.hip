// Lattice beam-search kernels for gfx950: word-addressed logits (edge logits,
// selected-vocabulary log-sum-exp), candidate scoring + stable top-k, n-best
// back-pointer trace, row softmax.  All of it is small, irregular, HBM/latency
// bound integer + scalar work: one 64-lane wave (or one workgroup) per sentence,
// wavefront shuffles for the reductions, no MFMA.
#include "jlm_common.h"
#include <type_traits>
#include <stdlib.h>

// ------------------------------------------------------------ word-list logits
// One workgroup per (sentence, frame) group.  The group's <= beam hypothesis
// rows of T are staged in LDS once; 8 lanes share one word (each reads 16-B
// chunks k = sub, sub+8, ... of the word's weight row: 128 B contiguous per
// word per pass -> coalesced), 32 words per pass; rows are processed 8 at a time
// out of registers; partial dots are reduced over the 8 lanes with 3 xor-shuffles.
#define WL_THREADS 256
#define WL_ROWS 16        // hypothesis rows per pass: beam <= 16 reads every weight row once

template <int MODE>   // 0: edge logits, 1: log-sum-exp over the list
__global__ __launch_bounds__(WL_THREADS) void wordlist_kernel(
    SegTable segs, const float *__restrict__ b2, const float *__restrict__ T, int ldt,
    const int *__restrict__ g0v, const int *__restrict__ cnt, const int *__restrict__ cnt_idx,
    const int *__restrict__ wl, const int *__restrict__ wl_w, const int *__restrict__ wl_off, const int *__restrict__ wl_idx,
    int wl_base, const int *__restrict__ wl_out, float *__restrict__ edge, float *__restrict__ run_max,
    double *__restrict__ run_sum, double *__restrict__ lse, int merge, int beam) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int j = blockIdx.x;
    const int nrows = min(cnt[cnt_idx[j]], beam);
    if (nrows <= 0) return;
    const int gbase = g0v[j];
    const int lid = wl_base + wl_idx[j];
    const int w0 = wl_off[lid], nw = wl_off[lid + 1] - w0;
    if (MODE == 0 && nw == 0) return;
    const int tid = threadIdx.x;
    float *red = sm + (size_t)WL_ROWS * ldt;         // [4 waves][WL_ROWS][2] reduction scratch
    const int sub = tid & 7, slot = tid >> 3;
    for (int rc = 0; rc < nrows; rc += WL_ROWS) {
        {   // stage this pass's rows T[gbase + rc .. + WL_ROWS) (consecutive in g): LDS is WL_ROWS rows whatever the beam
            if (rc) __syncthreads();
            const int nr = min(WL_ROWS, nrows - rc);
            const f32x4 *src = reinterpret_cast<const f32x4 *>(T + (size_t)(gbase + rc) * ldt);
            f32x4 *dst = reinterpret_cast<f32x4 *>(sm);
            for (int i = tid; i < nr * (ldt / 4); i += WL_THREADS) dst[i] = src[i];
            __syncthreads();
        }
        float rm[WL_ROWS], rs[WL_ROWS];
#pragma unroll
        for (int r = 0; r < WL_ROWS; ++r) { rm[r] = JLM_NEG_BIG; rs[r] = 0.0f; }
        for (int wb = 0; wb < nw; wb += WL_THREADS / 8) {
            const int wi = wb + slot;
            const bool valid = wi < nw;
            const int w = valid ? wl[w0 + wi] : -1;
            // wl_w (the *_perm entry points): the weight row is word wl_w[i]'s, the bias word wl[i]'s
            const int ww = (valid && wl_w) ? wl_w[w0 + wi] : w;
            int K4 = 0, toff = 0;
            const f32x4 *brow = nullptr;
            if (valid) {
                for (int si = 0; si < segs.n; ++si)
                    if (ww >= segs.s[si].v_start && ww < segs.s[si].v_end) {
                        K4 = segs.s[si].k >> 2;
                        toff = segs.s[si].t_off;
                        brow = reinterpret_cast<const f32x4 *>(segs.s[si].B + (size_t)(ww - segs.s[si].v_start) * segs.s[si].ldb);
                    }
            }
            float acc[WL_ROWS];
#pragma unroll
            for (int r = 0; r < WL_ROWS; ++r) acc[r] = 0.0f;
            // the word's whole share of the weight row is requested before any of it is used (k <= 256: at
            // most 8 x 16 B per lane): one memory round trip per word instead of one per 32 k-values
            f32x4 bv[8];
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                const int kc = sub + 8 * c8;
                bv[c8] = (kc < K4 && kc < 64) ? brow[kc] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                const int kc = sub + 8 * c8;
                if (kc >= K4 || kc >= 64) continue;
#pragma unroll
                for (int r = 0; r < WL_ROWS; ++r) {
                    if (rc + r < nrows) {
                        const f32x4 tv = *reinterpret_cast<const f32x4 *>(sm + (size_t)r * ldt + toff + kc * 4);
                        acc[r] = fmaf(bv[c8][0], tv[0], acc[r]);
                        acc[r] = fmaf(bv[c8][1], tv[1], acc[r]);
                        acc[r] = fmaf(bv[c8][2], tv[2], acc[r]);
                        acc[r] = fmaf(bv[c8][3], tv[3], acc[r]);
                    }
                }
            }
            for (int kc = sub + 64; kc < K4; kc += 8) {            // k > 256 (untied models): the tail, as before
                const f32x4 bw = brow[kc];
#pragma unroll
                for (int r = 0; r < WL_ROWS; ++r) {
                    if (rc + r < nrows) {
                        const f32x4 tv = *reinterpret_cast<const f32x4 *>(sm + (size_t)r * ldt + toff + kc * 4);
                        acc[r] = fmaf(bw[0], tv[0], acc[r]);
                        acc[r] = fmaf(bw[1], tv[1], acc[r]);
                        acc[r] = fmaf(bw[2], tv[2], acc[r]);
                        acc[r] = fmaf(bw[3], tv[3], acc[r]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < WL_ROWS; ++r) {
                acc[r] += __shfl_xor(acc[r], 1);
                acc[r] += __shfl_xor(acc[r], 2);
                acc[r] += __shfl_xor(acc[r], 4);
            }
            if (valid && sub == 0) {
                const float bw = b2[w];
#pragma unroll
                for (int r = 0; r < WL_ROWS; ++r) {
                    if (rc + r < nrows) {
                        const float y = acc[r] + bw;
                        if (MODE == 0) {
                            edge[(size_t)wl_out[w0 + wi] * beam + rc + r] = y;
                        } else {
                            const float mm = fmaxf(rm[r], y);
                            rs[r] = rs[r] * expf(rm[r] - mm) + expf(y - mm);
                            rm[r] = mm;
                        }
                    }
                }
            }
        }
        if (MODE == 1) {
            // workgroup reduction of (max, sum) per row: wave shuffles, then 4 waves through LDS
#pragma unroll
            for (int r = 0; r < WL_ROWS; ++r) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    float m2 = __shfl_xor(rm[r], off), s2 = __shfl_xor(rs[r], off);
                    lse_merge(rm[r], rs[r], m2, s2);
                }
            }
            const int wave = tid >> 6, lane = tid & 63;
            __syncthreads();
            if (lane == 0)
                for (int r = 0; r < WL_ROWS; ++r) {
                    red[(wave * WL_ROWS + r) * 2] = rm[r];
                    red[(wave * WL_ROWS + r) * 2 + 1] = rs[r];
                }
            __syncthreads();
            if (tid < WL_ROWS && rc + tid < nrows) {
                float m = red[tid * 2], s = red[tid * 2 + 1];
                for (int w2 = 1; w2 < WL_THREADS / 64; ++w2)
                    lse_merge(m, s, red[(w2 * WL_ROWS + tid) * 2], red[(w2 * WL_ROWS + tid) * 2 + 1]);
                const int g = gbase + rc + tid;
                double S = (double)s;
                float M = m;
                if (merge) {
                    const float pm = run_max[g];
                    const double ps = run_sum[g];
                    M = fmaxf(pm, m);
                    S = ps * exp((double)pm - (double)M) + (double)s * exp((double)m - (double)M);
                }
                run_max[g] = M;
                run_sum[g] = S;
                lse[g] = (double)M + log(S);
            }
        }
    }
}

static int seg_table(const jlm_segment *segs_host, int n_segs, SegTable &t) {
    if (n_segs < 1 || n_segs > JLM_MAX_SEGMENTS) return -1;
    t.n = n_segs;
    for (int i = 0; i < n_segs; ++i) {
        t.s[i] = segs_host[i];
        if (t.s[i].k % 4 || t.s[i].ldb % 4 || t.s[i].t_off % 4) return -1;
    }
    return 0;
}

static size_t wl_lds_bytes(int /*beam*/, int ldt) { return ((size_t)WL_ROWS * ldt + 4 * WL_ROWS * 2) * sizeof(float); }

extern "C" int jlm_edge_logits(const jlm_segment *segs_host, int n_segs, const float *b2, const float *T, int ldt,
                               const int *g0, const int *cnt, const int *cnt_idx, const int *wl, const int *wl_off,
                               const int *wl_idx, int wl_base, const int *wl_out, float *edge, int beam, int n_groups,
                               void *stream) {
    return jlm_edge_logits_perm(segs_host, n_segs, b2, T, ldt, g0, cnt, cnt_idx, wl, nullptr, wl_off, wl_idx, wl_base, wl_out,
                                edge, beam, n_groups, stream);
}

extern "C" int jlm_edge_logits_perm(const jlm_segment *segs_host, int n_segs, const float *b2, const float *T, int ldt,
                                    const int *g0, const int *cnt, const int *cnt_idx, const int *wl, const int *wl_w,
                                    const int *wl_off, const int *wl_idx, int wl_base, const int *wl_out, float *edge, int beam,
                                    int n_groups, void *stream) {
    SegTable t;
    if (seg_table(segs_host, n_segs, t) || ldt % 4) return -1;
    if (n_groups <= 0) return 0;
    size_t lds = wl_lds_bytes(beam, ldt);
    if (lds > 160 * 1024) return -1;
    static JlmLdsGrant grant;
    if (int rc = jlm_grant_lds(grant, reinterpret_cast<const void *>(wordlist_kernel<0>), (int)lds)) return rc;
    hipLaunchKernelGGL(wordlist_kernel<0>, dim3(n_groups), dim3(WL_THREADS), lds, (hipStream_t)stream, t, b2, T, ldt,
                       g0, cnt, cnt_idx, wl, wl_w, wl_off, wl_idx, wl_base, wl_out, edge, nullptr, nullptr, nullptr, 0, beam);
    JLM_LAUNCH_CHECK();
    return 0;
}

// matrix-pipe form for single-segment models (jlm_gemm.hip)
extern "C" int jlm_wordlist_lse_mfma(const jlm_segment *seg_host, const float *b2, const float *T, int ldt, const int *g0,
                                     const int *cnt, const int *cnt_idx, const int *wl, const int *wl_off,
                                     const int *wl_idx, int wl_base, float *run_max, double *run_sum, double *lse,
                                     int merge, int beam, int n_groups, void *stream);

extern "C" int jlm_wordlist_lse(const jlm_segment *segs_host, int n_segs, const float *b2, const float *T, int ldt,
                                const int *g0, const int *cnt, const int *cnt_idx, const int *wl, const int *wl_off,
                                const int *wl_idx, int wl_base, float *run_max, double *run_sum, double *lse, int merge,
                                int beam, int n_groups, void *stream) {
    return jlm_wordlist_lse_perm(segs_host, n_segs, b2, T, ldt, g0, cnt, cnt_idx, wl, nullptr, wl_off, wl_idx, wl_base, run_max,
                                 run_sum, lse, merge, beam, n_groups, stream);
}

extern "C" int jlm_wordlist_lse_perm(const jlm_segment *segs_host, int n_segs, const float *b2, const float *T, int ldt,
                                     const int *g0, const int *cnt, const int *cnt_idx, const int *wl, const int *wl_w,
                                     const int *wl_off, const int *wl_idx, int wl_base, float *run_max, double *run_sum,
                                     double *lse, int merge, int beam, int n_groups, void *stream) {
    SegTable t;
    if (seg_table(segs_host, n_segs, t) || ldt % 4) return -1;
    if (n_groups <= 0) return 0;
    static int use_mfma = -1;
    if (use_mfma < 0) { const char *e = getenv("JLM_WORDLIST_MFMA"); use_mfma = e ? atoi(e) : 1; }
    if (!wl_w && use_mfma && n_segs == 1 && beam <= 64 && segs_host[0].k <= 256) {
        int r = jlm_wordlist_lse_mfma(segs_host, b2, T, ldt, g0, cnt, cnt_idx, wl, wl_off, wl_idx, wl_base, run_max, run_sum,
                                      lse, merge, beam, n_groups, stream);
        if (r != -2) return r;
    }
    size_t lds = wl_lds_bytes(beam, ldt);
    if (lds > 160 * 1024) return -1;
    static JlmLdsGrant grant;
    if (int rc = jlm_grant_lds(grant, reinterpret_cast<const void *>(wordlist_kernel<1>), (int)lds)) return rc;
    hipLaunchKernelGGL(wordlist_kernel<1>, dim3(n_groups), dim3(WL_THREADS), lds, (hipStream_t)stream, t, b2, T, ldt,
                       g0, cnt, cnt_idx, wl, wl_w, wl_off, wl_idx, wl_base, nullptr, nullptr, run_max, run_sum, lse, merge, beam);
    JLM_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ beam step
// One wave per sentence.  Candidate c = (node - first node of the frame) * beam
// + slot enumerates (node, previous hypothesis) pairs in the reference's
// generation order (outer loop nodes, inner loop previous paths:
// decoder/decoder.py:172-182), so a lexicographic (score, c) minimum reproduces
// Python's stable sort.
//
// Lane l owns the candidates c = l, l + 64, ...: it writes their keys (and their
// predecessor rows) to LDS and is the only lane that ever reads them back, so the
// selection needs no workgroup synchronisation: every lane carries the minimum of
// its own candidates in registers, one round = a wave-wide arg-min over those 64
// pairs, after which only the winner's owner strikes its key and rescans its
// <= ceil(C / 64) entries.  Lane r remembers the r-th winner and all K rows are
// written together at the end (one round of global latency instead of K).
//
// Fused K6 tail (st.lse_part != NULL, mode 0): the (max, sum exp) partial slices that
// the vocabulary kernel left for this sentence's rows of frame - 1 are folded here --
// 8 lanes per row, slices strided over them, 3 xor-shuffle steps -- instead of in a
// separate jlm_lse_combine launch; the rows' positions in the slices are
// live_base[(frame - 1, sentence)] + slot, recorded when the rows were listed.
// Wave-wide lexicographic minimum of (v, i), result in every lane.  DPP cross-lane moves (a few
// cycles each) instead of __shfl_xor, which is ds_bpermute on this ISA: three dependent LDS-crossbar
// round trips per step made one selection round 0.9 us (8.8 of the kernel's 19 us for beam = 10).
// Mirror steps leave every lane of a 16-lane row with the row's minimum; row_bcast15 / 31 carry it to
// the last row; lane 63 holds the wave's and is read back through SGPRs.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void argmin_dpp_step(double &v, int &i) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const int i2 = __builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, 0xf, false);
    const double v2 = __hiloint2double(hi2, lo2);
    if (v2 < v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}
__device__ __forceinline__ void wave_argmin(double &v, int &i) {
    argmin_dpp_step<0xB1, 0xf>(v, i);      // quad_perm(1,0,3,2)
    argmin_dpp_step<0x4E, 0xf>(v, i);      // quad_perm(2,3,0,1)
    argmin_dpp_step<0x141, 0xf>(v, i);     // row_half_mirror
    argmin_dpp_step<0x140, 0xf>(v, i);     // row_mirror
    argmin_dpp_step<0x142, 0xa>(v, i);     // row_bcast15 into rows 1 and 3
    argmin_dpp_step<0x143, 0xc>(v, i);     // row_bcast31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    i = __builtin_amdgcn_readlane(i, 63);
    v = __hiloint2double(hi, lo);
}

template <int MODE>
__global__ __launch_bounds__(64) void beam_step_kernel(jlm_lattice lat, jlm_beam_state st, int frame, int max_cands) {
    extern __shared__ __attribute__((aligned(16))) double keys[];   // [max_cands] | MODE 2: [n_frames*beam] | int [max_cands] | [beam] | int [n_frames] | winners: [beam] double, [beam] int
    const int s = blockIdx.x, lane = threadIdx.x;
    const int B = lat.n_sent, beam = lat.beam, rmax = B * beam;
    const int len = lat.sent_len[s];
    const int fs = frame * B + s;
    if (frame > len) { if (lane == 0) st.cnt[fs] = 0; return; }
    const int nb = lat.end_off[fs], ne = lat.end_off[fs + 1];
    const int gout = frame * rmax + s * beam;
    const double INF = __longlong_as_double(0x7ff0000000000000LL);
    double *Sarr = keys + max_cands;                                       // MODE 2 only
    int *gp_of = reinterpret_cast<int *>(Sarr + (MODE == 2 ? lat.n_frames * beam : 0));
    double *lse_new = reinterpret_cast<double *>(gp_of + ((max_cands + 1) & ~1));   // [beam]
    int *cnt_s = reinterpret_cast<int *>(lse_new + beam);                           // [n_frames] this sentence's counts
    // the winners of the selection rounds, in rank order (beams above 64: a lane writes out several of them at the end)
    double *win_v = reinterpret_cast<double *>(cnt_s + ((lat.n_frames + 1) & ~1));  // [beam]
    int *win_i = reinterpret_cast<int *>(win_v + beam);                             // [beam]
    int K;
    if (frame == 0) {
        K = 1;
    } else {
        // ---- fused fold of the previous frame's vocabulary partials (this sentence's rows)
        const bool fused = MODE == 0 && st.lse_part != nullptr;
        const int fprev = (frame - 1) * B + s;
        for (int f = lane; f < frame; f += 64) cnt_s[f] = st.cnt[f * B + s];
        const int kprev = st.cnt[fprev];
        if (fused && kprev > 0) {
            const int base = st.live_base[fprev];
            const float2 *part = reinterpret_cast<const float2 *>(st.lse_part);
            const int sub = lane & 7;
            for (int r0 = 0; r0 < kprev; r0 += 8) {
                const int r = r0 + (lane >> 3);
                float m = JLM_NEG_BIG;
                double sm = 0.0;
                if (r < kprev)
                    for (int p = sub; p < st.n_parts; p += 8) {
                        const float2 v = part[(size_t)p * st.ld_part + base + r];
                        const float mm = fmaxf(m, v.x);
                        sm = sm * (double)expf(m - mm) + (double)v.y * (double)expf(v.x - mm);
                        m = mm;
                    }
#pragma unroll
                for (int off = 4; off >= 1; off >>= 1) {
                    const float m2 = __shfl_xor(m, off);
                    const double s2 = __shfl_xor(sm, off);
                    const float mm = fmaxf(m, m2);
                    sm = sm * (double)expf(m - mm) + s2 * (double)expf(m2 - mm);
                    m = mm;
                }
                if (sub == 0 && r < kprev) {
                    double l = (double)m + log(sm);
                    // inf / nan (jlm_beam_state.flags): the batch is flagged and the value replaced by a finite one -- a NaN key would leave
                    // the selection rounds below without a winner and the next LSTM step with garbage row indices
                    if (!(fabs(l) < 1.0e300)) { if (st.flags) atomicOr(st.flags, 1); l = 1.0e30; }
                    lse_new[r] = l;
                    st.lse[(size_t)(frame - 1) * rmax + s * beam + r] = l;
                }
            }
        }
        const int C = (ne - nb) * beam;
        if (MODE == 2) {
            // S(g) = sum of the CURRENT log-normalisers of g's ancestors: every path is
            // re-scored from the head (decoder_dynamic.py:150-175), frame by frame.
            for (int f = 0; f < frame; ++f) {
                const int c = st.cnt[f * B + s];
                for (int l = lane; l < c; l += 64) {
                    const int g = f * rmax + s * beam + l;
                    const int p = st.bp[g];
                    double S = 0.0;
                    if (p >= 0) {
                        const int pf = p / rmax, pk = p - pf * rmax - s * beam;
                        S = Sarr[pf * beam + pk] + st.lse[p];
                    }
                    Sarr[f * beam + l] = S;
                }
                __syncthreads();
            }
        }
        __syncthreads();                                   // lse_new (and Sarr) visible to every lane
        // ---- keys of this lane's candidates, its running minimum.  Four candidates at a time with the
        //      loads of each dependency level issued together (start frame; then score / lse / edge of
        //      the predecessor): a round trip per level and group instead of three per candidate --
        //      these chains, not the selection, were 18 of the kernel's 22 us.
        int nvalid = 0;
        double bv = INF;
        int bi = 0x7fffffff;
        constexpr int U = 4, NREG = 8;                     // the lane's first NREG candidates live in registers
        double kreg[NREG];
#pragma unroll
        for (int j = 0; j < NREG; ++j) kreg[j] = INF;
        auto group = [&](int it, auto REG0) {              // candidates lane + 64 (U it + u), u < U
            constexpr int reg0 = decltype(REG0)::value;    // first register slot of the group, or -1: keys[] in LDS
            const int c0 = lane + 64 * U * it;
            int n[U], k[U], sf[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = min(c0 + 64 * u, C - 1);     // past the end: a duplicate, discarded below
                n[u] = nb + c / beam;
                k[u] = c % beam;
                sf[u] = lat.node_start[n[u]];
            }
            bool ok[U];
            int gp[U];
            double scv[U], lsv[U], ysv[U];
            float ev[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = (c0 + 64 * u < C) && k[u] < cnt_s[sf[u]];
                gp[u] = sf[u] * rmax + s * beam + (ok[u] ? k[u] : 0);
                ev[u] = st.edge[(size_t)n[u] * beam + k[u]];
                scv[u] = (MODE == 2) ? 0.0 : st.score[gp[u]];
                lsv[u] = (MODE == 1) ? 0.0 : st.lse[gp[u]];
                ysv[u] = (MODE == 2) ? st.ysum[gp[u]] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + 64 * u;
                if (c >= C) break;
                double sc = INF;
                if (ok[u]) {
                    const double e = (double)ev[u];
                    if (MODE == 0) sc = scv[u] + (((fused && sf[u] == frame - 1) ? lse_new[k[u]] : lsv[u]) - e);
                    else if (MODE == 1) sc = scv[u] - e;
                    else sc = (Sarr[sf[u] * beam + k[u]] + lsv[u]) - (ysv[u] + e);
                    ++nvalid;
                }
                if constexpr (reg0 >= 0) kreg[reg0 + u] = sc; else keys[c] = sc;
                gp_of[c] = ok[u] ? gp[u] : -1;
                if (sc < bv) { bv = sc; bi = c; }          // ascending c: ties keep the lower index
            }
        };
        group(0, std::integral_constant<int, 0>{});
        if (C > 64 * U) group(1, std::integral_constant<int, U>{});
        for (int it = 2; 64 * U * it < C; ++it) group(it, std::integral_constant<int, -1>{});
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) nvalid += __shfl_xor(nvalid, off);
        K = min(beam, nvalid);
        // ---- K rounds of selection, registers + the owner's private LDS entries only
        for (int r = 0; r < K; ++r) {
            double v = bv;
            int i = bi;
            wave_argmin(v, i);
            if (lane == 0) { win_v[r] = v; win_i[r] = i; }
            if ((i & 63) == lane) {                        // the owner strikes the winner and rescans its own entries
                const int jw = i >> 6;
                if (jw >= NREG) keys[i] = INF;
                bv = INF;
                bi = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < NREG; ++j) {
                    if (j == jw) kreg[j] = INF;
                    if (kreg[j] < bv) { bv = kreg[j]; bi = lane + 64 * j; }
                }
                for (int c = lane + 64 * NREG; c < C; c += 64) {
                    const double kv = keys[c];
                    if (kv < bv) { bv = kv; bi = c; }
                }
            }
        }
        __syncthreads();                                   // gp_of of other lanes' candidates, the winners
    }
    // ---- the K surviving hypotheses, one lane each (beams above 64: lane l takes ranks l, l + 64, ...)
    for (int r = lane; r < K; r += 64) {
        const int g = gout + r;
        if (frame == 0) {
            st.score[g] = 0.0;
            if (st.ysum) st.ysum[g] = 0.0;
            st.bp[g] = -1;
            st.node[g] = nb;
            st.word[g] = lat.node_word[nb];
        } else {
            const int wi = win_i[r];
            const int n = nb + wi / beam, k = wi % beam;
            const int gp = gp_of[wi];
            st.score[g] = win_v[r];
            if (MODE == 2) st.ysum[g] = st.ysum[gp] + (double)st.edge[(size_t)n * beam + k];
            st.bp[g] = gp;
            st.node[g] = n;
            st.word[g] = lat.node_word[n];
        }
    }
    int base = 0;
    if (lane == 0) {
        st.cnt[fs] = K;
        if (frame < len) {      // the last frame's LSTM step is never consumed (decoder.py:233-237)
            base = atomicAdd(&st.n_live[frame], K);
            if (st.live_base) st.live_base[fs] = base;
        }
    }
    base = __shfl(base, 0);
    if (frame < len)
        for (int r = lane; r < K; r += 64) st.live[(size_t)frame * rmax + base + r] = gout + r;
}

// ---- the same step for cells whose candidates do not fit one wave's LDS (round 6): the candidates are taken `cap` at a time -- keys and
// predecessor rows of ONE chunk in LDS, the K selection rounds of the kernel above on it -- and every chunk leaves its K winners (key, global
// candidate index, predecessor row) behind; a last selection over the chunk winners gives the frame's K.  The order is the lexicographic
// (key, candidate index) minimum throughout, so the result is the one-chunk kernel's (and Python's stable sort) whatever the cut.  Nothing
// here is on the decode's usual path: a cell needs more than ~13 k candidates (1 300 nodes at beam 10) to get here -- before this kernel such
// sentences left the batch for a host-side search (Decoder._decode_unpruned with the beam, DynamicDecoder._decode_host).
template <int MODE>
__global__ __launch_bounds__(64) void beam_step_chunked_kernel(jlm_lattice lat, jlm_beam_state st, int frame, int cap, int nch_max) {
    extern __shared__ __attribute__((aligned(16))) double keys[];   // [cap] | MODE 2: [n_frames*beam] | int [cap] | [beam] | int [n_frames] | winners | chunk winners
    const int s = blockIdx.x, lane = threadIdx.x;
    const int B = lat.n_sent, beam = lat.beam, rmax = B * beam;
    const int len = lat.sent_len[s];
    const int fs = frame * B + s;
    if (frame > len) { if (lane == 0) st.cnt[fs] = 0; return; }
    const int nb = lat.end_off[fs], ne = lat.end_off[fs + 1];
    const int gout = frame * rmax + s * beam;
    const double INF = __longlong_as_double(0x7ff0000000000000LL);
    double *Sarr = keys + cap;                                             // MODE 2 only
    int *gp_of = reinterpret_cast<int *>(Sarr + (MODE == 2 ? lat.n_frames * beam : 0));
    double *lse_new = reinterpret_cast<double *>(gp_of + ((cap + 1) & ~1));         // [beam]
    int *cnt_s = reinterpret_cast<int *>(lse_new + beam);                           // [n_frames]
    double *win_v = reinterpret_cast<double *>(cnt_s + ((lat.n_frames + 1) & ~1));  // [beam]
    int *win_i = reinterpret_cast<int *>(win_v + beam);                             // [beam]
    int *win_g = win_i + beam;                                                      // [beam]
    double *cw_v = reinterpret_cast<double *>(win_g + beam + (beam & 1));           // [nch_max * beam] chunk winners: key,
    int *cw_i = reinterpret_cast<int *>(cw_v + (size_t)nch_max * beam);             //   global candidate index,
    int *cw_g = cw_i + (size_t)nch_max * beam;                                      //   predecessor row
    int K;
    if (frame == 0) {
        K = 1;
    } else {
        const bool fused = MODE == 0 && st.lse_part != nullptr;
        const int fprev = (frame - 1) * B + s;
        for (int f = lane; f < frame; f += 64) cnt_s[f] = st.cnt[f * B + s];
        const int kprev = st.cnt[fprev];
        if (fused && kprev > 0) {
            const int base = st.live_base[fprev];
            const float2 *part = reinterpret_cast<const float2 *>(st.lse_part);
            const int sub = lane & 7;
            for (int r0 = 0; r0 < kprev; r0 += 8) {
                const int r = r0 + (lane >> 3);
                float m = JLM_NEG_BIG;
                double sm = 0.0;
                if (r < kprev)
                    for (int p = sub; p < st.n_parts; p += 8) {
                        const float2 v = part[(size_t)p * st.ld_part + base + r];
                        const float mm = fmaxf(m, v.x);
                        sm = sm * (double)expf(m - mm) + (double)v.y * (double)expf(v.x - mm);
                        m = mm;
                    }
#pragma unroll
                for (int off = 4; off >= 1; off >>= 1) {
                    const float m2 = __shfl_xor(m, off);
                    const double s2 = __shfl_xor(sm, off);
                    const float mm = fmaxf(m, m2);
                    sm = sm * (double)expf(m - mm) + s2 * (double)expf(m2 - mm);
                    m = mm;
                }
                if (sub == 0 && r < kprev) {
                    double l = (double)m + log(sm);
                    if (!(fabs(l) < 1.0e300)) { if (st.flags) atomicOr(st.flags, 1); l = 1.0e30; }
                    lse_new[r] = l;
                    st.lse[(size_t)(frame - 1) * rmax + s * beam + r] = l;
                }
            }
        }
        const int C = (ne - nb) * beam;
        if (MODE == 2) {
            for (int f = 0; f < frame; ++f) {
                const int c = st.cnt[f * B + s];
                for (int l = lane; l < c; l += 64) {
                    const int g = f * rmax + s * beam + l;
                    const int p = st.bp[g];
                    double S = 0.0;
                    if (p >= 0) {
                        const int pf = p / rmax, pk = p - pf * rmax - s * beam;
                        S = Sarr[pf * beam + pk] + st.lse[p];
                    }
                    Sarr[f * beam + l] = S;
                }
                __syncthreads();
            }
        }
        __syncthreads();
        const int nch = (C + cap - 1) / cap;               // <= nch_max: the launcher sized the chunk winners for the batch's largest cell
        int nvalid_all = 0;
        for (int ch = 0; ch < nch; ++ch) {
            const int c_lo = ch * cap, Cc = min(C - c_lo, cap);
            int nvalid = 0;
            double bv = INF;
            int bi = 0x7fffffff;                           // LOCAL candidate index inside the chunk
            for (int lc = lane; lc < Cc; lc += 64) {
                const int c = c_lo + lc;
                const int n = nb + c / beam, k = c % beam;
                const int sf = lat.node_start[n];
                const bool ok = k < cnt_s[sf];
                const int gp = sf * rmax + s * beam + (ok ? k : 0);
                double sc = INF;
                if (ok) {
                    const double e = (double)st.edge[(size_t)n * beam + k];
                    if (MODE == 0) sc = st.score[gp] + (((fused && sf == frame - 1) ? lse_new[k] : st.lse[gp]) - e);
                    else if (MODE == 1) sc = st.score[gp] - e;
                    else sc = (Sarr[sf * beam + k] + st.lse[gp]) - (st.ysum[gp] + e);
                    ++nvalid;
                }
                keys[lc] = sc;
                gp_of[lc] = ok ? gp : -1;
                if (sc < bv) { bv = sc; bi = lc; }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) nvalid += __shfl_xor(nvalid, off);
            nvalid_all += nvalid;
            const int Kc = min(beam, nvalid);
            __syncthreads();                               // gp_of of other lanes' candidates
            for (int r = 0; r < beam; ++r) {
                if (r < Kc) {
                    double v = bv;
                    int i = bi;
                    wave_argmin(v, i);                     // (local indices rise with the global ones: the same order)
                    if (lane == 0) { cw_v[ch * beam + r] = v; cw_i[ch * beam + r] = c_lo + i; cw_g[ch * beam + r] = gp_of[i]; }
                    if ((i & 63) == lane) {
                        keys[i] = INF;
                        bv = INF;
                        bi = 0x7fffffff;
                        for (int lc = lane; lc < Cc; lc += 64) {
                            const double kv = keys[lc];
                            if (kv < bv) { bv = kv; bi = lc; }
                        }
                    }
                } else if (lane == 0) {
                    cw_v[ch * beam + r] = INF; cw_i[ch * beam + r] = 0x7fffffff; cw_g[ch * beam + r] = -1;
                }
            }
            __syncthreads();                               // the chunk's keys are dead: the next chunk overwrites them
        }
        K = min(beam, nvalid_all);
        // ---- the frame's K out of the chunk winners: lane l owns entries l, l + 64, ...; a candidate index is unique, so the owner of
        //      a winner is the lane whose best entry carries it
        const int E = nch * beam;
        double bv = INF;
        int bi = 0x7fffffff, be = -1;
        for (int e = lane; e < E; e += 64) {
            const double v = cw_v[e];
            const int i = cw_i[e];
            if (v < bv || (v == bv && i < bi)) { bv = v; bi = i; be = e; }
        }
        for (int r = 0; r < K; ++r) {
            double v = bv;
            int i = bi;
            wave_argmin(v, i);
            if (bi == i && be >= 0) {                      // this lane owns the winner
                win_v[r] = v; win_i[r] = i; win_g[r] = cw_g[be];
                cw_v[be] = INF; cw_i[be] = 0x7fffffff;
                bv = INF; bi = 0x7fffffff; be = -1;
                for (int e = lane; e < E; e += 64) {
                    const double v2 = cw_v[e];
                    const int i2 = cw_i[e];
                    if (v2 < bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; be = e; }
                }
            }
        }
        __syncthreads();
    }
    for (int r = lane; r < K; r += 64) {
        const int g = gout + r;
        if (frame == 0) {
            st.score[g] = 0.0;
            if (st.ysum) st.ysum[g] = 0.0;
            st.bp[g] = -1;
            st.node[g] = nb;
            st.word[g] = lat.node_word[nb];
        } else {
            const int wi = win_i[r];
            const int n = nb + wi / beam, k = wi % beam;
            const int gp = win_g[r];
            st.score[g] = win_v[r];
            if (MODE == 2) st.ysum[g] = st.ysum[gp] + (double)st.edge[(size_t)n * beam + k];
            st.bp[g] = gp;
            st.node[g] = n;
            st.word[g] = lat.node_word[n];
        }
    }
    int base = 0;
    if (lane == 0) {
        st.cnt[fs] = K;
        if (frame < len) {
            base = atomicAdd(&st.n_live[frame], K);
            if (st.live_base) st.live_base[fs] = base;
        }
    }
    base = __shfl(base, 0);
    if (frame < len)
        for (int r = lane; r < K; r += 64) st.live[(size_t)frame * rmax + base + r] = gout + r;
}

// LDS of one sentence's wave: keys [max_cands] f64 | mode 2: S [n_frames x beam] f64 | predecessor rows [max_cands] i32 |
// folded log-normalisers [beam] f64 | the sentence's counts [n_frames] i32 | winners [beam] f64 + [beam] i32
static size_t beam_step_lds_bytes(int beam, int n_frames, int mode, int max_cands) {
    return (size_t)max_cands * sizeof(double) + (mode == 2 ? (size_t)n_frames * beam * sizeof(double) : 0) +
           (size_t)((max_cands + 1) & ~1) * sizeof(int) + (size_t)beam * sizeof(double) + (size_t)((n_frames + 1) & ~1) * sizeof(int) +
           (size_t)beam * (sizeof(double) + sizeof(int)) + 8;
}

// the chunked kernel's: keys / predecessor rows of one chunk [cap], the chunk winners [nch x beam] (f64 + 2 x i32), one more i32 per rank
static size_t beam_step_chunked_lds_bytes(int beam, int n_frames, int mode, int cap, int nch) {
    return beam_step_lds_bytes(beam, n_frames, mode, cap) + (size_t)(beam + (beam & 1)) * sizeof(int) +
           (size_t)nch * beam * (sizeof(double) + 2 * sizeof(int));
}
// chunk size of the chunked kernel: half of what one wave's LDS would hold in one piece, the other half is for the chunk winners
static int beam_step_chunk_cap(int beam, int n_frames, int mode) {
    const size_t fixed = beam_step_lds_bytes(beam, n_frames, mode, 0);
    if (fixed + 256 * 12 > 160 * 1024) return 0;
    size_t c = (160 * 1024 - fixed) / 12 / 2;
    return (int)(c / 256 * 256);
}
static int beam_step_one_chunk_max(int beam, int n_frames, int mode) {
    const size_t fixed = beam_step_lds_bytes(beam, n_frames, mode, 0);
    if (fixed + 256 * 12 > 160 * 1024) return 0;
    size_t c = (160 * 1024 - fixed) / 12;
    c = c / 256 * 256;
    while (c > 0 && beam_step_lds_bytes(beam, n_frames, mode, (int)c) > 160 * 1024) c -= 256;
    return (int)c;
}

// Largest max_cands (candidates of one (frame, sentence) cell = nodes ending there x beam, as the plans round it: a
// multiple of 256) that jlm_beam_step accepts for this beam / frame count / mode; 0: none.  Round 6: cells above what one wave's
// LDS holds in one piece (~13 k candidates at beam 10) are selected chunk by chunk (beam_step_chunked_kernel), so the figure is what
// the chunk winners leave room for -- hundreds of thousands at beam 10, capped at 2^22.  Callers still route sentences with a larger cell
// to their host-side search instead of failing the batch (jlm_amd/decoder.py, decoder_dynamic.py).
extern "C" int jlm_beam_step_max_cands(int beam, int n_frames, int mode) {
    if (beam < 1 || beam > JLM_MAX_BEAM || n_frames < 1 || mode < 0 || mode > 2) return 0;
    const int one = beam_step_one_chunk_max(beam, n_frames, mode);
    const int cap = beam_step_chunk_cap(beam, n_frames, mode);
    if (one <= 0) return 0;
    if (cap <= 0) return one;
    const size_t base = beam_step_chunked_lds_bytes(beam, n_frames, mode, cap, 0);
    if (base >= 160 * 1024) return one;
    const size_t nch = (160 * 1024 - base) / ((size_t)beam * 16);
    size_t total = nch * (size_t)cap;
    if (total > (1u << 22)) total = 1u << 22;
    total = total / 256 * 256;
    return total > (size_t)one ? (int)total : one;
}

extern "C" int jlm_beam_step(const jlm_lattice *lat_host, const jlm_beam_state *st_host, int frame, int mode,
                             int max_cands, void *stream) {
    const jlm_lattice lat = *lat_host;
    const jlm_beam_state st = *st_host;
    if (lat.n_sent <= 0) return 0;
    if (mode == 2 && !st.ysum) return -1;
    if (mode < 0 || mode > 2) return -1;
    if (lat.beam < 1 || lat.beam > JLM_MAX_BEAM) return -1;
    if (st.lse_part && (!st.live_base || st.n_parts < 1 || mode != 0)) return -1;
    if (max_cands < 1) max_cands = 1;
    const size_t lds = beam_step_lds_bytes(lat.beam, lat.n_frames, mode, max_cands);
    // JLM_BEAM_CHUNK=<candidates>: every launch through the chunked kernel with that chunk size (tests: ordinary cells in several chunks)
    static const int chunk_env = getenv("JLM_BEAM_CHUNK") ? atoi(getenv("JLM_BEAM_CHUNK")) : 0;
    if (lds > 160 * 1024 || chunk_env > 0) {
        const int cap = chunk_env > 0 ? chunk_env : beam_step_chunk_cap(lat.beam, lat.n_frames, mode);
        if (cap <= 0) return -1;
        const int nch = (max_cands + cap - 1) / cap;
        const size_t lds_c = beam_step_chunked_lds_bytes(lat.beam, lat.n_frames, mode, cap, nch);
        if (lds_c > 160 * 1024) return -1;
        const void *fc = mode == 0 ? (const void *)beam_step_chunked_kernel<0>
                       : mode == 1 ? (const void *)beam_step_chunked_kernel<1> : (const void *)beam_step_chunked_kernel<2>;
        static JlmLdsGrant grant_c[3];
        if (lds_c > 64 * 1024)
            if (int rc = jlm_grant_lds(grant_c[mode], fc, (int)lds_c)) return rc;
        if (mode == 0) hipLaunchKernelGGL(beam_step_chunked_kernel<0>, dim3(lat.n_sent), dim3(64), lds_c, (hipStream_t)stream, lat, st, frame, cap, nch);
        else if (mode == 1) hipLaunchKernelGGL(beam_step_chunked_kernel<1>, dim3(lat.n_sent), dim3(64), lds_c, (hipStream_t)stream, lat, st, frame, cap, nch);
        else hipLaunchKernelGGL(beam_step_chunked_kernel<2>, dim3(lat.n_sent), dim3(64), lds_c, (hipStream_t)stream, lat, st, frame, cap, nch);
        JLM_LAUNCH_CHECK();
        return 0;
    }
    const void *fn = mode == 0 ? (const void *)beam_step_kernel<0>
                   : mode == 1 ? (const void *)beam_step_kernel<1> : (const void *)beam_step_kernel<2>;
    static JlmLdsGrant grant[3];
    if (lds > 64 * 1024)
        if (int rc = jlm_grant_lds(grant[mode], fn, (int)lds)) return rc;
    if (mode == 0) hipLaunchKernelGGL(beam_step_kernel<0>, dim3(lat.n_sent), dim3(64), lds, (hipStream_t)stream, lat, st, frame, max_cands);
    else if (mode == 1) hipLaunchKernelGGL(beam_step_kernel<1>, dim3(lat.n_sent), dim3(64), lds, (hipStream_t)stream, lat, st, frame, max_cands);
    else hipLaunchKernelGGL(beam_step_kernel<2>, dim3(lat.n_sent), dim3(64), lds, (hipStream_t)stream, lat, st, frame, max_cands);
    JLM_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ backtrace
__global__ void backtrace_kernel(jlm_lattice lat, jlm_beam_state st, int *out_nodes, int *out_len, double *out_score,
                                 int stride) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int B = lat.n_sent, beam = lat.beam, rmax = B * beam;
    if (idx >= rmax) return;
    const int s = idx / beam, r = idx % beam;
    const int len = lat.sent_len[s];
    const int c = st.cnt[len * B + s];
    if (r >= c) { out_len[idx] = 0; out_score[idx] = 0.0; return; }
    int g = len * rmax + s * beam + r;
    out_score[idx] = st.score[g];
    int d = 0;
    while (g >= 0 && d < stride) {
        out_nodes[(size_t)idx * stride + d] = st.node[g];
        ++d;
        g = st.bp[g];
    }
    out_len[idx] = d;
}

// Round 6: one WAVE per sentence.  The thread-per-path walk above is a chain of (sentence length + 1) dependent global loads per path
// (12 us per batch at the headline shape).  A sentence's whole back-pointer table is small -- (length + 1) x beam rows, 210 at the headline
// shape -- so the wave loads it ONCE (row e = frame x beam + rank in lane e % 64, register e / 64: node id and predecessor as a local row
// number), one round trip, and lane r then walks path r through the other lanes' registers with wavefront shuffles (ds_bpermute), no memory
// in the loop; the node ids are stored as they are met.  Shapes it hosts: beam <= 64 (a lane per path), n_frames x beam <= 64 NREG.
template <int NREG>
__global__ __launch_bounds__(64) void backtrace_wave_kernel(jlm_lattice lat, jlm_beam_state st, int *out_nodes, int *out_len, double *out_score,
                                                            int stride) {
    const int s = blockIdx.x, lane = threadIdx.x;
    const int B = lat.n_sent, beam = lat.beam, rmax = B * beam;
    const int len = lat.sent_len[s];
    const int E = (len + 1) * beam;
    int nd[NREG], pl[NREG];
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
        const int e = lane + 64 * j;
        const int f = e / beam, k = e - f * beam;
        const bool in = e < E;
        const size_t g = (size_t)(in ? f : 0) * rmax + s * beam + (in ? k : 0);
        // (rows past a frame's count hold whatever an earlier batch left there: no surviving path points at them)
        const int n = st.node[g], p = st.bp[g];
        nd[j] = in ? n : -1;
        int q = -1;
        if (in && p >= 0) { const int pf = p / rmax; q = pf * beam + (p - pf * rmax - s * beam); }
        pl[j] = q;
    }
    const int c = st.cnt[len * B + s];
    const bool valid = lane < beam && lane < c;
    const int idx = s * beam + lane;
    int e = valid ? len * beam + lane : -1, d = 0;
    const int steps = min(stride, len + 1);              // (a path visits every frame at most once)
    for (int t = 0; t < steps; ++t) {
        const int src = e & 63, jr = e >> 6;             // every lane takes part in the shuffles, whatever its own state
        int nv = -1, pv = -1;
#pragma unroll
        for (int j = 0; j < NREG; ++j) {
            const int a = __shfl(nd[j], src), b2 = __shfl(pl[j], src);
            if (jr == j) { nv = a; pv = b2; }
        }
        if (e >= 0) {
            out_nodes[(size_t)idx * stride + d] = nv;
            ++d;
            e = pv;
        }
    }
    if (lane < beam) {
        out_len[idx] = valid ? d : 0;
        out_score[idx] = valid ? st.score[(size_t)len * rmax + s * beam + lane] : 0.0;
    }
}

extern "C" int jlm_backtrace(const jlm_lattice *lat_host, const jlm_beam_state *st_host, int *out_nodes, int *out_len,
                             double *out_score, int stride, void *stream) {
    const jlm_lattice lat = *lat_host;
    const jlm_beam_state st = *st_host;
    const int total = lat.n_sent * lat.beam;
    if (total <= 0) return 0;
    // JLM_BACKTRACE_WAVE=0: the thread-per-path kernel for every shape (A/B, tests)
    static const int wave_env = getenv("JLM_BACKTRACE_WAVE") ? atoi(getenv("JLM_BACKTRACE_WAVE")) : 1;
    const long rows = (long)lat.n_frames * lat.beam;
    if (wave_env && lat.beam <= 64 && rows <= 64 * 16) {
        const dim3 grid(lat.n_sent), block(64);
        if (rows <= 64 * 4) hipLaunchKernelGGL(backtrace_wave_kernel<4>, grid, block, 0, (hipStream_t)stream, lat, st, out_nodes, out_len, out_score, stride);
        else if (rows <= 64 * 8) hipLaunchKernelGGL(backtrace_wave_kernel<8>, grid, block, 0, (hipStream_t)stream, lat, st, out_nodes, out_len, out_score, stride);
        else hipLaunchKernelGGL(backtrace_wave_kernel<16>, grid, block, 0, (hipStream_t)stream, lat, st, out_nodes, out_len, out_score, stride);
        JLM_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(backtrace_kernel, dim3((total + 127) / 128), dim3(128), 0, (hipStream_t)stream, lat, st,
                       out_nodes, out_len, out_score, stride);
    JLM_LAUNCH_CHECK();
    return 0;
}

// --------------------------------------------------------------- row softmax
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float *y, float *pred, int ld, int n_cols, int self_norm) {
    __shared__ float redm[4], reds[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    const float *yr = y + (size_t)r * ld;
    float *pr = pred + (size_t)r * ld;
    if (self_norm) {
        for (int c = tid; c < n_cols; c += 256) pr[c] = expf(yr[c]);
        return;
    }
    float m = JLM_NEG_BIG;
    for (int c = tid; c < n_cols; c += 256) m = fmaxf(m, yr[c]);
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((tid & 63) == 0) redm[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    float s = 0.0f;
    for (int c = tid; c < n_cols; c += 256) s += expf(yr[c] - m);
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if ((tid & 63) == 0) reds[tid >> 6] = s;
    __syncthreads();
    s = (reds[0] + reds[1]) + (reds[2] + reds[3]);
    const float inv = 1.0f / s;
    for (int c = tid; c < n_cols; c += 256) pr[c] = expf(yr[c] - m) * inv;
}

extern "C" int jlm_softmax_rows(const float *y, float *pred, int ld, int n_rows, int n_cols, int self_norm, void *stream) {
    if (n_rows <= 0) return 0;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, y, pred, ld, n_cols, self_norm);
    JLM_LAUNCH_CHECK();
    return 0;
}

extern "C" int jlm_abi_version(void) { return JLM_ABI_VERSION; }

extern "C" int jlm_device_arch(int dev, char *buf, int buflen) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return (int)e;
    int i = 0;
    for (; i < buflen - 1 && p.gcnArchName[i]; ++i) buf[i] = p.gcnArchName[i];
    buf[i] = 0;
    return 0;
}
