// jlm_decode.hip -- the frame loop of a batched decode as one C-ABI call (jlm_decode_frames, include/jlm_hip.h).
// Host code only: it enqueues the launchers of jlm_gemm.hip / jlm_split.hip / jlm_beam.hip in the order
// jlm_amd/engine.py documents, so a batch costs one FFI call instead of ~170.
#include <hip/hip_runtime.h>
#include "../../include/jlm_hip.h"

namespace {

// fork/join events for the edge-logit side stream.  A wait binds to the record that precedes it at
// enqueue time, so a rotating pool per device is enough; it is sized so that an event is not recorded
// again before the batches that could still be waiting on it (two or three in flight, two events per
// frame) have long been enqueued.
struct EventPool {
    static constexpr int N = 512, MAX_DEV = 16;
    hipEvent_t ev[MAX_DEV][N];
    int next[MAX_DEV] = {0};
    bool ready[MAX_DEV] = {false};
    hipError_t get(hipEvent_t *out) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= MAX_DEV) return hipErrorInvalidDevice;
        if (!ready[dev]) {
            for (int i = 0; i < N; ++i) {
                e = hipEventCreateWithFlags(&ev[dev][i], hipEventDisableTiming);
                if (e != hipSuccess) return e;
            }
            ready[dev] = true;
        }
        *out = ev[dev][next[dev]];
        next[dev] = (next[dev] + 1) % N;
        return hipSuccess;
    }
};
thread_local EventPool g_events;

#define JLM_TRY(x) do { int rc_ = (x); if (rc_ != 0) return rc_; } while (0)
#define JLM_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

}  // namespace

// ABI 11: the model's mixed segments are mx6 rows (FP6 cross-term planes, csrc/jlm_mx6_body.h) when their s8 is 0 -- the hypothesis
// rows are then packed by jlm_pack_t_mixed6; a model mixes the two formats in no launch (jlm_vocab_lse_mixed: -2)
static inline bool jlm_model_mx6(const jlm_decode_model *m) {
    if (!m->mixed_segs || !m->mixed_s8) return false;
    for (int i = 0; i < m->n_segs; ++i)
        if (m->mixed_segs[i].B) return m->mixed_s8[i] == 0.0f;
    return false;
}
#define JLM_PACK_T_MIXED(m) (jlm_model_mx6(m) ? jlm_pack_t_mixed6 : jlm_pack_t_mixed)

extern "C" int jlm_decode_frames(const jlm_decode_model *m, const jlm_decode_plan *p, const jlm_lattice *lat,
                                 const jlm_beam_state *st_in, void *stream, void *side_stream, void *const *events) {
    const int B = lat->n_sent, beam = lat->beam, F = lat->n_frames;
    const int rmax = B * beam;
    const bool dynamic = p->kind == 2, select = p->kind == 1, full = p->kind == 0;
    const int mode = m->self_norm ? 1 : (dynamic ? 2 : 0);
    const bool wl_split = m->split_segs != nullptr && m->n_segs == 1 && beam <= 64;
    // a segment with k > 256 (untied models: k = H) is outside the rows-stationary normalisers: one tile GEMM per segment
    // with a per-tile log-sum-exp epilogue (jlm_vocab_lse_partials), its slices folded by the next frame's beam step
    bool tile_form = false;
    if (full && !m->self_norm)
        for (int i = 0; i < m->n_segs; ++i)
            if (m->segs[i].k > 256) tile_form = true;
    jlm_beam_state st = *st_in;
    hipStream_t main_s = (hipStream_t)stream, side_s = events ? nullptr : (hipStream_t)side_stream;
    // events != NULL: JLM_EVENTS_PER_FRAME timing events per frame, recorded on `stream` (no side stream then, so that
    // every bracket holds exactly the kernels it names)
    auto stamp = [&](int f, int i) -> int {
        if (!events) return 0;
        return (int)hipEventRecord((hipEvent_t)events[(size_t)f * JLM_EVENTS_PER_FRAME + i], main_s);
    };
    hipEvent_t join = nullptr;
    int pending_parts = 0;

    // word-list normaliser: split rows (deep gather ring) for lists of 128 .. 4064 words, else the f32 kernel
    auto wl_lse = [&](const int *g0, const int *cidx, const int *words, const int *off, const int *idx, int base, int merge,
                      int n_groups, int max_words) -> int {
        if (wl_split && max_words >= 128 && max_words <= 4064) {
            int r = jlm_wordlist_lse_split(m->split_segs, m->split_t_scale[0], m->split_descale[0], m->b2, p->T, m->ldt, g0,
                                           st.cnt, cidx, words, off, idx, base, max_words, p->run_max, p->run_sum, st.lse,
                                           merge, beam, n_groups, stream);
            if (r != -2) return r;
        }
        return jlm_wordlist_lse(m->segs, m->n_segs, m->b2, p->T, m->ldt, g0, st.cnt, cidx, words, off, idx, base,
                                p->run_max, p->run_sum, st.lse, merge, beam, n_groups, stream);
    };
    // reference-compatibility mode of the incremental decoder on segmented models (jlm_decode_plan.di_wwords)
    const bool perm = dynamic && p->di_wwords && p->sg_wword;

#ifdef JLM_PROBE_SKIP
    // measurement builds only (tools/probes/skip_kernel.sh): JLM_SKIP=<bits> leaves launches out -- 1 edge logits, 2 T projection,
    // 4 LSTM step, 8 beam step, 16 vocabulary kernel, 32 packing of the T rows.  Results are wrong; the step time says what is on the critical path.
    // JLM_SKIP_AFTER=n: the first n frame loops run complete, so that the buffers a skipped kernel would have written hold
    // data of the usual kind (zeros in the operands of the matrix kernels draw less power: the clock rises and the probe lies)
    static const int skip_bits = [] { const char *e = getenv("JLM_SKIP"); return e ? atoi(e) : 0; }();
    static const int skip_after = [] { const char *e = getenv("JLM_SKIP_AFTER"); return e ? atoi(e) : 0; }();
    static int n_loops = 0;
    const int skip = (n_loops++ >= skip_after) ? skip_bits : 0;
#define JLM_SKIPPED(bit) (skip & (bit))
#else
#define JLM_SKIPPED(bit) 0
#endif
    for (int f = 0; f < F; ++f) {
        if (join) {
            JLM_HIP(hipStreamWaitEvent(main_s, join, 0));
            join = nullptr;
        }
        JLM_TRY(stamp(f, 0));
        if (dynamic && !m->self_norm && f >= 2) {
            int r = -2;
            if (wl_split && p->dd_max <= 128)
                r = jlm_wordlist_merge_split(m->split_segs, m->split_t_scale[0], m->split_descale[0], m->b2, p->T, m->ldt,
                                             st.cnt, B, beam, f - 1, p->dd_words, p->dd_off, f * B, p->dd_max, p->run_max,
                                             p->run_sum, st.lse, stream);
            if (r == -2) r = wl_lse(p->g0, p->cidx, p->dd_words, p->dd_off, p->sidx, f * B, 1, (f - 1) * B, p->dd_max);
            JLM_TRY(r);
        }
        JLM_TRY(stamp(f, 1));
        st.lse_part = pending_parts ? p->part : nullptr;
        st.ld_part = rmax;
        st.n_parts = pending_parts;
        if (!JLM_SKIPPED(8)) JLM_TRY(jlm_beam_step(lat, &st, f, mode, p->max_cands, stream));
        JLM_TRY(stamp(f, 2));
        pending_parts = 0;
        if (f == F - 1) break;
        const int *rows = st.live + (size_t)f * rmax;
        const int *ndev = st.n_live + f;
        if (JLM_SKIPPED(4)) {
        } else if (m->split_lstm && m->wt8)
            JLM_TRY(jlm_lstm_step_xg(p->h, p->c, m->H, p->h, p->c, rows, st.bp, st.word, m->wt8, m->xgate8, m->H,
                                     m->gate_descale, m->h_scale, m->untied ? p->T : nullptr, rmax, ndev, stream));
        else if (m->split_lstm)
            return -2;              // (a split-row model always carries wt8 / xgate8: DeviceModel builds them together)
        else
            JLM_TRY(jlm_lstm_step((const float *)p->h, p->c, m->H, (float *)p->h, p->c, rows, st.bp, st.word, m->emb,
                                  m->ld_emb, m->wt, m->gate_bias, m->kpad, m->H, m->E, rmax, ndev, stream));
        JLM_TRY(stamp(f, 3));
        if (!m->untied && !JLM_SKIPPED(2)) {
            if (m->split_lstm)
                JLM_TRY(jlm_gemm_nt_split(p->h, m->H, rows, m->pmt_split, m->H, nullptr, p->T, m->ldt, rows, nullptr,
                                          m->t_descale, rmax, m->n_t, m->H, ndev, stream));
            else
                JLM_TRY(jlm_gemm_nt((const float *)p->h, m->H, rows, m->pmt, m->H, nullptr, p->T, m->ldt, rows, nullptr, rmax,
                                    m->n_t, m->H, ndev, stream));
        }
        // segments of the full-vocabulary normaliser on mixed rows: this frame's live rows are packed once, here, behind T
        // (round 5: an untied model at H = 512 -- T is the state's f32 copy, one segment of k = 512 -- runs jlm_vocab_lse_mixed's wide
        //  one-row-set form instead of the tile GEMM below when its vocabulary matrix exists as mixed rows)
        bool hybrid = false, all_mixed = false;
        const bool untied_mixed = tile_form && m->untied && m->n_segs == 1 && m->mixed_segs && m->mixed_segs[0].B && p->Tm;
        if (full && !m->self_norm && ((!tile_form && m->mixed_segs && m->split_segs && p->Tm) || untied_mixed)) {
            jlm_segment only[JLM_MAX_SEGMENTS];
            float only_ts[JLM_MAX_SEGMENTS];
            int n_only = 0;
            for (int i = 0; i < m->n_segs; ++i)
                if (m->mixed_segs[i].B) { only[n_only] = m->mixed_segs[i]; only_ts[n_only++] = m->mixed_t_scale[i]; }
            if (n_only) {
                if (jlm_mixed_t_stride(only, n_only) != p->ld_tm) return -1;
                if (!JLM_SKIPPED(32)) JLM_TRY(JLM_PACK_T_MIXED(m)(only, only_ts, n_only, p->T, m->ldt, rows, f == 0 ? B : rmax, ndev, p->Tm, p->ld_tm, stream));
                hybrid = true;
                all_mixed = n_only == m->n_segs;
                // (ABI 10) a segment whose head stays on split rows: the launch over both formats
                if (all_mixed && m->mixed_head_split && m->split_segs)
                    for (int i = 0; i < m->n_segs; ++i)
                        if (m->mixed_head_split[i] > 0) all_mixed = false;
            }
        }
        const int cell = f * B;
        void *est = stream;
        if (side_s) {          // the edge logits need only T: they run beside the normaliser
            hipEvent_t fork;
            JLM_HIP(g_events.get(&fork));
            JLM_HIP(hipEventRecord(fork, main_s));
            JLM_HIP(hipStreamWaitEvent(side_s, fork, 0));
            est = side_stream;
        }
        if (!JLM_SKIPPED(1)) JLM_TRY(jlm_edge_logits_perm(m->segs, m->n_segs, m->b2, p->T, m->ldt, p->g0 + cell, st.cnt, p->cidx + cell, p->sg_word,
                                     perm ? p->sg_wword : nullptr, p->sg_off, p->sidx, cell, p->sg_node, p->edge, beam, B, est));
        if (side_s) {
            JLM_HIP(g_events.get(&join));
            JLM_HIP(hipEventRecord(join, side_s));
        }
        JLM_TRY(stamp(f, 4));
        if (!m->self_norm && !JLM_SKIPPED(16)) {
            if (perm)
                JLM_TRY(jlm_wordlist_lse_perm(m->segs, m->n_segs, m->b2, p->T, m->ldt, p->g0 + cell, st.cnt, p->cidx + cell,
                                              p->di_words, p->di_wwords, p->di_off, p->di_idx, 2 * cell, p->run_max, p->run_sum,
                                              st.lse, 0, beam, B, stream));
            else if (dynamic)
                JLM_TRY(wl_lse(p->g0 + cell, p->cidx + cell, p->di_words, p->di_off, p->di_idx, 2 * cell, 0, B, p->di_max));
            else if (select)
                JLM_TRY(wl_lse(p->g0 + cell, p->cidx + cell, p->vs_words, p->vs_off, p->sidx, 0, 0, B, p->vs_max));
            else if (tile_form && !all_mixed) {
                int n_parts = 0;
                for (int i = 0; i < m->n_segs; ++i) {
                    const jlm_segment &sg = m->segs[i];
                    // capacity is checked BEFORE the launch that would write the slices (one per 128-word tile)
                    if (n_parts + (sg.v_end - sg.v_start + 127) / 128 > p->max_parts) return -1;
                    int r = (m->untied && m->untied_split && m->split_lstm)
                                ? jlm_vocab_lse_partials_split(m->untied_split, m->H, sg.v_end - sg.v_start, m->H, p->h, m->H, rows,
                                                               m->b2 + sg.v_start, m->untied_descale, p->part, rmax, n_parts,
                                                               rmax, ndev, stream)
                                : jlm_vocab_lse_partials(sg.B, sg.ldb, sg.v_end - sg.v_start, sg.k, p->T + sg.t_off, m->ldt, rows,
                                                         m->b2 + sg.v_start, p->part, rmax, n_parts, rmax, ndev, stream);
                    if (r < 0) return r;
                    n_parts += r;
                }
                if (n_parts > p->max_parts) return -1;
                pending_parts = n_parts;
            } else {
                // frame 0 has one row per sentence: the bound lets the kernel cut the vocabulary into more ranges
                const int bound = f == 0 ? B : rmax;
                // the share of the chip this batch's normaliser takes: its range count is capped so that ranges x row tiles
                // (one 8-wave workgroup per CU each) fill that share; the launcher rounds down to a multiple of 8 ranges
                int cap = p->max_parts;
                if (p->lse_cu_share_pct > 0 && p->lse_cu_share_pct < 100) {
                    const int n_ptiles = (bound + 255) / 256;
                    int c = 256 * p->lse_cu_share_pct / 100 / n_ptiles;
                    if (c < 1) c = 1;
                    c += m->n_segs - 1;            // slices = columns + the segment boundaries columns straddle
                    if (c < cap) cap = c;
                }
                int r = -2;
                if (all_mixed)      // every segment on mixed rows (lse_fixed_ref: without a running maximum where the kernel has such a form)
                    r = (m->lse_fixed_ref ? jlm_vocab_lse_mixed_fr : jlm_vocab_lse_mixed)(m->mixed_segs, m->mixed_descale, m->mixed_s8, m->mixed_bias2,
                                                                                         m->n_segs, p->Tm, p->ld_tm, p->part, rmax, cap, bound, ndev, stream);
                else if (hybrid)    // -2: a shape the hybrid kernel does not host -- the split rows of every segment exist
                    r = jlm_vocab_lse_hybrid(m->split_segs, m->split_t_scale, m->split_descale, m->split_bias_col, m->mixed_segs,
                                             m->mixed_descale, m->mixed_s8, m->mixed_head_split, m->n_segs, m->b2, p->T, m->ldt, p->Tm,
                                             p->ld_tm, rows, p->part, rmax, cap, bound, ndev, stream);
                if (r == -2)
                    r = m->split_segs
                            ? jlm_vocab_lse_split(m->split_segs, m->split_t_scale, m->split_descale, m->split_bias_col,
                                                  m->n_segs, m->b2, p->T, m->ldt, rows, p->part, rmax, cap, bound,
                                                  ndev, stream)
                            : jlm_vocab_lse_stationary(m->segs, m->n_segs, m->b2, p->T, m->ldt, rows, p->part, rmax,
                                                       p->max_parts, bound, ndev, stream);
                if (r < 0) return r;
                pending_parts = r;
            }
        }
        JLM_TRY(stamp(f, 5));
    }
    if (join) JLM_HIP(hipStreamWaitEvent(main_s, join, 0));
    return jlm_backtrace(lat, &st, p->out_nodes, p->out_len, p->out_score, p->stride, stream);
}

// ABI 8: the full-vocabulary normaliser of a few PROBE rows, in the form asked for -- what DeviceModel's load-time calibration
// of the mixed rows runs (jlm_amd/model.py _calibrate_mixed): `steps` LSTM steps from the zero state over given words, the T
// projection of the last step's rows, then jlm_vocab_lse_split (form 0) or exactly what jlm_decode_frames launches for a model
// with mixed rows (form 1: jlm_pack_t_mixed + jlm_vocab_lse_mixed / _hybrid).  Row g = t * rows + r is hypothesis r after t
// steps; rowlist[g] = g, prev[g] = g - rows (< 0 for block 1: zero state), word[g] = the word consumed by step t.
extern "C" int jlm_lse_probe(const jlm_decode_model *m, const int *rowlist, const int *prev, const int *word, int steps, int rows,
                             void *h, float *c, float *T, void *Tm, int ld_tm, int form, float *part, int max_parts, void *stream) {
    if (steps < 1 || rows < 1 || !m->split_lstm || !m->wt8 || m->self_norm) return -2;
    if (m->untied) {
        // an untied model (round 5): T is the f32 copy of the state the step writes beside its split rows; form 0 = the tile GEMM on
        // split rows (jlm_vocab_lse_partials_split), form 1 = the mixed rows of the vocabulary matrix (k = 512: the wide kernel)
        if (!m->untied_split || m->n_segs != 1) return -2;
        for (int t = 1; t <= steps; ++t)
            JLM_TRY(jlm_lstm_step_xg(h, c, m->H, h, c, rowlist + (size_t)t * rows, prev, word, m->wt8, m->xgate8, m->H, m->gate_descale,
                                     m->h_scale, T, rows, nullptr, stream));
        const int *rl = rowlist + (size_t)steps * rows;
        const jlm_segment &sg = m->segs[0];
        if (form == 0) {
            if ((sg.v_end - sg.v_start + 127) / 128 > max_parts) return -1;
            return jlm_vocab_lse_partials_split(m->untied_split, m->H, sg.v_end - sg.v_start, m->H, h, m->H, rl, m->b2 + sg.v_start,
                                                m->untied_descale, part, rows, 0, rows, nullptr, stream);
        }
        if (!m->mixed_segs || !m->mixed_segs[0].B || !Tm) return -2;
        if (jlm_mixed_t_stride(m->mixed_segs, 1) != ld_tm) return -1;
        JLM_TRY(JLM_PACK_T_MIXED(m)(m->mixed_segs, m->mixed_t_scale, 1, T, m->ldt, rl, rows, nullptr, Tm, ld_tm, stream));
        return (m->lse_fixed_ref ? jlm_vocab_lse_mixed_fr : jlm_vocab_lse_mixed)(m->mixed_segs, m->mixed_descale, m->mixed_s8, m->mixed_bias2, 1, Tm,
                                                                                 ld_tm, part, rows, max_parts, rows, nullptr, stream);
    }
    if (!m->split_segs || !m->pmt_split) return -2;
    for (int t = 1; t <= steps; ++t)
        JLM_TRY(jlm_lstm_step_xg(h, c, m->H, h, c, rowlist + (size_t)t * rows, prev, word, m->wt8, m->xgate8, m->H, m->gate_descale,
                                 m->h_scale, nullptr, rows, nullptr, stream));
    const int *rl = rowlist + (size_t)steps * rows;
    JLM_TRY(jlm_gemm_nt_split(h, m->H, rl, m->pmt_split, m->H, nullptr, T, m->ldt, rl, nullptr, m->t_descale, rows, m->n_t, m->H,
                              nullptr, stream));
    if (form == 0)
        return jlm_vocab_lse_split(m->split_segs, m->split_t_scale, m->split_descale, m->split_bias_col, m->n_segs, m->b2, T, m->ldt, rl,
                                   part, rows, max_parts, rows, nullptr, stream);
    if (!m->mixed_segs || !Tm) return -2;
    jlm_segment only[JLM_MAX_SEGMENTS];
    float only_ts[JLM_MAX_SEGMENTS];
    int n_only = 0;
    for (int i = 0; i < m->n_segs; ++i)
        if (m->mixed_segs[i].B) { only[n_only] = m->mixed_segs[i]; only_ts[n_only++] = m->mixed_t_scale[i]; }
    if (!n_only || jlm_mixed_t_stride(only, n_only) != ld_tm) return -1;
    JLM_TRY(JLM_PACK_T_MIXED(m)(only, only_ts, n_only, T, m->ldt, rl, rows, nullptr, Tm, ld_tm, stream));
    bool cut = false;
    if (m->mixed_head_split)
        for (int i = 0; i < m->n_segs; ++i)
            if (m->mixed_segs[i].B && m->mixed_head_split[i] > 0) cut = true;
    if (n_only == m->n_segs && !cut)       // (the form the decode launches: without a running maximum when the model says so)
        return (m->lse_fixed_ref ? jlm_vocab_lse_mixed_fr : jlm_vocab_lse_mixed)(m->mixed_segs, m->mixed_descale, m->mixed_s8, m->mixed_bias2,
                                                                                 m->n_segs, Tm, ld_tm, part, rows, max_parts, rows, nullptr, stream);
    return jlm_vocab_lse_hybrid(m->split_segs, m->split_t_scale, m->split_descale, m->split_bias_col, m->mixed_segs, m->mixed_descale,
                                m->mixed_s8, m->mixed_head_split, m->n_segs, m->b2, T, m->ldt, Tm, ld_tm, rl, part, rows, max_parts, rows,
                                nullptr, stream);
}
