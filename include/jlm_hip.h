/*
 * jlm_hip.h -- C ABI of libjlm_hip.so, the MI355X (gfx950) implementation of
 * JLM's LSTM inference + lattice beam-search hot path.
 *
 * The reference has no FFI of its own: the path is pure Python over numpy
 * (decoder/model.py, decoder/decoder.py, decoder/decoder_dynamic.py).  Each
 * entry point below therefore replaces a group of numpy / Python statements of
 * the reference, cited as file:line.  The Python classes in jlm_amd/ keep the
 * reference's signatures and call these through ctypes (INTEGRATION.md shows
 * the binding a maintainer of the reference would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless its name ends in _host;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work;
 *   - return value: 0 on success, otherwise the hipError_t of the failed call
 *     (or -1 for an argument the kernels cannot handle, e.g. K % 4 != 0);
 *   - matrices are float32, row-major, leading dimension in floats, a multiple
 *     of 4, base pointers 16-byte aligned; scores are float64;
 *   - "rows" are beam hypotheses.  A hypothesis lives in global row
 *         g = frame * rmax + sentence * beam + slot,  rmax = nsent * beam,
 *     and all per-hypothesis arrays (score, lse, bp, node, word, h, c, T) are
 *     indexed by g.  `live` lists (compact r -> g) name the rows a frame steps.
 *   - counts that exist only on the device (number of live rows of a frame)
 *     are passed as `const int *n_dev`; kernels launched for the static upper
 *     bound exit early past *n_dev.
 */
#ifndef JLM_HIP_H
#define JLM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Library / device probe.  Returns the ABI version (JLM_ABI_VERSION). */
#define JLM_ABI_VERSION 11
#define JLM_MAX_BEAM 1024           /* ABI 6: jlm_beam_step takes beams above one wave (64): a lane owns several ranks */
int jlm_abi_version(void);
/* Writes gfx arch name (e.g. "gfx950:sramecc+:xnack-") of device `dev`. */
int jlm_device_arch(int dev, char *buf, int buflen);

/* ------------------------------------------------------------------------
 * K1+K2+K3+K9: embedding gather, fused gate GEMM, sigmoid/tanh, state update.
 * Replaces LSTM_Model._lstm_cell (decoder/model.py:125-139) and the state
 * gather/scatter of Decoder._batch_predict (decoder/decoder.py:206-218).
 *
 * For compact row r < min(n_rows_max, *n_dev):   g = rows ? rows[r] : r
 *   p = prev[g] (row whose state is consumed; <0 means zero state)
 *   x = [ h[p, 0:H] | emb[word[g], 0:E] ]
 *   z = x . Wt^T + bias     Wt is the packed [4H, kpad] gate matrix
 *   c[g] = c[p]*sig(z_f) + tanh(z_g)*sig(z_i) ;  h[g] = tanh(c[g])*sig(z_o)
 * Packed gate layout (jlm_amd/model.py packs it): row n of Wt / bias is
 *   n = (u / 16) * 64 + gate * 16 + (u % 16), gate order i,f,o,g
 * with Wt[n, 0:H] = HM_gate[:, u], Wt[n, H:H+E] = IM_gate[:, u], zero padded to
 * kpad (multiple of 32).  Requires H % 32 == 0, E % 4 == 0.
 */
int jlm_lstm_step(const float *h_in, const float *c_in, int ld_state,
                  float *h_out, float *c_out,
                  const int *rows, const int *prev, const int *word,
                  const float *emb, int ld_emb,
                  const float *wt, const float *bias, int kpad,
                  int H, int E, int n_rows_max, const int *n_dev, void *stream);

/* ------------------------------------------------------------------------
 * K4 / generic NT GEMM with optional row gathers and column bias:
 *   C[c_rows[m], n] = sum_k A[a_rows[m], k] * B[b_rows[n], k] + bias[n]
 * Replaces np.dot(hidden, PM) (model.py:145,162,184,186), the V_table
 * projections np.dot(temp, VT.T) (model.py:175,177) and, with b_rows = vocab,
 * the materialised logits of LSTM_Model.project (model.py:141-193).
 * NULL row maps mean identity; bias may be NULL.  K % 4 == 0.
 */
int jlm_gemm_nt(const float *A, int lda, const int *a_rows,
                const float *B, int ldb, const int *b_rows,
                float *C, int ldc, const int *c_rows, const float *bias,
                int M, int N, int K, const int *m_dev, void *stream);

/* ------------------------------------------------------------------------
 * K5+K6 fused: one vocabulary segment's logits are produced tile by tile on
 * the MFMA pipe and reduced on the fly to per-row (max, sum exp) partials; the
 * [rows, V] logits never reach HBM.  Replaces the full-vocabulary branch of
 * LSTM_Model.project + softmax (model.py:141-193,15-20).
 *   logit[v, r] = sum_k Bseg[v, k] * T[rows[r], k] + bias[v]   v < n_vocab
 *   part[(tile0 + v/128) * ld_part + r] = (max_v, sum_v exp(logit - max)) over the tile
 * Returns the number of vocab tiles used (>=0) or a negative error.
 */
int jlm_vocab_lse_partials(const float *Bseg, int ldb, int n_vocab, int K,
                           const float *T, int ldt, const int *rows,
                           const float *bias, float *part, int ld_part, int tile0,
                           int n_rows_max, const int *n_dev, void *stream);
/* lse[g] = log sum exp over n_tiles partials, float64;  g = rows[r]. */
int jlm_lse_combine(const float *part, int ld_part, int n_tiles,
                    const int *rows, double *lse,
                    int n_rows_max, const int *n_dev, void *stream);

/* ------------------------------------------------------------------------
 * Vocabulary segments for word-addressed logits (tied softmax: one segment;
 * D-softmax / D-softmax*: model.py:144-181).  Host struct, copied per call.
 */
typedef struct {
    int v_start, v_end;      /* word ids [v_start, v_end) */
    int k;                   /* contraction length (multiple of 4 after padding) */
    int t_off;               /* column offset of this segment's input inside T */
    const float *B;          /* [v_end - v_start, ldb] block, device */
    int ldb;
} jlm_segment;
#define JLM_MAX_SEGMENTS 8

/* Same reduction, rows-stationary form (the one the decoders use): all segments
 * of the model in ONE launch; a workgroup keeps 128 hypothesis rows' MFMA
 * fragments in registers and streams a range of vocabulary tiles past them, so
 * each row gets one (max, sum exp) partial per vocabulary RANGE:
 *   part[p * ld_part + r],  p < return value  (<= max_parts, <= 96)
 * Needs every segment's k <= 256; returns -2 otherwise (use the tile form).
 * Returns the number of partial slices (fold them with jlm_lse_combine) or <0. */
int jlm_vocab_lse_stationary(const jlm_segment *segs_host, int n_segs, const float *b2,
                             const float *T, int ldt, const int *rows,
                             float *part, int ld_part, int max_parts,
                             int n_rows_max, const int *n_dev, void *stream);

/* ------------------------------------------------------------------------
 * Split-f16 ("f16x3") operands.  The f32 matrix pipe of gfx950 runs at 1/16 of
 * the f16 rate; these entry points carry every f32 value x as two f16 halves
 *   x * scale = hi + lo (+ r, |r| <= 2^-22 |x * scale|)
 * and form a product as hi.hi + hi.lo + lo.hi in three f16 MFMAs with f32
 * accumulation -- f32-grade results (error pinned in tests/test_gpu_kernels.py)
 * at 16/3 of the f32 MFMA rate.  "Split rows": a row of k values, k padded with
 * zeros to a multiple of 16, stored as k/8 blocks of [8 x f16 hi][8 x f16 lo]
 * (4 bytes per value, row stride ld_dst in 4-byte units, a multiple of 16).
 * scale must be a power of two (so that dividing it out again is exact) chosen
 * so that max|x| * scale stays below 65504.  jlm_pack_split_f16 writes the
 * blocks that cover k rounded up to 16 values (zero padded) and leaves the rest
 * of each destination row alone, so a matrix can be packed in column ranges
 * with different scales (dst / src advanced by a multiple of 16 values). */
int jlm_pack_split_f16(const float *src, int rows, int k, int ld, float scale,
                       void *dst, int ld_dst, void *stream);

/* ABI 4: k-means compressed weights (train/comp.py:52-80; selected by decoder/model.py:74-78 through `comp`): per tensor a
 * uint8 code array and a float32 codebook of <= 256 entries.  dst[r][c] = codebook[code[r][c]] for c < k, on the device
 * (what np.take(codebook, code) does on the host in train/comp.py:70): the codes stay resident, the float panel is
 * expanded from them where the kernels need it. */
int jlm_dequant_u8(const uint8_t *code, int rows, int k, int ld_code, const float *codebook, int n_codes,
                   float *dst, int ld_dst, void *stream);

/* ABI 4: the decode's LSTM step (K1+K2+K3+K9; decoder/model.py:125-139 with the state gather / scatter of
 * Decoder._batch_predict, decoder/decoder.py:206-218), table form, one 160-row x 128-gate-column tile per CU.
 * Same row semantics as jlm_lstm_step.  Operands:
 *   h_in / h_out   split rows of the state scaled by h_scale (|h| < 1, h_scale a power of two <= 2^14);
 *   wt8            split rows [4H, H] of the state half of the gate matrix scaled by 1 / (descale * h_scale),
 *                  in the gate-interleave-8 row order  n = (u / 8) * 32 + gate * 8 + (u % 8), gate order i,f,o,g
 *                  (a 32-row MFMA block = four gates of eight units: the cell update needs no transposition);
 *   xgate8         f32 [V, 4H], same column order: (emb[w] . W_x^T + bias) / descale for every vocabulary word
 *                  (model.py:125-131 computes x.IM_g + h.HM_g + b_g; the table is the x.IM_g + b_g part);
 *   h_f32_out      optional (may be NULL): h' also as plain f32 rows, same stride (untied models: T is the state itself and
 *                  the edge-logit / word-list kernels read T as f32);
 *   c stays f32.   H % 32 == 0, ld_state % 16 == 0.  State rows are addressed as 16-byte records through a 31-bit index:
 *   (highest row number + 1) x ld_state / 4 < 2^31 (16.7 M rows at H = 512). */
int jlm_lstm_step_xg(const void *h_in, const float *c_in, int ld_state, void *h_out, float *c_out,
                     const int *rows, const int *prev, const int *word,
                     const void *wt8, const float *xgate8, int H, float descale, float h_scale, float *h_f32_out,
                     int n_rows_max, const int *n_dev, void *stream);

/* jlm_vocab_lse_partials on split rows (the tile form for k > 256: untied models, k = H; model.py:189-191):
 *   logit[v, r] = descale * sum_k Bsplit[v, k] * Tsplit[rows[r], k] + bias[v]
 * Bsplit = split rows of the segment's matrix scaled by 2^eB, Tsplit = split rows of the row operand scaled by 2^eT (for an
 * untied model the state rows the LSTM step wrote, eT = 14), descale = 2^-(eT + eB); strides in 4-byte units, K % 16 == 0.
 * Same partial-slice contract and return value as jlm_vocab_lse_partials. */
int jlm_vocab_lse_partials_split(const void *Bsplit, int ldb, int n_vocab, int K, const void *Tsplit, int ldt, const int *rows,
                                 const float *bias, float descale, float *part, int ld_part, int tile0,
                                 int n_rows_max, const int *n_dev, void *stream);

/* jlm_gemm_nt on split rows: C = descale * (A . B^T) + bias, C plain f32. */
int jlm_gemm_nt_split(const void *A, int lda, const int *a_rows, const void *B, int ldb, const int *b_rows,
                      float *C, int ldc, const int *c_rows, const float *bias, float descale,
                      int M, int N, int K, const int *m_dev, void *stream);

/* One column of split rows from a vector: dst[r][col] = split(v[r] * scale). */
int jlm_pack_split_f16_col(const float *v, int rows, float scale, void *dst, int ld_dst, int col, void *stream);

/* jlm_vocab_lse_stationary on split rows: segs[i].B = split rows of the
 * segment's output embedding scaled by 2^eB_i, segs[i].ldb their stride in
 * 4-byte units, segs[i].k the true contraction length (<= 256, multiple of 4).
 * T is plain f32 (the kernel splits its rows while loading them, after scaling
 * by t_scale[i] = 2^eT_i); descale[i] = 2^-(eT_i + eB_i).
 * bias_col (may be NULL): bias_col[i] = segs[i].k says that column k of the
 * segment's split rows (the first padded one; needs k % 16 != 0) holds
 * b2[word] * 2^eB_i -- the kernel then feeds 1.0 at that position of every T
 * row and the bias costs nothing in the fold; bias_col[i] = -1: b2 is added in
 * the fold.  Same partial-slice contract and return value as
 * jlm_vocab_lse_stationary.  The vocabulary is cut into COLUMNS of equal cost
 * (one workgroup per column and 256-row tile), at most 256 / row tiles of them;
 * a column that crosses a segment boundary writes one slice per segment it
 * touches, so a launch writes at most columns + n_segs - 1 slices, and
 * max_parts (the capacity of `part` in slices) bounds the column count to
 * max_parts - (n_segs - 1). */
int jlm_vocab_lse_split(const jlm_segment *segs_host, const float *t_scale, const float *descale,
                        const int *bias_col, int n_segs, const float *b2,
                        const float *T, int ldt, const int *rows,
                        float *part, int ld_part, int max_parts,
                        int n_rows_max, const int *n_dev, void *stream);

/* Word-list groups: one per (sentence, frame).  Group j covers hypothesis rows
 * g0[j] .. g0[j]+cnt[cnt_idx[j]]-1 and the word list number l = wl_base +
 * wl_idx[j], i.e. words wl[wl_off[l] .. wl_off[l+1]). */

/* K7 operand: logits of the lattice edges leaving a frame.  For every group j,
 * word position i in its list and beam slot k:
 *   edge[wl_out[i] * beam + k] = T[g0+k] . B[w_i] + b2[w_i]
 * (wl_out = lattice node id of the edge).  Replaces the indexing
 * pred[node.word_idx] of Path.append_node (decoder.py:43-49,172-182) -- only
 * the logits the lattice can consume are ever formed. */
int jlm_edge_logits(const jlm_segment *segs_host, int n_segs, const float *b2,
                    const float *T, int ldt,
                    const int *g0, const int *cnt, const int *cnt_idx,
                    const int *wl, const int *wl_off, const int *wl_idx, int wl_base,
                    const int *wl_out, float *edge, int beam, int n_groups, void *stream);

/* K5b/K6/K11: log-sum-exp over a selected vocabulary (vocab_select), or its
 * online extension by newly needed words (incremental vocabulary selection).
 * Replaces project(hidden, vocab)+softmax (model.py:184,15-20; decoder.py:202-218)
 * and the back-fill + re-softmax of DynamicDecoder._incremental_decode
 * (decoder_dynamic.py:133-148).  merge=0: (m,s) := over the list; merge=1:
 * (m,s) := (m,s) (+) list.  lse[g] = m + log(s) is refreshed either way.
 * Duplicate words in a list count twice, as in the reference. */
int jlm_wordlist_lse(const jlm_segment *segs_host, int n_segs, const float *b2,
                     const float *T, int ldt,
                     const int *g0, const int *cnt, const int *cnt_idx,
                     const int *wl, const int *wl_off, const int *wl_idx, int wl_base,
                     float *run_max, double *run_sum, double *lse,
                     int merge, int beam, int n_groups, void *stream);

/* jlm_edge_logits / jlm_wordlist_lse with the weight row and the bias of a list position taken from DIFFERENT words:
 * position i uses the weight row of word wl_w[i] and the bias of word wl[i] (wl_w == NULL: the plain forms).  This is
 * what the reference computes in DynamicDecoder on D-softmax / D-softmax* models, where project() returns the
 * columns of a vocabulary subset segment-major and the caller reads them in list order (model.py:152-158,168-179
 * under decoder_dynamic.py:130) -- reproduced behind DynamicDecoder.compat_quirks. */
int jlm_edge_logits_perm(const jlm_segment *segs_host, int n_segs, const float *b2,
                         const float *T, int ldt,
                         const int *g0, const int *cnt, const int *cnt_idx,
                         const int *wl, const int *wl_w, const int *wl_off, const int *wl_idx, int wl_base,
                         const int *wl_out, float *edge, int beam, int n_groups, void *stream);
int jlm_wordlist_lse_perm(const jlm_segment *segs_host, int n_segs, const float *b2,
                          const float *T, int ldt,
                          const int *g0, const int *cnt, const int *cnt_idx,
                          const int *wl, const int *wl_w, const int *wl_off, const int *wl_idx, int wl_base,
                          float *run_max, double *run_sum, double *lse,
                          int merge, int beam, int n_groups, void *stream);

/* jlm_wordlist_lse on split rows, single-segment models (seg->B = split rows scaled by 2^eB,
 * t_scale = 2^eT, descale = 2^-(eT+eB) as for jlm_vocab_lse_split; b2 is added in the fold).
 * max_words = longest word list among the groups (<= 4064).  Returns -2 when the shape is outside
 * the kernel (k > 256, beam > 32, longer lists): use jlm_wordlist_lse then. */
int jlm_wordlist_lse_split(const jlm_segment *seg_host, float t_scale, float descale, const float *b2,
                           const float *T, int ldt,
                           const int *g0, const int *cnt, const int *cnt_idx,
                           const int *wl, const int *wl_off, const int *wl_idx, int wl_base, int max_words,
                           float *run_max, double *run_sum, double *lse,
                           int merge, int beam, int n_groups, void *stream);

/* K11 for a whole frame in one launch (DynamicDecoder._incremental_decode, decoder_dynamic.py:133-148):
 * every row g = fr * rmax + s * beam + slot, fr < n_old_frames, slot < cnt[fr * n_sent + s], of every
 * sentence s merges the words wl[wl_off[wl_base + s] .. wl_off[wl_base + s + 1]) into its running
 * (run_max, run_sum) and refreshes lse -- what jlm_wordlist_lse(merge = 1) does for n_old_frames * n_sent
 * groups, but with one workgroup per sentence gathering the list once.  rmax = n_sent * beam.
 * Returns -2 outside the kernel's shape (max_words > 128, k > 256, beam > 32). */
int jlm_wordlist_merge_split(const jlm_segment *seg_host, float t_scale, float descale, const float *b2,
                             const float *T, int ldt, const int *cnt, int n_sent, int beam, int n_old_frames,
                             const int *wl, const int *wl_off, int wl_base, int max_words,
                             float *run_max, double *run_sum, double *lse, void *stream);

/* ------------------------------------------------------------------------
 * Lattice of a batch (CSR, built on the host by jlm_amd/lattice.py following
 * Decoder._build_lattice, decoder.py:79-135), resident in HBM for the decode.
 */
typedef struct {
    int n_sent, beam, n_frames;   /* n_frames = max sentence length + 1 */
    const int *sent_len;          /* [n_sent] kana length */
    const int *end_off;           /* [n_frames*n_sent + 1] nodes ending at (frame, sentence) */
    const int *node_start;        /* [n_nodes] start frame (-1 for <eos>) */
    const int *node_word;         /* [n_nodes] softmax row of the word */
} jlm_lattice;

typedef struct {
    double *score;                /* [G] accumulated -log p (decoder.py:36,49) */
    double *lse;                  /* [G] log-normaliser of the row's next-word distribution */
    double *ysum;                 /* [G] dynamic decoder: sum of edge logits along the path */
    int *bp;                      /* [G] previous hypothesis row (-1 at the root) */
    int *node;                    /* [G] lattice node consumed last */
    int *word;                    /* [G] its softmax row (LSTM input) */
    int *cnt;                     /* [n_frames*n_sent] hypotheses alive per (frame, sentence) */
    int *live;                    /* [n_frames*rmax] compact list of rows to step per frame */
    int *n_live;                  /* [n_frames] */
    const float *edge;            /* [n_nodes*beam] edge logits */
    /* -- ABI 2: fused K6 tail.  live_base[(frame, sentence)] = position of the sentence's first row
     * in live[frame] (written by jlm_beam_step when it lists the rows; may be NULL).  lse_part != NULL
     * (mode 0 only): the n_parts partial slices [n_parts][ld_part] of (max, sum exp) pairs that
     * jlm_vocab_lse_* left for the rows of frame - 1, indexed by live position, are folded into
     * lse[] by jlm_beam_step(frame) itself -- no jlm_lse_combine launch in between. */
    int *live_base;               /* [n_frames*n_sent] */
    const float *lse_part;
    int ld_part, n_parts;
    /* ABI 11: one device int (NULL: none) that jlm_beam_step ORs 1 into when a folded log-normaliser is not finite -- a row whose
     * logits left the range of the fixed-reference normaliser (jlm_vocab_lse_mixed_fr: sum 2^y overflowed to inf or vanished to 0).
     * An overflowed row's hypotheses score -inf and are pruned silently otherwise; the caller zeroes the int per batch and reads it
     * back with the traces (jlm_amd/engine.py: DecodeEngine.collect raises). */
    int *flags;
} jlm_beam_state;

/* K7+K8: candidate scoring and stable per-sentence top-k for frame `frame`.
 * Replaces Decoder._build_current_frame + sort/truncate (decoder.py:164-182,
 * 227-229).  mode 0: static decoder, score = score[p] + lse[p] - edge;
 * mode 1: self-normalised model, score = score[p] - edge (model.py:117-118);
 * mode 2: DynamicDecoder (decoder_dynamic.py:53-91,150-175): every path is
 * re-scored from the head with the current normalisers.
 * Ties keep candidate generation order (node order, then beam slot).  Rows of
 * a sentence's last frame are not listed in `live`: the reference steps them
 * too but never reads the result (decoder.py:233-237).
 * max_cands >= beam * (largest number of nodes ending at one (frame, sentence)).
 * 1 <= beam <= JLM_MAX_BEAM (the reference has no limit, decoder.py:227-229); a cell's candidates live in one wave's
 * LDS -- in one piece up to ~13 k, above that chunk by chunk with the chunks' winners merged (round 6: same order, same result);
 * -1 when max_cands exceeds jlm_beam_step_max_cands(beam, n_frames, mode). */
int jlm_beam_step(const jlm_lattice *lat_host, const jlm_beam_state *st_host,
                  int frame, int mode, int max_cands, void *stream);

/* ABI 7: the vocabulary projection + log-sum-exp with the two cross terms of the split product on the INT8 matrix pipe
 * (csrc/jlm_mixed.hip; reference project + softmax, decoder/model.py:141-193, 15-20).
 *   t.b ~ t_hi.b_hi (f16 x f16, v_mfma_f32_32x32x16_f16) + [t_hi.b_lo + t_lo.b_hi] (int8 x int8 into one i32 accumulator,
 *         v_mfma_i32_32x32x32_i8; hi8 = rint(hi / s), lo8 = rint(lo / (s 2^-11)), s a power of two per T row / per segment)
 * "Mixed rows": per 32 k-values a 128-byte block [32 x f16 hi | 32 x int8 hi8 | 32 x int8 lo8]; the bias of a word rides in
 * the f16 part as columns k (hi of b2 2^eB log2 e) and k + 1 (its f16 residual x 2^11), so a row has nb = ceil((k + 2) / 32)
 * blocks (ld_dst = 32 nb in 4-byte units, nb <= 8) -- or, for k a multiple of 32, nb = k / 32 and no bias columns (below).
 * jlm_pack_mixed: src [rows, k] f32 (stride ld) and bias [rows] -> dst; scale = 2^eB, bias_scale = 2^eB log2 e, s8 = the
 * segment's int8 scale (a power of two >= max |f16(src scale)| / 127). */
int jlm_pack_mixed(const float *src, int rows, int k, int ld, const float *bias, float scale, float bias_scale, float s8,
                   void *dst, int ld_dst, void *stream);
/* The hypothesis side: T [G, ldt] f32 -> packed rows Tm, COMPACT: packed row r = hypothesis row rows[r] (a frame's live rows: one
 * small buffer, rewritten every frame) (ld_tm = jlm_mixed_t_stride(segs, n_segs), 4-byte units per row): per segment nb blocks of
 * x = T 2^eT log2 e in the same block format, the bias constants 2^eT / 2^(eT-11) at columns k, k + 1 of the f16 part, and
 * JLM_MAX_SEGMENTS floats per row: the row's int8 scale per segment.  Once per row and frame (one wave per row); the vocabulary
 * kernel's workgroups only load the result.  t_scale[i] = 2^eT_i (a power of two).
 * ABI 9: the buffer is an opaque image of WHOLE 32-row blocks -- the caller allocates ceil(n_rows_max / 32) * 32 rows of ld_tm
 * floats -- laid out granule-major inside a block (16-byte granule g of row r at block (r / 32) + g * 512 + (r % 32) * 16, the
 * scales behind the granules), so that a wave of the vocabulary kernel reads one contiguous kilobyte per operand load. */
int jlm_mixed_t_stride(const jlm_segment *segs_host, int n_segs);
int jlm_pack_t_mixed(const jlm_segment *segs_host, const float *t_scale, int n_segs, const float *T, int ldt, const int *rows,
                     int n_rows_max, const int *n_dev, void *Tm, int ld_tm, void *stream);
/* ABI 11 (round 6): mx6 rows -- the two cross terms of the split product as FP6 (e2m3) x FP6 on the block-scaled matrix instruction
 * (v_mfma_scale_f32_32x32x64_f8f6f4: one instruction per 32 k-values for both terms, accumulated into the f16 pass's f32 accumulator;
 * csrc/jlm_mx6_body.h, csrc/jlm_mx6.hip; reference project + softmax, decoder/model.py:141-193, 15-20).  Same 128-byte blocks, strides
 * and buffers as the int8 form; granules 4-6 of a block hold the FP6 planes (hi6, lo6) and granule 7 of a row's first block their E8M0
 * scales, one per plane and 32 k-values.  Selected by s8 = 0:
 *   jlm_pack_mixed(..., s8 = 0, ...)           packs a vocabulary block as mx6 rows (at most 8 blocks per row: k + 2 <= 256 or k = 256);
 *   jlm_pack_t_mixed6                            packs hypothesis rows in that form (same arguments and stride as jlm_pack_t_mixed);
 *   jlm_vocab_lse_mixed(_fr)(..., s8[i] = 0 for EVERY segment, ...)  runs the launch on mx6 rows (-2: formats mixed within a launch,
 *                                                or k = 512); jlm_vocab_lse_hybrid takes int8 rows only (-2);
 *   jlm_decode_model.mixed_s8[i] = 0 for every mixed segment makes jlm_decode_frames / jlm_lse_probe use the two above. */
int jlm_pack_t_mixed6(const jlm_segment *segs_host, const float *t_scale, int n_segs, const float *T, int ldt, const int *rows,
                      int n_rows_max, const int *n_dev, void *Tm, int ld_tm, void *stream);
/* segs[i].B = mixed rows, segs[i].ldb = 32 nb, segs[i].k the true contraction length; descale[i] = 2^-(eT_i + eB_i), s8[i] as
 * above; Tm = the packed hypothesis rows.  Same partial-slice contract and return value as jlm_vocab_lse_split; -2: a shape
 * this form does not take (more than 8 blocks; segments of both bias forms in one launch).
 * A contraction that fills its last block (k a multiple of 32: the tied k = 256 models) has no columns left for the bias: its
 * rows are packed with ld_dst = k (jlm_pack_mixed then ignores `bias`), segs[i].ldb = k says so, and the kernel takes the
 * biases from bias2 [V] = b2 log2(e) (device; may be NULL otherwise) -- staged into LDS beside each tile, added in the combine. */
int jlm_vocab_lse_mixed(const jlm_segment *segs_host, const float *descale, const float *s8, const float *bias2, int n_segs,
                        const void *Tm, int ld_tm, float *part, int ld_part, int max_parts, int n_rows_max,
                        const int *n_dev, void *stream);
/* ABI 9: the same launch WITHOUT a running maximum where a kernel form for it exists (the wide kernel's tied k = 256 and k = 512 forms;
 * every other shape runs exactly as jlm_vocab_lse_mixed): s = sum over the words of 2^(base-2 logit) against the fixed reference 0, slices
 * (0, s) -- three VALU instructions per logit less.  Valid while every row's largest logit stays within about +-69 (base-2: +-100: f32
 * range over 2^16 words); a row outside it comes back as s = 0 or inf.  The caller decides per model (jlm_decode_model.lse_fixed_ref). */
int jlm_vocab_lse_mixed_fr(const jlm_segment *segs_host, const float *descale, const float *s8, const float *bias2, int n_segs,
                        const void *Tm, int ld_tm, float *part, int ld_part, int max_parts, int n_rows_max,
                        const int *n_dev, void *stream);

/* One launch over segments of BOTH formats (csrc/jlm_split.hip, vocab_lse_hybrid_kernel): mixed[i].B != NULL runs segment i on
 * its mixed rows (mixed[i].ldb = 32 nb; mx_descale[i], mx_s8[i] as for jlm_vocab_lse_mixed; Tm = rows packed by
 * jlm_pack_t_mixed over the MIXED segments only, in segment order, for the same `rows`), the others on their split rows exactly as
 * jlm_vocab_lse_split (segs / t_scale / descale / bias_col cover every segment).  The int8 cross terms pay where the matrix
 * instructions dominate a block (k = 200, 100); where the fold does (k = 50) the three f16 passes stay.  -2: a shape the kernel
 * does not host (mixed: k + 2 in (192, 208], (96, 112] or (32, 64]; split: k <= 208, not a multiple of 16, with its bias column): use
 * jlm_vocab_lse_split.
 * ABI 10, head_split (host array of n_segs ints, or NULL): the first head_split[i] words of MIXED segment i (a multiple of 128, less
 * than the segment) run on its split rows, the rest on its mixed rows.  In a trained model the frequent words -- the low ids of the
 * first segment -- carry the probability mass and with it the int8 cross terms' contribution to the log-normaliser's error (measured
 * on logits of +-20: head segment 3.1e-6 rms, the other two 2e-8); three f16 passes for those few thousand words buy the split form's
 * accuracy at the mixed form's cost for the other 90 % of the vocabulary.  The loader picks the cut (DeviceModel._calibrate_mixed). */
int jlm_vocab_lse_hybrid(const jlm_segment *segs_host, const float *t_scale, const float *descale, const int *bias_col,
                         const jlm_segment *mixed, const float *mx_descale, const float *mx_s8, const int *head_split, int n_segs,
                         const float *b2, const float *T, int ldt, const void *Tm, int ld_tm, const int *rows, float *part, int ld_part,
                         int max_parts, int n_rows_max, const int *n_dev, void *stream);

/* ABI 6: the largest max_cands (a multiple of 256, as the plans round it) jlm_beam_step accepts for this beam,
 * frame count and mode -- the launcher's own LDS formula, so that callers can route sentences with a larger lattice
 * cell to a host-side search (Decoder._decode_unpruned / DynamicDecoder._decode_host) instead of failing the batch.
 * Round 6: cells that do not fit one wave's LDS in one piece are selected chunk by chunk, so the figure is what the chunk
 * winners leave room for (3.4 M candidates at beam 10, 0.5 M at beam 64): no real lexicon gets near it.
 * 0: no cell fits (beam or frame count too large).  Pure host function, no GPU needed. */
int jlm_beam_step_max_cands(int beam, int n_frames, int mode);

/* K10: n-best read-out.  For sentence s and rank r < cnt at its last frame:
 * out_nodes[(s*beam+r)*stride + d] = node ids from the LAST word back to the
 * root, out_len = number of nodes, out_score = path score (decoder.py:237). */
int jlm_backtrace(const jlm_lattice *lat_host, const jlm_beam_state *st_host,
                  int *out_nodes, int *out_len, double *out_score, int stride, void *stream);

/* K6: row softmax / exp for the LSTM_Model.predict API (model.py:15-20,117-120).
 * pred[r, :] = self_norm ? exp(y[r, :]) : softmax(y[r, :]). */
int jlm_softmax_rows(const float *y, float *pred, int ld, int n_rows, int n_cols,
                     int self_norm, void *stream);

/* ------------------------------------------------------------------------
 * ABI 3: the frame loop itself.  Decoder.decode (decoder/decoder.py:220-241:
 * for every frame build the candidates, keep the best `beam`, step the LSTM of
 * the survivors) and DynamicDecoder.decode / _incremental_decode
 * (decoder/decoder_dynamic.py:177-194, 93-175) as ONE call that enqueues the
 * whole launch sequence of a batch -- per frame
 *   [incremental: merge the frame's new words into all older rows]  jlm_wordlist_merge_split | jlm_wordlist_lse(merge)
 *   jlm_beam_step                (folds the previous frame's normaliser slices)
 *   jlm_lstm_step_xg | jlm_lstm_step(_split), jlm_gemm_nt(_split) (T projection)
 *   jlm_edge_logits              (on `side_stream` when given, forked after T and joined before the next beam step)
 *   [jlm_pack_t_mixed +] jlm_vocab_lse_hybrid | jlm_vocab_lse_split | _stationary | jlm_wordlist_lse(_split)
 * and jlm_backtrace at the end -- exactly the calls a host would make one by
 * one through the entry points above (jlm_amd/engine.py does, for timing and
 * for models outside this call's shapes), without ~170 trips through the host
 * language's FFI per batch.  Nothing here synchronises with the device.
 */
typedef struct {
    const jlm_segment *segs;        /* f32 segments (edge logits, short word lists) */
    int n_segs;
    const float *b2;
    int H, ldt;
    int untied;                     /* T aliases h: no T projection (model.py:189-191) */
    int self_norm;                  /* no normaliser at all (model.py:117-118) */
    int split_lstm;                 /* state rows and gate matrix are split rows */
    /* jlm_lstm_step operands (split_lstm == 0) */
    const float *emb; int ld_emb; const float *wt; const float *gate_bias; int kpad, E;
    /* jlm_lstm_step_xg operands (split_lstm == 1; the input side is the per-word table xgate8).  ABI 9: the round-1 split step
     * (jlm_lstm_step_split: wt_split / kpad_split / xgate) is gone */
    float gate_descale, h_scale;
    const void *wt8; const float *xgate8;
    /* untied model on split rows (ABI 4; untied_split != NULL): the vocabulary matrix UM^T [V, H] as split rows scaled by
     * 2^eB, untied_descale = 2^-(14 + eB); the state rows plan.h are split rows then, plan.T their plain f32 copy */
    const void *untied_split; float untied_descale;
    /* ABI 9: 1 = the full-vocabulary normaliser on mixed rows runs jlm_vocab_lse_mixed_fr (no running maximum): set by the loader when
     * the model's own log-normalisers (load-time probe) sit well inside the f32 range */
    int lse_fixed_ref;
    /* T projection: [n_t, H] panel, plain or split rows */
    const float *pmt; const void *pmt_split; int n_t; float t_descale;
    /* full-vocabulary normaliser: split segments (NULL: f32 rows-stationary form) */
    const jlm_segment *split_segs; const float *split_t_scale; const float *split_descale; const int *split_bias_col;
    /* ABI 7: segments of that normaliser on MIXED rows (jlm_vocab_lse_hybrid; NULL: none): n_segs entries, mixed_segs[i].B ==
     * NULL leaves segment i on its split rows; mixed_t_scale / mixed_descale / mixed_s8 [n_segs] as for jlm_pack_t_mixed /
     * jlm_vocab_lse_mixed.  Used when the plan carries the packed-row buffer (plan.Tm). */
    const jlm_segment *mixed_segs; const float *mixed_t_scale; const float *mixed_descale; const float *mixed_s8;
    const float *mixed_bias2;       /* b2 log2(e) [V] for mixed segments without bias columns (NULL: none) */
    /* ABI 10: [n_segs] or NULL -- the leading words of a mixed segment that stay on split rows (jlm_vocab_lse_hybrid head_split) */
    const int *mixed_head_split;
} jlm_decode_model;

typedef struct {
    int kind;                       /* 0 static, full vocabulary; 1 static, selected vocabulary; 2 incremental */
    int max_cands;
    void *h; float *c; float *T;    /* [G, H] state rows (f32 or split), [G, H] f32, [G, ldt] f32 */
    const int *g0, *cidx, *sidx;    /* [n_frames*n_sent]: first row of a cell, cell index, sentence index */
    const int *sg_word, *sg_off, *sg_node;   /* lattice edges grouped by START (frame, sentence): word, CSR, node id */
    float *edge;                    /* == jlm_beam_state.edge */
    const int *vs_words, *vs_off; int vs_max;   /* kind 1: per-sentence selected vocabulary, longest list */
    /* kind 2: the vocabulary a (frame, sentence) cell c starts with = di_words[di_off[2c] .. di_off[2c+1])
     * (slices of per-sentence sequences, include/jlm_host.h jlm_dynamic_vocab); di_idx[c] = 2 * sentence */
    const int *di_words, *di_off, *di_idx; int di_max;
    const int *dd_words, *dd_off; int dd_max;   /* kind 2: words new at a frame, per (frame, sentence) */
    float *run_max; double *run_sum;            /* kinds 1, 2: running (max, sum exp) per row */
    float *part; int max_parts;                 /* kind 0: [max_parts][rmax][2] partial slices */
    /* kind 0: share (percent, 0 = all) of the CUs the vocabulary kernel may fill.  With two batches in flight on two streams
     * the other batch's latency-bound kernels (beam step, LSTM step, T projection) run on the CUs it leaves free instead of
     * behind it: 2.61 -> 2.33 ms per step at BASELINE configs[1] with 16 instead of 24 vocabulary ranges (tools/ab_np.py) */
    int lse_cu_share_pct;
    int *out_nodes; int *out_len; double *out_score; int stride;    /* jlm_backtrace outputs */
    /* kind 2 with a SEGMENTED projection, reference-compatibility mode (both NULL otherwise).  The reference's project()
     * returns a vocabulary subset's columns segment-major while DynamicDecoder indexes them -- and adds the bias -- in list
     * order (decoder_dynamic.py:76,130,172 over model.py:152-158,168-179), so list position j of a cell's first
     * vocabulary pairs the weight row of word di_wwords[j] with the bias and the identity of word di_words[j]; sg_wword[e]
     * is the word whose weight row the reference reads for lattice edge e (parallel to sg_word). */
    const int *di_wwords, *sg_wword;
    /* ABI 7, kind 0 with model.mixed_segs: [n_sent * beam][ld_tm] packed hypothesis rows of the frame being stepped
     * (jlm_pack_t_mixed over the model's mixed segments, right behind the T projection; ld_tm = jlm_mixed_t_stride of those) */
    void *Tm; int ld_tm;
} jlm_decode_plan;

/* Returns 0 or a hipError_t.  st_host->lse_part / n_parts are managed by the call.  A full-vocabulary
 * model with a segment of k > 256 (untied: k = H, model.py:189-191) takes the tile form of the normaliser
 * (jlm_vocab_lse_partials per segment, one slice per 128 words: plan.max_parts >= the number of such tiles).
 *
 * events (may be NULL): JLM_EVENTS_PER_FRAME * n_frames hipEvent_t created by the caller (timing enabled).
 * They are recorded on `stream` around the kernel groups of every frame -- the edge logits then run on
 * `stream` too, so each bracket holds exactly what it names:
 *   [0] frame start  [1] after the incremental merge ("vocab fix", decoder_dynamic.py:112-148)
 *   [2] after the beam step ([1]..[2] = "lattice path fix" for the incremental decoder, :150-175)
 *   [3] after the LSTM step ([2]..[3] = perf_log_lstm, decoder.py:206-218; the gate GEMM alone)
 *   [4] after the T projection and the edge logits   [5] after the normaliser ([4]..[5] = the vocabulary
 *   kernel alone; [3]..[5] = perf_log_softmax).  Frames past the last stepped one record [0]..[2] only. */
#define JLM_EVENTS_PER_FRAME 6
int jlm_decode_frames(const jlm_decode_model *model_host, const jlm_decode_plan *plan_host,
                      const jlm_lattice *lat_host, const jlm_beam_state *st_host,
                      void *stream, void *side_stream, void *const *events);

/* ABI 8: probe of the full-vocabulary normaliser, for the load-time calibration of the mixed rows (jlm_amd/model.py
 * DeviceModel._calibrate_mixed; the reference computes every logit in float64, decoder/model.py:141-193,15-20 -- which
 * int8 cross terms can follow only as far as the model's logit range lets them).  Runs `steps` LSTM steps (jlm_lstm_step_xg)
 * of `rows` hypotheses from the zero state -- row g = t * rows + r is hypothesis r after t steps: rowlist[g] = g,
 * prev[g] = g - rows (negative in block 1), word[g] = the word step t consumes; h, c [(steps + 1) * rows, H], T
 * [(steps + 1) * rows, ldt] -- then the T projection of the last block and its normaliser slices into part [max_parts][rows][2]:
 * form 0 = jlm_vocab_lse_split, form 1 = what jlm_decode_frames launches with the model's mixed rows (jlm_pack_t_mixed into
 * Tm [rows][ld_tm] + jlm_vocab_lse_mixed / jlm_vocab_lse_hybrid).  Returns the number of slices (>= 1), -2 for a model
 * without that form, -1 / a hipError_t as the launchers do. */
int jlm_lse_probe(const jlm_decode_model *model_host, const int *rowlist, const int *prev, const int *word, int steps, int rows,
                  void *h, float *c, float *T, void *Tm, int ld_tm, int form, float *part, int max_parts, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* JLM_HIP_H */
