/*
 * jlm_host.h -- C ABI of libjlm_host.so: the native (host-side, multi-threaded)
 * lattice builder.  It replaces the Python dictionary scan of
 * Decoder._build_lattice / _build_lattice_vocab (reference decoder/decoder.py:
 * 79-151) for whole batches; no GPU, no torch.  jlm_amd/lattice.py binds it with
 * ctypes and falls back to nothing: the pure-Python builder in the same file is
 * the specification the native one is tested against (tests/test_lattice_native.py).
 *
 * All arrays are int32 unless stated; text and readings are UTF-32 code points.
 */
#ifndef JLM_HOST_H
#define JLM_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JLM_HOST_ABI_VERSION 3
int jlm_host_abi_version(void);

typedef struct jlm_lexicon jlm_lexicon;

/* Reading dictionary (reference data/reading_dict.pkl, data.py:57-76) restricted to
 * in-vocabulary words: reading r = code points reading_cp[reading_off[r] ..
 * reading_off[r+1]), entries entry_off[r] .. entry_off[r+1] with softmax row
 * entry_word[] and lexicon index entry_lex[], already sorted by lexicon index
 * (decoder.py:95).  eos_word / unk_word = w2i['<eos>'] / w2i['<unk>']. */
jlm_lexicon *jlm_lexicon_create(const uint32_t *reading_cp, const int32_t *reading_off,
                                const int32_t *entry_off, const int32_t *entry_word,
                                const int32_t *entry_lex, int32_t n_readings,
                                int32_t eos_word, int32_t unk_word);
void jlm_lexicon_destroy(jlm_lexicon *lx);

/* Lattice of a batch as CSR (layout: jlm_amd/lattice.py).
 * Sentence s = text[text_off[s] .. text_off[s+1]); n_frames = longest sentence + 1;
 * cells are frame-major: cell = frame * n_sent + sentence.
 * Node ids follow (end frame, sentence, generation order) -- the reference's
 * backward_lookup order (decoder.py:79-135), which is the beam's tie-break order.
 * node_lex: lexicon index, -1 = <eos>, -2 = raw-symbol <unk> fallback (decoder.py:128-130).
 * end_off / sg_off have n_frames*n_sent + 1 entries; sg_* list the nodes STARTING in a cell.
 * Returns the number of nodes; if it exceeds node_cap only the offsets and
 * *max_nodes_per_cell are valid and the call must be repeated with larger arrays. */
int64_t jlm_lattice_build(const jlm_lexicon *lx, const uint32_t *text, const int32_t *text_off,
                          int32_t n_sent, int32_t n_frames, int64_t node_cap,
                          int32_t *node_start, int32_t *node_word, int32_t *node_lex,
                          int32_t *node_sent, int32_t *node_end, int32_t *end_off,
                          int32_t *sg_off, int32_t *sg_node, int32_t *sg_word,
                          int32_t *max_nodes_per_cell, int32_t n_threads);

/* Static vocabulary selection (decoder.py:137-151): per sentence the sorted unique
 * softmax rows of its lattice, united with rows 0 .. top_samples-1 (top_sampling).
 * vs_off has n_sent + 1 entries.  Returns the total length (retry if > cap). */
int64_t jlm_static_vocab(const int32_t *node_word, const int32_t *node_sent, int64_t n_nodes,
                         int32_t n_sent, int32_t top_samples, int64_t cap,
                         int32_t *vs_words, int32_t *vs_off, int32_t n_threads);

/* The same from the lattice's own cell structure (host ABI 3): sentence s owns the nodes
 * end_off[f * n_sent + s] .. end_off[f * n_sent + s + 1] of every frame f -- no pass over node_sent, no
 * per-sentence buckets; the words set bits in a bitmap over the ids and are read back in order (no sort). */
int64_t jlm_static_vocab_cells(const int32_t *node_word, const int32_t *end_off, int32_t n_sent, int32_t n_frames,
                               int32_t top_samples, int64_t cap, int32_t *vs_words, int32_t *vs_off);

/* Word lists of the incremental-vocabulary decoder (decoder_dynamic.py:30-46,112-127) for a
 * batch lattice (node_word / end_off from jlm_lattice_build).  With lv[k] the reference's
 * cumulative per-frame vocabulary list and delta[i] = sorted(set(lv[i]) - set(lv[i-1])):
 *   init list  of cell (k, s) = lv[k] + delta[k+1]   (what frame k's rows are first normalised over;
 *                                frame 0 keeps duplicated sampled ids, as the reference does)
 *   delta list of cell (i, s) = delta[i]             (appended to every older frame at step i)
 * Because lv[] is cumulative, every init list is a slice of one per-sentence sequence
 *   seq_s = surplus copies of lv[0]'s duplicates ++ sorted(set(lv[0])) ++ delta[1] ++ ... ++ delta[L]
 * (same multiset as the reference's list; the order inside a log-sum-exp is free): the call
 * returns the sequences back to back in seq_words and, per cell c = k * n_sent + s, the slice
 *   init_range[2c] .. init_range[2c+1]   (empty for k >= sent_len[s])
 * -- O(L) words per sentence instead of O(L^2).  The delta lists are a CSR over cells
 * (delta_off: n_frames*n_sent + 1 entries).
 * extra_ids / extra_off (may be NULL): per-sentence sampled ids appended to lv[0]
 * (top_sampling: 0..samples-1; random_sampling: the caller's np.random draw).
 * Returns the total sequence length; *delta_total the delta total; if either exceeds its
 * capacity only init_range / delta_off are valid (retry with larger arrays). */
int64_t jlm_dynamic_vocab(const int32_t *node_word, const int32_t *end_off, const int32_t *sent_len,
                          int32_t n_sent, int32_t n_frames,
                          const int32_t *extra_ids, const int32_t *extra_off,
                          int64_t seq_cap, int64_t delta_cap,
                          int32_t *seq_words, int32_t *init_range,
                          int32_t *delta_words, int32_t *delta_off, int64_t *delta_total,
                          int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif /* JLM_HOST_H */
