// Native (host, multi-threaded) word-lattice builder: the C++ counterpart of
// jlm_amd/lattice.py's BatchLattice, i.e. of Decoder._build_lattice /
// _build_lattice_vocab in the reference (decoder/decoder.py:79-151).  C ABI in
// include/jlm_host.h; no GPU, no torch.
//
// The reading dictionary becomes a trie over code points held in one flat
// open-addressing hash table keyed by (parent node, code point): the reference's
// O(L^2) substring look-ups become one trie walk per start position.  Entries of a
// reading are pre-filtered to in-vocabulary words and pre-sorted by lexicon id,
// which is the order the reference adds nodes in (decoder.py:95-126).
#include <algorithm>
#include <iterator>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/jlm_host.h"

namespace {

struct Trie {
    // node 0 = root.  children: open-addressed table (parent << 32 | cp) -> child id; key and value share a 16-byte slot (one
    // cache line per probe), multiplicative hash
    struct Slot { uint64_t key; int32_t val; int32_t pad; };      // key 0 = empty (keys are stored + 1)
    std::vector<Slot> slots;
    int shift = 64;
    uint64_t mask = 0;
    std::vector<int32_t> ent_off;    // per trie node: entries [ent_off[n], ent_off[n+1])
    std::vector<int32_t> ent_word, ent_lex;
    int32_t eos_word = 0, unk_word = 0, max_len = 1;

    void resize(uint64_t cap) {      // cap: a power of two
        slots.assign(cap, Slot{0, 0, 0});
        mask = cap - 1;
        shift = 64;
        while ((1ull << (64 - shift)) < cap) --shift;
    }
    uint64_t home(uint64_t key) const { return (key * 0x9E3779B97F4A7C15ULL) >> shift; }
    int32_t child(int32_t parent, uint32_t cp) const {
        const uint64_t key = (((uint64_t)(uint32_t)parent) << 32 | cp) + 1;
        for (uint64_t i = home(key);; i = (i + 1) & mask) {
            const Slot &sl = slots[i];
            if (sl.key == key) return sl.val;
            if (sl.key == 0) return -1;
        }
    }
    void insert(int32_t parent, uint32_t cp, int32_t id) {
        const uint64_t key = (((uint64_t)(uint32_t)parent) << 32 | cp) + 1;
        uint64_t i = home(key);
        while (slots[i].key != 0) i = (i + 1) & mask;
        slots[i].key = key;
        slots[i].val = id;
    }
};

// One dictionary match of a sentence: the reading text[start, end) ends at trie node `tnode` (tnode == -2: the raw-symbol
// fallback of decoder.py:128-130, one node).  A match expands to the entries of its trie node -- the lattice nodes -- all in
// the cell (end frame, sentence); first_id = the global id of its first node.
struct Match {
    int32_t start, end, tnode, first_id;
};

struct SentMatches {
    std::vector<Match> m;            // generation order: start ascending, length ascending
    std::vector<int32_t> by_end;     // indices into m, stably sorted by end
    std::vector<int32_t> n_end, n_start;     // lattice nodes per end frame (0 .. L) / per start frame (0 .. L - 1)
};

inline int32_t match_count(const Trie &t, const Match &x) { return x.tnode < 0 ? 1 : t.ent_off[x.tnode + 1] - t.ent_off[x.tnode]; }

// the walk of decoder.py:104-135 over one sentence: matches instead of nodes (a match is 16 bytes, its nodes 20 bytes EACH)
void sentence_matches(const Trie &t, const uint32_t *text, int L, SentMatches &out) {
    out.m.clear();
    out.m.reserve((size_t)L * 8);
    out.n_end.assign(L + 1, 0);
    out.n_start.assign(L + 1, 0);
    out.n_end[0] = 1;                                    // the <eos> root
    for (int i = 0; i < L; ++i) {
        int32_t n = 0;
        const int jmax = std::min(L - i, (int)t.max_len);
        for (int j = 0; j < jmax; ++j) {
            if (n >= 0) n = t.child(n, text[i + j]);
            if (n >= 0) {
                const int32_t c = t.ent_off[n + 1] - t.ent_off[n];
                if (c > 0) {
                    out.m.push_back(Match{i, i + j + 1, n, 0});
                    out.n_end[i + j + 1] += c;
                    out.n_start[i] += c;
                }
            }
            if (j == 0 && !out.n_end[i + 1]) {         // raw-symbol fallback: no node ends behind this symbol yet
                out.m.push_back(Match{i, i + 1, -2, 0});
                out.n_end[i + 1] += 1;
                out.n_start[i] += 1;
            }
            if (n < 0 && j > 0) break;                  // no longer reading can match
        }
    }
    // stable counting sort of the matches by end frame
    const int M = (int)out.m.size();
    std::vector<int32_t> pos(L + 2, 0);
    for (int k = 0; k < M; ++k) ++pos[out.m[k].end + 1];
    for (int e = 0; e <= L; ++e) pos[e + 1] += pos[e];
    out.by_end.resize(M);
    for (int k = 0; k < M; ++k) out.by_end[pos[out.m[k].end]++] = k;
}

template <class F>
void parallel_for(int n, int n_threads, F f) {
    if (n_threads <= 1 || n < 2 * n_threads) { for (int i = 0; i < n; ++i) f(i); return; }
    std::vector<std::thread> th;
    for (int w = 0; w < n_threads; ++w)
        th.emplace_back([=]() { for (int i = w; i < n; i += n_threads) f(i); });
    for (auto &x : th) x.join();
}

}  // namespace

struct jlm_lexicon {
    Trie t;
};

extern "C" jlm_lexicon *jlm_lexicon_create(const uint32_t *reading_cp, const int32_t *reading_off, const int32_t *entry_off,
                                           const int32_t *entry_word, const int32_t *entry_lex, int32_t n_readings,
                                           int32_t eos_word, int32_t unk_word) {
    auto *lx = new jlm_lexicon();
    Trie &t = lx->t;
    t.eos_word = eos_word; t.unk_word = unk_word;
    const int64_t total_cp = reading_off[n_readings];
    uint64_t cap = 16;
    while (cap < (uint64_t)(total_cp + 1) * 2) cap <<= 1;
    t.resize(cap);
    int32_t n_nodes = 1;
    std::vector<int32_t> node_of_reading(n_readings);
    for (int32_t r = 0; r < n_readings; ++r) {
        int32_t n = 0;
        const int len = reading_off[r + 1] - reading_off[r];
        if (len > t.max_len) t.max_len = len;
        for (int k = 0; k < len; ++k) {
            const uint32_t cp = reading_cp[reading_off[r] + k];
            int32_t c = t.child(n, cp);
            if (c < 0) { c = n_nodes++; t.insert(n, cp, c); }
            n = c;
        }
        node_of_reading[r] = n;
    }
    // entries per trie node (a reading is unique, so a node gets at most one reading's entries)
    std::vector<int32_t> cnt(n_nodes + 1, 0);
    for (int32_t r = 0; r < n_readings; ++r) cnt[node_of_reading[r] + 1] += entry_off[r + 1] - entry_off[r];
    t.ent_off.assign(n_nodes + 1, 0);
    for (int32_t n = 0; n < n_nodes; ++n) t.ent_off[n + 1] = t.ent_off[n] + cnt[n + 1];
    t.ent_word.resize(t.ent_off[n_nodes]); t.ent_lex.resize(t.ent_off[n_nodes]);
    for (int32_t r = 0; r < n_readings; ++r) {
        int32_t o = t.ent_off[node_of_reading[r]];
        for (int32_t e = entry_off[r]; e < entry_off[r + 1]; ++e, ++o) { t.ent_word[o] = entry_word[e]; t.ent_lex[o] = entry_lex[e]; }
    }
    return lx;
}

extern "C" void jlm_lexicon_destroy(jlm_lexicon *lx) { delete lx; }

extern "C" int64_t jlm_lattice_build(const jlm_lexicon *lx, const uint32_t *text, const int32_t *text_off, int32_t n_sent,
                                     int32_t n_frames, int64_t node_cap, int32_t *node_start, int32_t *node_word,
                                     int32_t *node_lex, int32_t *node_sent, int32_t *node_end, int32_t *end_off,
                                     int32_t *sg_off, int32_t *sg_node, int32_t *sg_word, int32_t *max_nodes_per_cell,
                                     int32_t n_threads) {
    const Trie &t = lx->t;
    const int B = n_sent, F = n_frames;
    // Two passes.  1: every sentence's dictionary matches and its node counts per end / start frame -> the cell offsets
    // (end_off, sg_off: cell = frame * n_sent + sentence).  2: every sentence writes its nodes, cell by cell -- a cell's nodes are
    // one contiguous run of each output array (a node-by-node scatter of 175 k nodes over five arrays was 1.4 ms of a 4.7-ms
    // build, growing four vectors per sentence node by node another 1.2) -- and its start-grouped list (sg_*) the same way.
    std::vector<SentMatches> sm(B);
    parallel_for(B, n_threads, [&](int s) { sentence_matches(t, text + text_off[s], text_off[s + 1] - text_off[s], sm[s]); });
    const int64_t ncell = (int64_t)F * B;
    int64_t total = 0;
    int32_t mx = 0;
    end_off[0] = 0; sg_off[0] = 0;
    for (int f = 0; f < F; ++f)
        for (int s = 0; s < B; ++s) {
            const int L = text_off[s + 1] - text_off[s];
            const int32_t ce = f <= L ? sm[s].n_end[f] : 0, cs = f < L ? sm[s].n_start[f] : 0;
            const int64_t c = (int64_t)f * B + s;
            end_off[c + 1] = end_off[c] + ce;
            sg_off[c + 1] = sg_off[c] + cs;
            if (ce > mx) mx = ce;
            total += ce;
        }
    (void)ncell;
    *max_nodes_per_cell = mx;
    if (total > node_cap) return total;
    // cells in MEMORY order (frame-major, sentence-minor): every output array is written front to back -- sentence by sentence the
    // 37 k runs of a batch landed 35 KB apart in seven arrays: 1.07 ms of a 1.65-ms build
    std::vector<uint32_t> kpos(B, 0), qpos(B, 0);
    for (int s = 0; s < B; ++s) {                          // the roots: frame 0's only nodes
        const int32_t id = end_off[s];
        node_start[id] = -1; node_word[id] = t.eos_word; node_lex[id] = -1; node_sent[id] = s; node_end[id] = 0;
    }
    for (int e = 1; e < F; ++e)
        for (int s = 0; s < B; ++s) {
            const int L = text_off[s + 1] - text_off[s];
            if (e > L) continue;
            SentMatches &x = sm[s];
            uint32_t k = kpos[s];
            int32_t id = end_off[(int64_t)e * B + s];
            for (; k < x.by_end.size() && x.m[x.by_end[k]].end == e; ++k) {
                Match &m = x.m[x.by_end[k]];
                m.first_id = id;
                if (m.tnode < 0) {
                    node_start[id] = m.start; node_word[id] = t.unk_word; node_lex[id] = -2; node_sent[id] = s; node_end[id] = e;
                    ++id;
                } else {
                    for (int32_t q = t.ent_off[m.tnode]; q < t.ent_off[m.tnode + 1]; ++q, ++id) {
                        node_start[id] = m.start; node_word[id] = t.ent_word[q]; node_lex[id] = t.ent_lex[q];
                        node_sent[id] = s; node_end[id] = e;
                    }
                }
            }
            kpos[s] = k;
        }
    // nodes grouped by the cell they START in, ascending node id inside a cell: a start's matches in length order ARE in
    // ascending id order (a longer reading ends in a later frame, i.e. a later cell)
    for (int i = 0; i + 1 < F; ++i)
        for (int s = 0; s < B; ++s) {
            const int L = text_off[s + 1] - text_off[s];
            if (i >= L) continue;
            const SentMatches &x = sm[s];
            uint32_t q = qpos[s];
            int32_t o = sg_off[(int64_t)i * B + s];
            for (; q < x.m.size() && x.m[q].start == i; ++q) {
                const Match &m = x.m[q];
                const int32_t c = match_count(t, m);
                for (int32_t r = 0; r < c; ++r, ++o) { sg_node[o] = m.first_id + r; sg_word[o] = node_word[m.first_id + r]; }
            }
            qpos[s] = q;
        }
    return total;
}

// per sentence sorted unique word ids over its lattice (+ the first `top_samples` ids), decoder.py:137-151
extern "C" int64_t jlm_static_vocab(const int32_t *node_word, const int32_t *node_sent, int64_t n_nodes, int32_t n_sent,
                                    int32_t top_samples, int64_t cap, int32_t *vs_words, int32_t *vs_off, int32_t n_threads) {
    // Sorted unique ids per sentence WITHOUT a sort: the sentence's words set bits in a bitmap over the word ids, the set
    // bits are read back in order.  (A bucket per sentence + std::sort + std::unique was 4.2 ms per 256 x 20-kana batch --
    // more than the GPU needs for the batch, and what made Decoder.decode_batch(vocab_select=True) host-bound.)
    (void)n_threads;
    // 1. the words of every sentence, contiguous (counting sort by sentence: the nodes come cell by cell)
    std::vector<int64_t> soff((size_t)n_sent + 1, 0);
    int32_t wmax = top_samples > 0 ? top_samples - 1 : 0;
    for (int64_t i = 0; i < n_nodes; ++i) {
        ++soff[(size_t)node_sent[i] + 1];
        if (node_word[i] > wmax) wmax = node_word[i];
    }
    for (int s = 0; s < n_sent; ++s) soff[s + 1] += soff[s];
    std::vector<int32_t> flat((size_t)n_nodes);
    {
        std::vector<int64_t> cur(soff.begin(), soff.end() - 1);
        for (int64_t i = 0; i < n_nodes; ++i) flat[(size_t)cur[node_sent[i]]++] = node_word[i];
    }
    // 2. per sentence: set, scan, clear
    std::vector<uint64_t> bits((size_t)wmax / 64 + 1, 0);
    const int64_t top_words = ((int64_t)top_samples + 63) / 64;
    int64_t total = 0;
    vs_off[0] = 0;
    for (int s = 0; s < n_sent; ++s) {
        int64_t lo = (int64_t)bits.size(), hi = -1;
        for (int64_t i = soff[s]; i < soff[s + 1]; ++i) {
            const int64_t w = flat[(size_t)i] >> 6;
            bits[(size_t)w] |= 1ull << (flat[(size_t)i] & 63);
            if (w < lo) lo = w;
            if (w > hi) hi = w;
        }
        if (top_samples > 0) {                 // + range(samples), decoder.py:143-146
            for (int64_t w = 0; w < top_words; ++w) {
                const int64_t left = (int64_t)top_samples - 64 * w;
                bits[(size_t)w] |= left >= 64 ? ~0ull : ((1ull << left) - 1);
            }
            lo = 0;
            if (top_words - 1 > hi) hi = top_words - 1;
        }
        for (int64_t w = lo; w <= hi; ++w) {
            uint64_t x = bits[(size_t)w];
            if (!x) continue;
            bits[(size_t)w] = 0;
            while (x) {
                const int b = __builtin_ctzll(x);
                x &= x - 1;
                if (total < cap) vs_words[total] = (int32_t)(64 * w + b);
                ++total;
            }
        }
        vs_off[s + 1] = (int32_t)total;
    }
    return total;
}

extern "C" int64_t jlm_static_vocab_cells(const int32_t *node_word, const int32_t *end_off, int32_t n_sent, int32_t n_frames,
                                          int32_t top_samples, int64_t cap, int32_t *vs_words, int32_t *vs_off) {
    const int B = n_sent, F = n_frames;
    const int64_t n_nodes = end_off[(int64_t)F * B];
    int32_t wmax = top_samples > 0 ? top_samples - 1 : 0;
    for (int64_t i = 0; i < n_nodes; ++i)
        if (node_word[i] > wmax) wmax = node_word[i];
    std::vector<uint64_t> bits((size_t)wmax / 64 + 1, 0);
    const int64_t top_words = ((int64_t)top_samples + 63) / 64;
    int64_t total = 0;
    vs_off[0] = 0;
    for (int s = 0; s < B; ++s) {
        int64_t lo = (int64_t)bits.size(), hi = -1;
        for (int f = 0; f < F; ++f) {
            const int64_t c = (int64_t)f * B + s;
            for (int32_t i = end_off[c]; i < end_off[c + 1]; ++i) {
                const int32_t id = node_word[i];
                const int64_t w = id >> 6;
                bits[(size_t)w] |= 1ull << (id & 63);
                if (w < lo) lo = w;
                if (w > hi) hi = w;
            }
        }
        if (top_samples > 0) {                 // + range(samples), decoder.py:143-146
            for (int64_t w = 0; w < top_words; ++w) {
                const int64_t left = (int64_t)top_samples - 64 * w;
                bits[(size_t)w] |= left >= 64 ? ~0ull : ((1ull << left) - 1);
            }
            lo = 0;
            if (top_words - 1 > hi) hi = top_words - 1;
        }
        for (int64_t w = lo; w <= hi; ++w) {
            uint64_t x = bits[(size_t)w];
            if (!x) continue;
            bits[(size_t)w] = 0;
            while (x) {
                const int b = __builtin_ctzll(x);
                x &= x - 1;
                if (total < cap) vs_words[total] = (int32_t)(64 * w + b);
                ++total;
            }
        }
        vs_off[s + 1] = (int32_t)total;
    }
    return total;
}

// Word lists of the incremental-vocabulary decoder (decoder_dynamic.py:30-46,112-127), see
// include/jlm_host.h.  Every list the device needs is a slice of ONE per-sentence sequence
//   seq_s = dup(lv0) ++ uniq(lv0) ++ delta[1] ++ delta[2] ++ ... ++ delta[L]
// (lv0 = frame 0's words + the sampled ids, which the reference does not de-duplicate; dup() = its
// surplus copies), because lv[k] is cumulative: as a multiset, the list frame k's rows are first
// normalised over, lv[k] + delta[k+1], is seq_s[n_dup .. end of delta[k+1]) for k >= 1 and
// seq_s[0 .. end of delta[1]) for k = 0.  Materialising the lists themselves is O(L^2) words per
// sentence (1.7 M ints and 15 ms of host time per 256 x 20-kana batch); the sequence is O(L).
namespace {
struct DynSeq {
    std::vector<int32_t> seq;        // the sequence above
    std::vector<int32_t> cum;        // cum[i] = end of delta[i] inside seq (cum[0] = end of uniq(lv0))
    int32_t n_dup = 0;
};
// "seen" is a per-thread stamp array over word ids (stamp == this sentence's ticket: already in the
// vocabulary), so a frame costs its own words only -- no merge of the whole cumulative set per frame.
void dyn_sentence(const int32_t *node_word, const int32_t *end_off, int B, int s, int L, const int32_t *extra, int n_extra,
                  DynSeq &out) {
    thread_local std::vector<uint32_t> stamp;
    thread_local uint32_t ticket = 0;
    if (++ticket == 0) { std::fill(stamp.begin(), stamp.end(), 0u); ticket = 1; }
    auto seen = [&](int32_t w) -> bool {              // marks w; true if it was marked before
        if ((size_t)w >= stamp.size()) stamp.resize((size_t)w + 1024, 0u);
        if (stamp[w] == ticket) return true;
        stamp[w] = ticket;
        return false;
    };
    std::vector<int32_t> lv0, fw;
    lv0.assign(node_word + end_off[s], node_word + end_off[s + 1]);     // frame 0, NOT de-duplicated (decoder_dynamic.py:34-43)
    lv0.insert(lv0.end(), extra, extra + n_extra);
    std::sort(lv0.begin(), lv0.end());
    out.seq.clear(); out.cum.assign(L + 1, 0);
    for (size_t i = 1; i < lv0.size(); ++i)
        if (lv0[i] == lv0[i - 1]) out.seq.push_back(lv0[i]);          // surplus copies first
    out.n_dup = (int32_t)out.seq.size();
    for (int32_t w : lv0)
        if (!seen(w)) out.seq.push_back(w);                            // sorted(set(lv0))
    out.cum[0] = (int32_t)out.seq.size();
    for (int i = 1; i <= L; ++i) {
        fw.clear();
        for (const int32_t *w = node_word + end_off[(int64_t)i * B + s], *e = node_word + end_off[(int64_t)i * B + s + 1]; w < e; ++w)
            if (!seen(*w)) fw.push_back(*w);
        std::sort(fw.begin(), fw.end());                               // delta[i] = sorted(set(lv[i]) - set(lv[i-1]))
        out.seq.insert(out.seq.end(), fw.begin(), fw.end());
        out.cum[i] = (int32_t)out.seq.size();
    }
}
}  // namespace

extern "C" int64_t jlm_dynamic_vocab(const int32_t *node_word, const int32_t *end_off, const int32_t *sent_len,
                                     int32_t n_sent, int32_t n_frames, const int32_t *extra_ids, const int32_t *extra_off,
                                     int64_t seq_cap, int64_t delta_cap, int32_t *seq_words, int32_t *init_range,
                                     int32_t *delta_words, int32_t *delta_off, int64_t *delta_total, int32_t n_threads) {
    const int B = n_sent, F = n_frames;
    std::vector<DynSeq> all(B);
    parallel_for(B, n_threads, [&](int s) {
        const int32_t *ex = extra_ids ? extra_ids + extra_off[s] : nullptr;
        const int nex = extra_ids ? extra_off[s + 1] - extra_off[s] : 0;
        dyn_sentence(node_word, end_off, B, s, sent_len[s], ex, nex, all[s]);
    });
    std::vector<int64_t> base(B + 1, 0);
    for (int s = 0; s < B; ++s) base[s + 1] = base[s] + (int64_t)all[s].seq.size();
    int64_t td = 0;
    delta_off[0] = 0;
    for (int f = 0; f < F; ++f)
        for (int s = 0; s < B; ++s) {
            const int64_t c = (int64_t)f * B + s;
            const DynSeq &x = all[s];
            const int L = sent_len[s];
            if (f >= 1 && f <= L) td += x.cum[f] - x.cum[f - 1];
            delta_off[c + 1] = (int32_t)td;
            // init list of cell (f, s): lv[f] + delta[f+1]; rows of the last frame are never normalised
            int64_t b = 0, e = 0;
            if (f < L) { b = base[s] + (f == 0 ? 0 : x.n_dup); e = base[s] + x.cum[f + 1]; }
            init_range[2 * c] = (int32_t)b; init_range[2 * c + 1] = (int32_t)e;
        }
    *delta_total = td;
    if (base[B] > seq_cap || td > delta_cap) return base[B];
    parallel_for(B, n_threads, [&](int s) {
        const DynSeq &x = all[s];
        std::memcpy(seq_words + base[s], x.seq.data(), x.seq.size() * sizeof(int32_t));
        for (int f = 1; f <= sent_len[s]; ++f) {
            const int64_t c = (int64_t)f * B + s;
            std::memcpy(delta_words + delta_off[c], x.seq.data() + x.cum[f - 1], (size_t)(x.cum[f] - x.cum[f - 1]) * sizeof(int32_t));
        }
    });
    return base[B];
}

extern "C" int jlm_host_abi_version(void) { return JLM_HOST_ABI_VERSION; }
