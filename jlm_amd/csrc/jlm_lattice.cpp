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
    // node 0 = root.  children: hash (parent << 32 | cp) -> child id
    std::vector<uint64_t> keys;      // 0 = empty slot (key is stored + 1)
    std::vector<int32_t> vals;
    uint64_t mask = 0;
    std::vector<int32_t> ent_off;    // per trie node: entries [ent_off[n], ent_off[n+1])
    std::vector<int32_t> ent_word, ent_lex;
    int32_t eos_word = 0, unk_word = 0, max_len = 1;

    static uint64_t mix(uint64_t x) {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
        return x;
    }
    int32_t child(int32_t parent, uint32_t cp) const {
        const uint64_t key = (((uint64_t)(uint32_t)parent) << 32 | cp) + 1;
        for (uint64_t i = mix(key) & mask;; i = (i + 1) & mask) {
            if (keys[i] == key) return vals[i];
            if (keys[i] == 0) return -1;
        }
    }
    void insert(int32_t parent, uint32_t cp, int32_t id) {
        const uint64_t key = (((uint64_t)(uint32_t)parent) << 32 | cp) + 1;
        uint64_t i = mix(key) & mask;
        while (keys[i] != 0) i = (i + 1) & mask;
        keys[i] = key;
        vals[i] = id;
    }
};

struct SentNodes {
    std::vector<int32_t> end, start, word, lex;   // generation order
};

void sentence_nodes(const Trie &t, const uint32_t *text, int L, SentNodes &out) {
    out.end.clear(); out.start.clear(); out.word.clear(); out.lex.clear();
    out.end.push_back(0); out.start.push_back(-1); out.word.push_back(t.eos_word); out.lex.push_back(-1);
    std::vector<char> has(L + 2, 0);
    for (int i = 0; i < L; ++i) {
        int32_t n = 0;
        const int jmax = std::min(L - i, (int)t.max_len);
        for (int j = 0; j < jmax; ++j) {
            if (n >= 0) n = t.child(n, text[i + j]);
            if (n >= 0) {
                for (int32_t e = t.ent_off[n]; e < t.ent_off[n + 1]; ++e) {
                    out.end.push_back(i + j + 1); out.start.push_back(i);
                    out.word.push_back(t.ent_word[e]); out.lex.push_back(t.ent_lex[e]);
                    has[i + j + 1] = 1;
                }
            }
            if (j == 0 && !has[i + 1]) {          // raw-symbol fallback, decoder.py:128-130
                out.end.push_back(i + 1); out.start.push_back(i);
                out.word.push_back(t.unk_word); out.lex.push_back(-2);
                has[i + 1] = 1;
            }
            if (n < 0 && j > 0) break;            // no longer reading can match
        }
    }
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
    t.keys.assign(cap, 0); t.vals.assign(cap, 0); t.mask = cap - 1;
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
    std::vector<SentNodes> sn(B);
    parallel_for(B, n_threads, [&](int s) { sentence_nodes(t, text + text_off[s], text_off[s + 1] - text_off[s], sn[s]); });
    const int64_t ncell = (int64_t)F * B;
    std::vector<int32_t> ecount(ncell, 0), scount(ncell, 0);
    int64_t total = 0;
    for (int s = 0; s < B; ++s) {
        const SentNodes &x = sn[s];
        total += (int64_t)x.end.size();
        for (size_t i = 0; i < x.end.size(); ++i) {
            ++ecount[(int64_t)x.end[i] * B + s];
            if (x.start[i] >= 0) ++scount[(int64_t)x.start[i] * B + s];
        }
    }
    end_off[0] = 0; sg_off[0] = 0;
    int32_t mx = 0;
    for (int64_t c = 0; c < ncell; ++c) {
        end_off[c + 1] = end_off[c] + ecount[c];
        sg_off[c + 1] = sg_off[c] + scount[c];
        if (ecount[c] > mx) mx = ecount[c];
    }
    *max_nodes_per_cell = mx;
    if (total > node_cap) return total;
    // scatter in generation order: stable inside an (end frame, sentence) cell
    std::vector<int32_t> ecur(end_off, end_off + ncell);
    parallel_for(B, n_threads, [&](int s) {
        const SentNodes &x = sn[s];
        for (size_t i = 0; i < x.end.size(); ++i) {
            const int32_t id = ecur[(int64_t)x.end[i] * B + s]++;
            node_start[id] = x.start[i]; node_word[id] = x.word[i]; node_lex[id] = x.lex[i];
            node_sent[id] = s; node_end[id] = x.end[i];
        }
    });
    // nodes grouped by the cell they START in, ascending node id inside a cell
    std::vector<int32_t> scur(sg_off, sg_off + ncell);
    for (int64_t id = 0; id < total; ++id) {
        if (node_start[id] < 0) continue;
        const int32_t o = scur[(int64_t)node_start[id] * B + node_sent[id]]++;
        sg_node[o] = (int32_t)id; sg_word[o] = node_word[id];
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
