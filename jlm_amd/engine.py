"""Frame-synchronous batched lattice decode on one MI355X.

All sentences of a batch advance one kana frame at a time; every frame is a
fixed sequence of kernel launches on one HIP stream with no host
synchronisation until the n-best read-out:

  static (Decoder.decode, reference decoder.py:220-241), frame f:
    beam_step(f)                     K7+K8  candidates, stable top-k, back pointers
    lstm_step(live rows of f)        K1+K2+K3+K9 fused gate GEMM
    project_T                        K4 (+ V_table projections)
    full_vocab_lse | wordlist_lse    K5+K6 fused log-normaliser (never the logits)
    edge_logits(nodes starting at f) logits the lattice can consume next

  dynamic (DynamicDecoder.decode, reference decoder_dynamic.py:177-194), frame f:
    wordlist_lse(merge) on frames <= f-2 with the words new at f   (K11)
    beam_step(f, mode 2)             re-scores every path from the head (K12)
    lstm_step / project_T / wordlist_lse(init list of f) / edge_logits

The reference evaluates frame i-1 lazily at step i (decoder_dynamic.py:130);
stepping it eagerly is the same arithmetic.  The final frame is never stepped
(decoder_dynamic.py never does; decoder.py does and discards the result).
"""
import numpy as np

from . import _lib
from .model import _Stamp


class DecodeEngine:
    def __init__(self, dev_model):
        self.m = dev_model
        self.torch = dev_model.torch
        self.device = dev_model.device
        self.last_timing = None
        self.last_state = None
        self.recorder = None            # optional model.KernelRecorder (bench.py)
        self.last_n_live = None

    def _upload_ints(self, arrays):
        """One H2D copy for a dict of int32 arrays -> dict of device pointers."""
        torch = self.torch
        names = list(arrays)
        sizes = [int(arrays[n].size) for n in names]
        offs = np.zeros(len(names) + 1, dtype=np.int64)
        np.cumsum([(s + 3) // 4 * 4 for s in sizes], out=offs[1:])
        host = np.zeros(int(offs[-1]) + 4, dtype=np.int32)
        for n, o, s in zip(names, offs[:-1], sizes):
            host[o:o + s] = arrays[n].reshape(-1)
        buf = torch.from_numpy(host).to(self.device)
        base = buf.data_ptr()
        return buf, {n: base + 4 * int(o) for n, o in zip(names, offs[:-1])}

    def decode(self, lat, kind="static", vocab=None, dyn_lists=None, topN=10, timing=False, keep_state=False):
        """lat: BatchLattice.  kind: 'static' | 'dynamic'.
        vocab: (words, off) CSR of per-sentence selected vocabularies (static
        vocab_select) or None for the full vocabulary.
        dyn_lists: (init_words, init_off, delta_words, delta_off) for 'dynamic'.
        -> list (per sentence) of [(neg_log_prob, [word, ...])][:topN]"""
        torch, m, L = self.torch, self.m, _lib.lib()
        B, beam, F = lat.n_sent, lat.beam, lat.n_frames
        if B == 0:
            return []
        rmax = B * beam
        G = F * rmax
        ncell = F * B
        dev = self.device
        st = m.stream()
        self_norm = m.self_norm
        dynamic = kind == "dynamic"
        mode = 1 if self_norm else (2 if dynamic else 0)

        ints = dict(sent_len=lat.sent_len, end_off=lat.end_off, node_start=lat.node_start, node_word=lat.node_word,
                    sg_off=lat.sg_off, sg_word=lat.sg_word, sg_node=lat.sg_node,
                    g0=(np.arange(F, dtype=np.int32)[:, None] * rmax + np.arange(B, dtype=np.int32)[None, :] * beam),
                    cidx=np.arange(ncell, dtype=np.int32),
                    sidx=np.tile(np.arange(B, dtype=np.int32), F))
        if vocab is not None:
            ints["vs_words"], ints["vs_off"] = vocab
        if dynamic:
            ints["di_words"], ints["di_off"], ints["dd_words"], ints["dd_off"] = dyn_lists
        ibuf, ip = self._upload_ints(ints)

        H, ldt = m.H, m.ldt
        f64, f32, i32 = torch.float64, torch.float32, torch.int32
        score = torch.empty(G, device=dev, dtype=f64)
        lse = torch.empty(G, device=dev, dtype=f64)
        ysum = torch.empty(G, device=dev, dtype=f64) if dynamic else None
        bp = torch.empty(G, device=dev, dtype=i32)
        node = torch.empty(G, device=dev, dtype=i32)
        word = torch.empty(G, device=dev, dtype=i32)
        cnt = torch.zeros(ncell, device=dev, dtype=i32)
        live = torch.empty(G, device=dev, dtype=i32)
        n_live = torch.zeros(F, device=dev, dtype=i32)
        edge = torch.empty(max(lat.n_nodes, 1) * beam, device=dev, dtype=f32)
        h = torch.empty((G, H), device=dev, dtype=f32)
        c = torch.empty((G, H), device=dev, dtype=f32)
        T = h if m.mode == "untied" else torch.empty((G, ldt), device=dev, dtype=f32)
        use_wordlist = (vocab is not None) or dynamic
        if not self_norm:
            if use_wordlist:
                run_max = torch.empty(G, device=dev, dtype=f32)
                run_sum = torch.empty(G, device=dev, dtype=f64)
            else:
                n_part = max(m.n_vocab_tiles, 1)
                part = torch.empty((n_part, rmax, 2), device=dev, dtype=f32)

        latS = _lib.Lattice(B, beam, F, ip["sent_len"], ip["end_off"], ip["node_start"], ip["node_word"])
        stS = _lib.BeamState(score.data_ptr(), lse.data_ptr(), ysum.data_ptr() if dynamic else None,
                             bp.data_ptr(), node.data_ptr(), word.data_ptr(), cnt.data_ptr(), live.data_ptr(),
                             n_live.data_ptr(), edge.data_ptr())
        hp, cp, Tp = h.data_ptr(), c.data_ptr(), T.data_ptr()
        bpp, wordp, cntp = bp.data_ptr(), word.data_ptr(), cnt.data_ptr()
        livep, nlivep, lsep = live.data_ptr(), n_live.data_ptr(), lse.data_ptr()
        b2p = m.b2.data_ptr()
        segs, nsegs = m.seg_array, m.n_segs
        ev = []
        for f in range(F):
            if dynamic and not self_norm and f >= 2:
                # K11: older frames learn the words that first appear at frame f
                _lib.check(L.jlm_wordlist_lse(segs, nsegs, b2p, Tp, ldt, ip["g0"], cntp, ip["cidx"],
                                              ip["dd_words"], ip["dd_off"], ip["sidx"], f * B,
                                              run_max.data_ptr(), run_sum.data_ptr(), lsep, 1, beam, (f - 1) * B, st),
                           "jlm_wordlist_lse(merge)")
            _lib.check(L.jlm_beam_step(latS, stS, f, mode, lat.max_cands, st), "jlm_beam_step")
            if f == F - 1:
                break
            rows = livep + 4 * f * rmax
            ndev = nlivep + 4 * f
            if timing:
                e0, e1, e2 = (_Stamp(torch, dev) for _ in range(3))
                e0.record()
            m.lstm_step(hp, cp, H, hp, cp, rows, bpp, wordp, rmax, ndev, st, self.recorder)
            if timing:
                e1.record()
            m.project_T(hp, H, Tp, rows, rmax, ndev, st)
            cell = 4 * f * B
            if not self_norm:
                if dynamic:
                    _lib.check(L.jlm_wordlist_lse(segs, nsegs, b2p, Tp, ldt, ip["g0"] + cell, cntp, ip["cidx"] + cell,
                                                  ip["di_words"], ip["di_off"], ip["sidx"], f * B,
                                                  run_max.data_ptr(), run_sum.data_ptr(), lsep, 0, beam, B, st),
                               "jlm_wordlist_lse(init)")
                elif vocab is not None:
                    _lib.check(L.jlm_wordlist_lse(segs, nsegs, b2p, Tp, ldt, ip["g0"] + cell, cntp, ip["cidx"] + cell,
                                                  ip["vs_words"], ip["vs_off"], ip["sidx"], 0,
                                                  run_max.data_ptr(), run_sum.data_ptr(), lsep, 0, beam, B, st),
                               "jlm_wordlist_lse(vocab_select)")
                else:
                    m.full_vocab_lse(Tp, rows, part.data_ptr(), rmax, n_part, lsep, rmax, ndev, st, self.recorder)
            _lib.check(L.jlm_edge_logits(segs, nsegs, b2p, Tp, ldt, ip["g0"] + cell, cntp, ip["cidx"] + cell,
                                         ip["sg_word"], ip["sg_off"], ip["sidx"], f * B, ip["sg_node"],
                                         edge.data_ptr(), beam, B, st), "jlm_edge_logits")
            if timing:
                e2.record()
                ev.append((e0, e1, e2))

        stride = F + 1
        out_nodes = torch.empty((rmax, stride), device=dev, dtype=i32)
        out_len = torch.empty(rmax, device=dev, dtype=i32)
        out_score = torch.empty(rmax, device=dev, dtype=f64)
        _lib.check(L.jlm_backtrace(latS, stS, out_nodes.data_ptr(), out_len.data_ptr(), out_score.data_ptr(), stride, st),
                   "jlm_backtrace")
        if self.recorder is not None:
            self.last_n_live = n_live.cpu().numpy()
        nodes_h = out_nodes.cpu().numpy()
        len_h = out_len.cpu().numpy()
        score_h = out_score.cpu().numpy()
        if timing:
            self.last_timing = [(a.seconds_to(b), b.seconds_to(c2)) for a, b, c2 in ev]
        if keep_state:
            self.last_state = dict(score=score, lse=lse, ysum=ysum, bp=bp, node=node, word=word, cnt=cnt, live=live,
                                   n_live=n_live, edge=edge, h=h, c=c, T=T, ints=ibuf)
        return self._read_out(lat, nodes_h, len_h, score_h, topN)

    @staticmethod
    def _read_out(lat, nodes_h, len_h, score_h, topN):
        B, beam = lat.n_sent, lat.beam
        # flatten every path (reversed: last word first), drop the <eos> root
        sel, lens = [], []
        for s in range(B):
            for r in range(min(beam, topN)):
                i = s * beam + r
                n = int(len_h[i])
                if n == 0:
                    break
                sel.append((i, n))
        if sel:
            flat = np.concatenate([nodes_h[i, :n - 1][::-1] for i, n in sel]) if any(n > 1 for _, n in sel) else np.zeros(0, np.int64)
            words = lat.words_of(flat) if flat.size else np.zeros(0, dtype=object)
        out = [[] for _ in range(B)]
        pos = 0
        for i, n in sel:
            k = n - 1
            ws = words[pos:pos + k].tolist() if k else []
            pos += k
            out[i // beam].append((float(score_h[i]), ws))
        return out
