"""Frame-synchronous batched lattice decode on one MI355X.

All sentences of a batch advance one kana frame at a time; every frame is a
fixed sequence of kernel launches with no host synchronisation until the
n-best read-out:

  static (Decoder.decode, reference decoder.py:220-241), frame f:
    beam_step(f)                     K7+K8  candidates, stable top-k, back pointers
    lstm_step(live rows of f)        K1+K2+K3+K9 fused gate GEMM
    project_T                        K4 (+ V_table projections)
    full_vocab_lse | wordlist_lse    K5+K6 fused log-normaliser (never the logits)
    edge_logits(nodes starting at f) logits the lattice can consume next
                                     (side stream: independent of the normaliser)

  dynamic (DynamicDecoder.decode, reference decoder_dynamic.py:177-194), frame f:
    wordlist_lse(merge) on frames <= f-2 with the words new at f   (K11)
    beam_step(f, mode 2)             re-scores every path from the head (K12)
    lstm_step / project_T / wordlist_lse(init list of f) / edge_logits

The reference evaluates frame i-1 lazily at step i (decoder_dynamic.py:130);
stepping it eagerly is the same arithmetic.  The final frame is never stepped
(decoder_dynamic.py never does; decoder.py does and discards the result).

Buffers live in per-shape plans that are reused across batches (no allocation or
re-upload of constants per batch).  submit()/collect() split a decode into the
asynchronous device part and the host read-out, so the strings of batch i are
built while the GPU decodes batch i+1 (two plans of the same shape alternate).

For the word-list decodes (vocab_select, incremental vocabulary) and self-normalised models a plan
that has been launched eagerly twice captures its ~170-launch sequence in a hipGraph and replays it
from then on (JLM_GRAPH=0 keeps eager launches): their kernels are short enough for the Python
enqueue loop to be the bottleneck (interleaved A/B, tools/ab_engine.py: incremental 3.08 -> 2.72 ms,
vocab_select 2.82 -> 2.42 ms per step).  The full-vocabulary decode is device bound and stays
eager: with replay its two streams overlap worse (3.60 vs 3.15 ms).  One-off shapes never pay for
a capture.
"""
import os

import numpy as np

from . import _lib
from .model import _Stamp


_READOUT = [False]


def _readout_ext():
    """jlm_amd._readout (built by __graft_entry__.build()), or None; JLM_NATIVE_READOUT=0 disables it."""
    if _READOUT[0] is False:
        ext = None
        if os.environ.get("JLM_NATIVE_READOUT", "1") != "0":
            try:
                from . import _readout as ext
            except ImportError:
                ext = None
        _READOUT[0] = ext
    return _READOUT[0]


def _round_up(x, m):
    return (int(x) + m - 1) // m * m


class _Plan:
    """Device buffers + captured graph for one decode shape."""

    INT_ARRAYS = ("sent_len", "end_off", "node_start", "node_word", "sg_off", "sg_word", "sg_node", "g0", "cidx", "sidx",
                  "sidx2", "vs_words", "vs_off", "di_words", "di_off", "dd_words", "dd_off")

    def __init__(self, eng, key, caps):
        torch, m, dev = eng.torch, eng.m, eng.device
        self.key, self.caps = key, dict(caps)
        kind, vmode, B, beam, F = key[:5]
        self.B, self.beam, self.F = B, beam, F
        rmax, ncell = B * beam, F * B
        G = F * rmax
        self.rmax, self.G, self.ncell = rmax, G, ncell
        sizes = dict(sent_len=B, end_off=ncell + 1, node_start=caps["nodes"], node_word=caps["nodes"], sg_off=ncell + 1,
                     sg_word=caps["nodes"], sg_node=caps["nodes"], g0=ncell, cidx=ncell, sidx=ncell, sidx2=ncell,
                     vs_words=caps["vs"], vs_off=B + 1, di_words=caps["di"], di_off=2 * ncell, dd_words=caps["dd"],
                     dd_off=ncell + 1)
        off, self.ioff = 0, {}
        for n in self.INT_ARRAYS:
            self.ioff[n] = off
            off += _round_up(max(sizes[n], 1), 4)
        self.isize = {n: sizes[n] for n in self.INT_ARRAYS}
        self.host_ints = torch.zeros(off + 4, dtype=torch.int32)
        if dev.type == "cuda":
            self.host_ints = self.host_ints.pin_memory()
        self.host_np = self.host_ints.numpy()
        self.dev_ints = torch.zeros(off + 4, dtype=torch.int32, device=dev)
        base = self.dev_ints.data_ptr()
        self.ip = {n: base + 4 * o for n, o in self.ioff.items()}
        # constant index arrays
        self._set("g0", (np.arange(F, dtype=np.int32)[:, None] * rmax + np.arange(B, dtype=np.int32)[None, :] * beam))
        self._set("cidx", np.arange(ncell, dtype=np.int32))
        self._set("sidx", np.tile(np.arange(B, dtype=np.int32), F))
        # the incremental decoder's frame-initial lists are (begin, end) slices of per-sentence sequences
        # (lattice.dynamic_vocab): list number 2 * cell reads its two offsets from di_off[2 * cell .. + 1]
        self._set("sidx2", 2 * np.tile(np.arange(B, dtype=np.int32), F))
        f64, f32, i32 = torch.float64, torch.float32, torch.int32
        dynamic = kind == "dynamic"
        e = lambda n, dt: torch.empty(n, device=dev, dtype=dt)
        self.score, self.lse = e(G, f64), e(G, f64)
        self.ysum = e(G, f64) if dynamic else None
        self.bp, self.node, self.word = e(G, i32), e(G, i32), e(G, i32)
        self.cnt = torch.zeros(ncell, device=dev, dtype=i32)
        self.live = e(G, i32)
        self.live_base = torch.zeros(ncell, device=dev, dtype=i32)
        self.n_live = torch.zeros(F, device=dev, dtype=i32)
        self.edge = e(max(caps["nodes"], 1) * beam, f32)
        H, ldt = m.H, m.ldt
        self.h, self.c = e((G, H), f32), e((G, H), f32)
        self.T = self.h if m.mode == "untied" else e((G, ldt), f32)
        self.run_max = self.run_sum = self.part = None
        self.n_part = 0
        if not m.self_norm:
            if vmode != "full":
                self.run_max, self.run_sum = e(G, f32), e(G, f64)
            else:
                self.n_part = max(m.n_vocab_tiles, 1)
                self.part = e((self.n_part, rmax, 2), f32)
        self.stride = F + 1
        self.out_nodes = e((rmax, self.stride), i32)
        self.out_len = e(rmax, i32)
        self.out_score = e(rmax, f64)
        pin = (lambda t: t.pin_memory()) if dev.type == "cuda" else (lambda t: t)
        self.h_nodes = pin(torch.empty((rmax, self.stride), dtype=i32))
        self.h_len = pin(torch.empty(rmax, dtype=i32))
        self.h_score = pin(torch.empty(rmax, dtype=f64))
        self.h_nlive = pin(torch.empty(F, dtype=i32))
        self.busy = False
        self.latS = _lib.Lattice(B, beam, F, self.ip["sent_len"], self.ip["end_off"], self.ip["node_start"],
                                 self.ip["node_word"])
        self.stS = _lib.BeamState(self.score.data_ptr(), self.lse.data_ptr(),
                                  self.ysum.data_ptr() if dynamic else None, self.bp.data_ptr(), self.node.data_ptr(),
                                  self.word.data_ptr(), self.cnt.data_ptr(), self.live.data_ptr(), self.n_live.data_ptr(),
                                  self.edge.data_ptr(), self.live_base.data_ptr(), None, 0, 0)
        self.graph = None
        self.warm = False
        self.nbytes = sum(t.numel() * t.element_size() for t in vars(self).values() if isinstance(t, torch.Tensor) and t.device == dev)
        # jlm_decode_plan: the same buffers for the native frame loop (jlm_decode_frames)
        ptr = lambda t: t.data_ptr() if t is not None else None
        d = self.desc = _lib.DecodePlan()
        d.kind = 2 if dynamic else (1 if vmode == "select" else 0)
        d.max_cands = caps["cands"]
        d.h, d.c, d.T = self.h.data_ptr(), self.c.data_ptr(), self.T.data_ptr()
        d.g0, d.cidx, d.sidx = self.ip["g0"], self.ip["cidx"], self.ip["sidx"]
        d.sg_word, d.sg_off, d.sg_node, d.edge = self.ip["sg_word"], self.ip["sg_off"], self.ip["sg_node"], self.edge.data_ptr()
        d.vs_words, d.vs_off = self.ip["vs_words"], self.ip["vs_off"]
        d.di_words, d.di_off, d.di_idx = self.ip["di_words"], self.ip["di_off"], self.ip["sidx2"]
        d.dd_words, d.dd_off = self.ip["dd_words"], self.ip["dd_off"]
        d.run_max, d.run_sum, d.part, d.max_parts = ptr(self.run_max), ptr(self.run_sum), ptr(self.part), self.n_part
        d.out_nodes, d.out_len, d.out_score = self.out_nodes.data_ptr(), self.out_len.data_ptr(), self.out_score.data_ptr()
        d.stride = self.stride

    def _set(self, name, arr):
        arr = np.asarray(arr, dtype=np.int32).reshape(-1)
        assert arr.size <= self.isize[name], (name, arr.size, self.isize[name])
        o = self.ioff[name]
        self.host_np[o:o + arr.size] = arr

    def fits(self, need):
        return all(self.caps[k] >= v for k, v in need.items())


class DecodeEngine:
    MAX_PLANS = 4                  # plans kept regardless of their size (two or three are in flight at a time)
    MAX_PLANS_SMALL = 48           # ... and as many more as fit PLAN_BYTES of device memory
    PLAN_BYTES = 4 << 30

    def __init__(self, dev_model):
        self.m = dev_model
        self.torch = dev_model.torch
        self.device = dev_model.device
        self.last_timing = None
        self.last_fix_timing = None     # dynamic decoder under timing: (vocab fix, path fix) seconds per frame
        self.last_state = None
        self.recorder = None            # optional model.KernelRecorder (bench.py): forces eager launches
        self.last_n_live = None
        self.use_graph = self.device.type == "cuda" and os.environ.get("JLM_GRAPH", "0") == "1"
        self.use_side = os.environ.get("JLM_SIDE", "1") != "0"       # edge logits beside the normaliser
        self.graph_full = os.environ.get("JLM_GRAPH_FULL", "0") == "1"  # replay for the full-vocabulary decode too
        # the frame loop as ONE native call (jlm_decode_frames) instead of ~170 ctypes calls per batch;
        # JLM_NATIVE_LOOP=0 (and timing / recorder runs) enqueue the launches one by one from Python
        self.native_loop = os.environ.get("JLM_NATIVE_LOOP", "1") != "0"
        self.plans = []
        self._side = {}            # side stream of each launch stream (edge logits beside the normaliser)
        # Consecutive batches go to alternating HIP streams: the latency-bound kernels of batch i+1
        # (beam step, LSTM step, T projection: two thirds of the launches, a third of the time, most CUs
        # idle) fill in beside the vocabulary kernel of batch i.  JLM_STREAMS=1 keeps one stream.
        self.n_streams = max(1, int(os.environ.get("JLM_STREAMS", "2")))
        self._streams = []
        self._rr = 0

    # ------------------------------------------------------------------ plans
    def _plan_for(self, kind, vmode, lat, need, size_class=()):
        # Buffers are sized for the frame count rounded up to 8 so that ragged inputs (every chunk has its own longest
        # sentence) share plans instead of allocating ~1 GB of state rows and pinned staging per distinct length; the
        # frame loop runs lat.n_frames.  A captured graph bakes the loop in, so with replay the exact count is the key.
        fkey = lat.n_frames if self.use_graph else _round_up(lat.n_frames, 8)
        key = (kind, vmode, lat.n_sent, lat.beam, fkey, size_class)
        for i, p in enumerate(self.plans):
            if p.key == key and p.fits(need) and not p.busy:
                self.plans.append(self.plans.pop(i))
                return p
        caps = {k: _round_up(int(v * 1.25) + 64, 1024) for k, v in need.items()}
        caps["cands"] = max(_round_up(need["cands"], 256), 1024)
        self.plans = [p for p in self.plans if p.busy or p.key != key or p.fits(need)]
        p = _Plan(self, key, caps)
        # Least recently used idle plans go when the set outgrows its budget: a count for the big batch plans (hundreds
        # of MB each), bytes for the small ones -- sentence-at-a-time callers (eval.py) meet a new (length bucket, list
        # size) shape every few calls, and re-allocating a plan costs more than the decode it serves.
        idle = [q for q in self.plans if not q.busy]
        total = sum(q.nbytes for q in self.plans) + p.nbytes
        while idle and (len(self.plans) >= self.MAX_PLANS_SMALL or
                        (total > self.PLAN_BYTES and len(self.plans) >= self.MAX_PLANS)):
            q = idle.pop(0)
            self.plans.remove(q)
            total -= q.nbytes
        self.plans.append(p)
        return p

    # ---------------------------------------------------------------- enqueue
    def _enqueue(self, p, timing):
        """The whole launch sequence of one batch (no host synchronisation inside)."""
        torch, m, L = self.torch, self.m, _lib.lib()
        kind, vmode, B, beam = p.key[:4]
        F = p.latS.n_frames                 # this batch's frames (<= the plan's capacity p.F)
        rmax = p.rmax
        dynamic = kind == "dynamic"
        self_norm = m.self_norm
        mode = 1 if self_norm else (2 if dynamic else 0)
        cuda = self.device.type == "cuda"
        main = torch.cuda.current_stream() if cuda else None
        st = main.cuda_stream if cuda else 0
        side = None
        # with two batches in flight a side stream per batch needs more hardware queues than ROCm's default
        # (jlm_amd/__init__.py); without them the edge logits run on the batch's own stream (-4 %, not -30 %)
        from . import hw_queues_ok
        side_ok = self.use_side and (self.n_streams < 2 or hw_queues_ok() or os.environ.get("JLM_SIDE") == "1")
        if cuda and side_ok and self.recorder is None and not timing:
            side = self._side.get(st)
            if side is None:
                side = self._side[st] = torch.cuda.Stream()
        native = getattr(L, "jlm_decode_frames", None) if self.native_loop else None
        if native is not None and self.recorder is None and not timing:
            p.cnt.zero_()
            p.n_live.zero_()
            d = p.desc
            d.vs_max, d.di_max, d.dd_max = p.max_words["vs"], p.max_words["di"], p.max_words["dd"]
            rc = native(m.decode_desc(), d, p.latS, p.stS, st, side.cuda_stream if side is not None else None)
            if rc != -2:
                _lib.check(rc, "jlm_decode_frames")
                return []
        ip = p.ip
        H, ldt = m.H, m.ldt
        hp, cp, Tp = p.h.data_ptr(), p.c.data_ptr(), p.T.data_ptr()
        bpp, wordp, cntp = p.bp.data_ptr(), p.word.data_ptr(), p.cnt.data_ptr()
        livep, nlivep, lsep = p.live.data_ptr(), p.n_live.data_ptr(), p.lse.data_ptr()
        b2p = m.b2.data_ptr()
        segs, nsegs = m.seg_array, m.n_segs
        cands = p.caps["cands"]
        p.cnt.zero_()
        p.n_live.zero_()
        ev = []
        self._fix_ev = []
        join = None
        pending_parts = 0
        wl_split = getattr(m, "split_array", None) is not None and nsegs == 1 and beam <= 32

        def wl_lse(g0, cidx, words, off, base, merge, n_groups, what, max_words, idx=None):
            """jlm_wordlist_lse; on the split rows (deep gather ring, 128 KB of LDS per workgroup) when the
            model has them and the lists are long enough to be bound by the gather -- the short delta
            lists of the incremental decoder (tens of words, thousands of groups) are bound by how many
            workgroups fit a CU and stay on the 33-KB f32 kernel"""
            if wl_split and 128 <= max_words <= 4064:
                r = L.jlm_wordlist_lse_split(m.split_array, m.split_t_scale[0], m.split_descale[0], b2p, Tp, ldt, g0, cntp,
                                             cidx, words, off, idx or ip["sidx"], base, max_words, p.run_max.data_ptr(),
                                             p.run_sum.data_ptr(), lsep, merge, beam, n_groups, st)
                if r != -2:
                    _lib.check(r, "jlm_wordlist_lse_split(%s)" % what)
                    return
            _lib.check(L.jlm_wordlist_lse(segs, nsegs, b2p, Tp, ldt, g0, cntp, cidx, words, off, idx or ip["sidx"], base,
                                          p.run_max.data_ptr(), p.run_sum.data_ptr(), lsep, merge, beam, n_groups, st),
                       "jlm_wordlist_lse(%s)" % what)

        for f in range(F):
            if join is not None:
                main.wait_event(join)
                join = None
            if timing and dynamic:
                f0, f1, f2 = (_Stamp(torch, self.device) for _ in range(3))
                f0.record()
            if dynamic and not self_norm and f >= 2:
                # K11: older frames learn the words that first appear at frame f
                r = -2
                if wl_split and p.max_words["dd"] <= 128:
                    # one workgroup per sentence: its delta list is gathered once for all of its older rows
                    r = L.jlm_wordlist_merge_split(m.split_array, m.split_t_scale[0], m.split_descale[0], b2p, Tp, ldt, cntp,
                                                   B, beam, f - 1, ip["dd_words"], ip["dd_off"], f * B, p.max_words["dd"],
                                                   p.run_max.data_ptr(), p.run_sum.data_ptr(), lsep, st)
                    if r != -2:
                        _lib.check(r, "jlm_wordlist_merge_split")
                if r == -2:
                    wl_lse(ip["g0"], ip["cidx"], ip["dd_words"], ip["dd_off"], f * B, 1, (f - 1) * B, "merge",
                           p.max_words["dd"])
            # the full-vocabulary normaliser of frame f-1 left partial slices: beam_step folds them itself
            p.stS.lse_part = p.part.data_ptr() if pending_parts else None
            p.stS.ld_part, p.stS.n_parts = rmax, pending_parts
            if timing and dynamic:
                f1.record()
            _lib.check(L.jlm_beam_step(p.latS, p.stS, f, mode, cands, st), "jlm_beam_step")
            if timing and dynamic:
                f2.record()
                self._fix_ev.append((f0, f1, f2))
            pending_parts = 0
            if f == F - 1:
                break
            rows = livep + 4 * f * rmax
            ndev = nlivep + 4 * f
            if timing:
                e0, e1, e2 = (_Stamp(torch, self.device) for _ in range(3))
                e0.record()
            m.lstm_step(hp, cp, H, hp, cp, rows, bpp, wordp, rmax, ndev, st, self.recorder, split=m.split_lstm)
            if timing:
                e1.record()
            m.project_T(hp, H, Tp, rows, rmax, ndev, st, split=m.split_lstm)
            cell = 4 * f * B
            est = st
            if side is not None:       # edge logits need only T: run them beside the normaliser
                fork = torch.cuda.Event()
                fork.record(main)
                side.wait_event(fork)
                est = side.cuda_stream
            _lib.check(L.jlm_edge_logits(segs, nsegs, b2p, Tp, ldt, ip["g0"] + cell, cntp, ip["cidx"] + cell,
                                         ip["sg_word"], ip["sg_off"], ip["sidx"], f * B, ip["sg_node"],
                                         p.edge.data_ptr(), beam, B, est), "jlm_edge_logits")
            if side is not None:
                join = torch.cuda.Event()
                join.record(side)
            if not self_norm:
                if dynamic:
                    wl_lse(ip["g0"] + cell, ip["cidx"] + cell, ip["di_words"], ip["di_off"], 2 * f * B, 0, B, "init",
                           p.max_words["di"], ip["sidx2"])
                elif vmode == "select":
                    wl_lse(ip["g0"] + cell, ip["cidx"] + cell, ip["vs_words"], ip["vs_off"], 0, 0, B, "vocab_select",
                           p.max_words["vs"])
                else:
                    # frame 0 has one row per sentence: telling the kernel so lets it cut the vocabulary into
                    # more ranges (one resident round of workgroups) instead of leaving nine tenths of the CUs idle
                    bound = B if f == 0 else rmax
                    pending_parts = m.full_vocab_lse(Tp, rows, p.part.data_ptr(), rmax, p.n_part, lsep, bound, ndev, st,
                                                      self.recorder, combine=False)
            if timing:
                e2.record()
                ev.append((e0, e1, e2))
        if join is not None:
            main.wait_event(join)
        _lib.check(L.jlm_backtrace(p.latS, p.stS, p.out_nodes.data_ptr(), p.out_len.data_ptr(), p.out_score.data_ptr(),
                                   p.stride, st), "jlm_backtrace")
        return ev

    # ----------------------------------------------------------------- decode
    def submit(self, lat, kind="static", vocab=None, dyn_lists=None, topN=10, timing=False):
        """Enqueue one batch (upload, launch sequence, asynchronous read-back) and return a
        ticket for :meth:`collect`.  Nothing here waits for the GPU.  Successive calls use
        alternating streams (see __init__); every ticket owns its plan's buffers until collected."""
        torch = self.torch
        if self.device.type != "cuda" or self.n_streams < 2 or timing or self.recorder is not None:
            return self._submit(lat, kind, vocab, dyn_lists, topN, timing)
        if not self._streams:
            self._streams = [torch.cuda.Stream() for _ in range(self.n_streams)]
        strm = self._streams[self._rr]
        self._rr = (self._rr + 1) % self.n_streams
        cur = torch.cuda.current_stream()
        if not cur.query():                                  # after whatever the caller queued (weight uploads, ...);
            strm.wait_stream(cur)                            # nothing pending there in the steady state: no event, no wait
        with torch.cuda.stream(strm):
            return self._submit(lat, kind, vocab, dyn_lists, topN, timing)

    def _submit(self, lat, kind, vocab, dyn_lists, topN, timing):
        torch = self.torch
        dynamic = kind == "dynamic"
        vmode = "dynamic" if dynamic else ("select" if vocab is not None else "full")
        need = dict(nodes=lat.n_nodes, vs=len(vocab[0]) if vocab is not None else 0,
                    di=len(dyn_lists[0]) if dynamic else 0, dd=len(dyn_lists[2]) if dynamic else 0, cands=lat.max_cands)

        # longest word list per kind of call (selected vocabulary / frame-initial / frame-delta lists).  Which
        # kernel a call uses depends on it, and a captured graph bakes that choice in, so the size class is
        # part of the plan's identity.
        def longest(offs):
            o = np.asarray(offs)
            return int(np.diff(o).max()) if o.size > 1 else 0
        max_words = dict(vs=longest(vocab[1]) if vocab is not None else 0,
                         di=int((dyn_lists[1][1::2] - dyn_lists[1][0::2]).max()) if dynamic else 0,
                         dd=longest(dyn_lists[3]) if dynamic else 0)
        size_class = tuple((v < 128, v <= 128, v <= 4064) for v in (max_words["vs"], max_words["di"], max_words["dd"]))
        p = self._plan_for(kind, vmode, lat, need, size_class)
        p.max_words = max_words
        p.busy = True
        assert lat.n_frames <= p.F
        p.latS.n_frames = lat.n_frames
        p._set("sent_len", lat.sent_len)
        p._set("end_off", lat.end_off)
        p._set("node_start", lat.node_start)
        p._set("node_word", lat.node_word)
        p._set("sg_off", lat.sg_off)
        p._set("sg_word", lat.sg_word)
        p._set("sg_node", lat.sg_node)
        if vocab is not None:
            p._set("vs_words", vocab[0])
            p._set("vs_off", vocab[1])
        if dynamic:
            p._set("di_words", dyn_lists[0])
            p._set("di_off", dyn_lists[1])
            p._set("dd_words", dyn_lists[2])
            p._set("dd_off", dyn_lists[3])
        p.dev_ints.copy_(p.host_ints, non_blocking=True)
        p.uses = getattr(p, "uses", 0) + 1
        # replay pays where the step is bound by the Python enqueue loop: the word-list decodes (vocab_select,
        # incremental) and self-normalised models, whose kernels are all short.  The full-vocabulary decode is
        # device bound and runs better eagerly on its two streams (3.15 vs 3.60 ms, tools/ab_engine.py).
        graph_ok = self.use_graph and (p.key[1] != "full" or self.m.self_norm or self.graph_full)
        eager = (not graph_ok) or timing or (self.recorder is not None) or (p.graph is None and p.uses <= 2)
        ev = []
        if eager:
            ev = self._enqueue(p, timing)
            p.warm = True
        else:
            if p.graph is None:                    # third use of this plan: worth a capture
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._enqueue(p, False)
                p.graph = g
            p.graph.replay()
        p.h_nodes.copy_(p.out_nodes, non_blocking=True)
        p.h_len.copy_(p.out_len, non_blocking=True)
        p.h_score.copy_(p.out_score, non_blocking=True)
        if self.recorder is not None:
            p.h_nlive.copy_(p.n_live, non_blocking=True)
        done = None
        if self.device.type == "cuda":
            done = torch.cuda.Event()
            done.record()
        return (p, lat, topN, ev, timing, done, list(getattr(self, "_fix_ev", [])) if timing else [])

    def collect(self, ticket):
        """Wait for a submitted batch and build its n-best lists."""
        p, lat, topN, ev, timing, done, fix_ev = ticket
        if done is not None:
            done.synchronize()
        if self.recorder is not None:
            self.last_n_live = p.h_nlive.numpy()[:p.latS.n_frames].copy()
        if timing:
            self.last_timing = [(a.seconds_to(b), b.seconds_to(c2)) for a, b, c2 in ev]
            self.last_fix_timing = [(a.seconds_to(b), b.seconds_to(c2)) for a, b, c2 in fix_ev]
        self.last_state = p
        out = self._read_out(lat, p.h_nodes.numpy(), p.h_len.numpy(), p.h_score.numpy(), topN)
        p.busy = False
        return out

    def decode(self, lat, kind="static", vocab=None, dyn_lists=None, topN=10, timing=False, keep_state=False):
        """lat: BatchLattice.  kind: 'static' | 'dynamic'.
        vocab: (words, off) CSR of per-sentence selected vocabularies (static
        vocab_select) or None for the full vocabulary.
        dyn_lists: (seq_words, init_range, delta_words, delta_off) for 'dynamic' (BatchLattice.dynamic_vocab).
        -> list (per sentence) of [(neg_log_prob, [word, ...])][:topN]"""
        if lat.n_sent == 0:
            return []
        return self.collect(self.submit(lat, kind, vocab, dyn_lists, topN, timing))

    @staticmethod
    def _read_out(lat, nodes_h, len_h, score_h, topN):
        """n-best lists of the batch from the back-traces: [(score, [word, ...])] per sentence, best first.
        Built by the C extension jlm_amd._readout (csrc/jlm_readout.c: 0.4 ms for 2 560 paths) when it is
        there, else by the numpy implementation below (1.8 ms) -- same result, tests/test_readout.py."""
        ext = _readout_ext()
        if ext is None:
            return DecodeEngine._read_out_py(lat, nodes_h, len_h, score_h, topN)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        nodes_h = i32(nodes_h)
        return ext.nbest(nodes_h, i32(len_h), np.ascontiguousarray(score_h, dtype=np.float64), i32(lat.node_lex),
                         i32(lat.node_sent), i32(lat.node_start), lat.builder.lex_list, lat.texts, lat.n_sent, lat.beam,
                         int(topN), nodes_h.shape[-1])

    @staticmethod
    def _read_out_py(lat, nodes_h, len_h, score_h, topN):
        """numpy form of _read_out: one gather of all paths' node ids, one node -> string lookup, two tolist() calls."""
        B, beam = lat.n_sent, lat.beam
        R = min(beam, topN)
        ln = np.asarray(len_h).reshape(B, beam)[:, :R]
        # ranks are filled from 0; a sentence's list ends at its first empty rank
        valid = np.logical_and.accumulate(ln > 0, axis=1)
        nrank = valid.sum(axis=1)
        rows = (np.arange(B)[:, None] * beam + np.arange(R)[None, :])[valid]          # path rows, sentence-major
        k = ln[valid].astype(np.int64) - 1                                             # words per path (root dropped)
        kmax = int(k.max()) if k.size else 0
        flat_words = []
        if kmax > 0:
            cols = np.arange(kmax)[None, :]
            keep = cols < k[:, None]
            src = np.where(keep, k[:, None] - 1 - cols, 0)                             # reversed: last word first in the trace
            ids = np.take_along_axis(np.asarray(nodes_h)[rows][:, :max(kmax, 1)], src, axis=1)[keep]
            flat_words = lat.words_of(ids).tolist()
        ends = np.cumsum(k).tolist()
        scores = np.asarray(score_h)[rows].tolist()
        out, p, a = [], 0, 0
        for s in range(B):
            lst = []
            for _ in range(int(nrank[s])):
                e = ends[p]
                lst.append((scores[p], flat_words[a:e]))
                a = e
                p += 1
            out.append(lst)
        return out
