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

Upload, the whole launch sequence and the read-back of a batch are ONE custom op (torch.ops.jlm.decode_batch ->
jlm_decode_frames, csrc/jlm_decode.hip): no host work between the frames, none of it under the interpreter lock.
"""
import os
import threading

import numpy as np

from . import _lib
from . import ops


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
    """Device buffers of one decode shape + their torch.classes.jlm.Plan (the C structs of include/jlm_hip.h)."""

    # the four node-sized arrays LAST: lattices built into page-locked blocks (lattice.StagingPool) copy them to the device
    # straight from there, and only the head [0, head_end) goes through this plan's staging block
    BIG_ARRAYS = ("node_start", "node_word", "sg_word", "sg_node")
    INT_ARRAYS = ("sent_len", "end_off", "sg_off", "g0", "cidx", "sidx", "sidx2", "vs_words", "vs_off", "di_words", "di_off", "dd_words",
                  "dd_off")

    def __init__(self, eng, key, caps):
        torch, m, dev = eng.torch, eng.m, eng.device
        self.key, self.caps = key, dict(caps)
        kind, vmode, B, beam, F = key[:5]
        perm = bool(key[6]) if len(key) > 6 else False       # reference-compatibility lists (dynamic x segmented, see submit)
        self.B, self.beam, self.F = B, beam, F
        rmax, ncell = B * beam, F * B
        G = F * rmax
        # the LSTM-step kernels address state rows as 16-byte records through a 31-bit index (csrc/jlm_gate.hip): H / 4 records per row
        if G * max(m.H // 4, 1) >= 0x7ffffff0:
            raise ValueError("a decode plan of %d state rows of %d units is beyond the LSTM-step kernels' addressing (rows x H / 4 < 2^31): "
                             "decode fewer sentences per batch (Decoder.max_batch)" % (G, m.H))
        self.rmax, self.G, self.ncell = rmax, G, ncell
        sizes = dict(sent_len=B, end_off=ncell + 1, node_start=caps["nodes"], node_word=caps["nodes"], sg_off=ncell + 1,
                     sg_word=caps["nodes"], sg_node=caps["nodes"], g0=ncell, cidx=ncell, sidx=ncell, sidx2=ncell,
                     vs_words=caps["vs"], vs_off=B + 1, di_words=caps["di"], di_off=2 * ncell, dd_words=caps["dd"],
                     dd_off=ncell + 1)
        names = self.INT_ARRAYS + (("di_wwords", "sg_wword") if perm else ()) + self.BIG_ARRAYS
        if perm:
            sizes.update(di_wwords=caps["di"], sg_wword=caps["nodes"])
        off, self.ioff = 0, {}
        for n in names:
            if n == self.BIG_ARRAYS[0]:
                self.head_end = off
            self.ioff[n] = off
            off += _round_up(max(sizes[n], 1), 4)
        self.isize = {n: sizes[n] for n in names}
        self.host_ints = torch.zeros(off + 4, dtype=torch.int32)
        if dev.type == "cuda":
            self.host_ints = self.host_ints.pin_memory()
        self.host_np = self.host_ints.numpy()
        self.dev_ints = torch.zeros(off + 4, dtype=torch.int32, device=dev)
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
        # untied models: T is the state itself -- the same buffer on the f32 path; with split state rows, their plain f32 copy
        self.T = (e((G, H), f32) if m.split_lstm else self.h) if m.mode == "untied" else e((G, ldt), f32)
        self.run_max = self.run_sum = self.part = None
        self.n_part = 0
        if not m.self_norm:
            if vmode != "full":
                self.run_max, self.run_sum = e(G, f32), e(G, f64)
            else:
                self.n_part = max(m.n_vocab_tiles, 1)
                self.part = e((self.n_part, rmax, 2), f32)
        # packed rows of the frame being stepped, for the segments of the normaliser on mixed rows (jlm_pack_t_mixed)
        self.Tm = None
        if self.part is not None and getattr(m, "ld_tm", 0):
            self.Tm = torch.zeros(((rmax + 31) // 32 * 32, m.ld_tm), device=dev, dtype=f32)      # whole 32-row blocks (granule-major, jlm_hip.h)
        self.stride = F + 1
        self.out_nodes = e((rmax, self.stride), i32)
        self.out_len = torch.zeros(rmax + 1, device=dev, dtype=i32)       # [rmax]: the batch's flag word (jlm_beam_state.flags, ABI 11)
        self.out_score = e(rmax, f64)
        pin = (lambda t: t.pin_memory()) if dev.type == "cuda" else (lambda t: t)
        self.h_nodes = pin(torch.empty((rmax, self.stride), dtype=i32))
        self.h_len = pin(torch.zeros(rmax + 1, dtype=i32))
        self.h_score = pin(torch.empty(rmax, dtype=f64))
        self.h_nlive = pin(torch.empty(F, dtype=i32))
        self.busy = False
        self.nbytes = sum(t.numel() * t.element_size() for t in vars(self).values() if isinstance(t, torch.Tensor) and t.device == dev)
        # torch.classes.jlm.Plan: the same buffers as jlm_lattice / jlm_beam_state / jlm_decode_plan for the frame-loop op
        tensors = dict(ints=self.dev_ints, score=self.score, lse=self.lse, bp=self.bp, node=self.node, word=self.word,
                       cnt=self.cnt, live=self.live, n_live=self.n_live, live_base=self.live_base, edge=self.edge, h=self.h,
                       c=self.c, T=self.T, out_nodes=self.out_nodes, out_len=self.out_len, out_score=self.out_score)
        for name in ("ysum", "run_max", "run_sum", "part", "Tm"):
            if getattr(self, name) is not None:
                tensors[name] = getattr(self, name)
        ints = dict(n_sent=B, beam=beam, frames=F, kind=2 if dynamic else (1 if vmode == "select" else 0),
                    max_cands=caps["cands"], max_parts=self.n_part, stride=self.stride)
        if self.Tm is not None:
            ints["ld_tm"] = m.ld_tm
        ints.update({"off_" + n: o for n, o in self.ioff.items()})
        self.obj = ops.backend().Plan(tensors, ints)

    def _set(self, name, arr):
        arr = np.asarray(arr, dtype=np.int32).reshape(-1)
        assert arr.size <= self.isize[name], (name, arr.size, self.isize[name])
        o = self.ioff[name]
        self.host_np[o:o + arr.size] = arr

    def fits(self, need):
        return all(self.caps[k] >= v for k, v in need.items())


class DecodeEngine:
    _stream_pool = {}              # device index -> launch streams shared by all engines of the process
    MAX_PLANS = 6                  # plans kept regardless of their size (up to four are in flight at a time, one is being read out)
    MAX_PLANS_SMALL = 48           # ... and as many more as fit PLAN_BYTES of device memory
    PLAN_BYTES = 4 << 30

    def __init__(self, dev_model):
        self.m = dev_model
        self.torch = dev_model.torch
        self.device = dev_model.device
        self.last_timing = None         # timed decode: (lstm, softmax) seconds per stepped frame (perf_log_* of decoder.py:206-218)
        self.last_fix_timing = None     # ... incremental decoder: (vocab fix, lattice path fix) seconds per frame
        self.last_kernel_ms = None      # ... {"gate_gemm": [...], "vocab_lse": [...]} milliseconds per launch (bench.py)
        self.last_state = None
        self.last_n_live = None
        self.keep_n_live = False        # bench.py: read back every frame's live-row count with a timed decode
        self.use_side = os.environ.get("JLM_SIDE", "1") != "0"       # edge logits beside the normaliser
        self.blocking_sync = False       # (JLM_BLOCKING_SYNC, rounds 2-4: the collecting thread asleep instead of spinning -- no less CPU, 1-3 % slower; removed)
        from .lattice import StagingPool
        self.staging_pool = None if os.environ.get("JLM_PINNED_LATTICE", "1") == "0" else StagingPool(self.torch, self.device.type == "cuda")
        self.plans = []
        # Consecutive batches go to alternating HIP streams: the latency-bound kernels of batch i+1
        # (beam step, LSTM step, T projection: two thirds of the launches, a third of the time, most CUs
        # idle) fill in beside the vocabulary kernel of batch i.  JLM_STREAMS=1 keeps one stream.
        # (round 3: FOUR batches in flight -- 1.95-1.98 against 2.04-2.06 ms per step once the mixed-row normaliser left more of
        # a frame to the latency-bound kernels; five: 2.36 -- Decoder.depth_for keeps three for batches above 8 192 rows)
        # THREE batches in flight on three streams, edge logits on the batch's own stream: 2.14-2.22 ms per step against
        # 2.32-2.36 with two streams + a side stream each (tools/ab_streams.py).  A frame of one batch is a chain of dependent
        # launches (beam step, LSTM step, T projection, vocabulary kernel: every one of them on the critical path of the step,
        # tools/probes/skip_kernel.sh) and the chains of three batches fill each other's gaps better than those of two; side
        # streams on top (six streams, fork / join events across them) cost more than the overlap buys: 3.3 ms.
        # four streams need more than ROCm's default of four hardware queues (one is the null stream's): there they run 2.31 ms per
        # step against 2.06 with three (jlm_amd/__init__.py sets GPU_MAX_HW_QUEUES=8 when it still can)
        from . import hw_queues_ok as _hwq
        self.n_streams = max(1, int(os.environ.get("JLM_STREAMS", "4" if _hwq() else "3")))
        # CU share of the vocabulary kernel beside other batches in flight: none since round 4 (the step is flat in it with four batches
        # in flight, profiles/r04_b_share_sweep.txt, and a share below 100 made a score's last bits depend on how many chunks the call
        # had); the JLM_LSE_SHARE knob left in round 5, the frame-loop op keeps its (unused) argument.
        self.lse_share_pct = 100 if self.n_streams >= 2 else 0
        self._streams = []
        self._rr = 0
        # `pipelined`: set by the caller around a pipelined sequence of submits (Decoder.decode_batch).  Per THREAD: two threads inside
        # decode_batch on the same decoder must not clear each other's flag (it decides the CU share, i.e. the vocabulary kernel's
        # column cuts, of every batch the thread submits meanwhile)
        self._tls = threading.local()
        self._submit_lock = threading.RLock()

    @property
    def pipelined(self):
        return getattr(self._tls, "pipelined", False)

    @pipelined.setter
    def pipelined(self, v):
        self._tls.pipelined = bool(v)

    def _ctx(self):
        return self.m._ctx()

    # ------------------------------------------------------------------ plans
    def _plan_for(self, kind, vmode, lat, need, size_class=(), perm=False):
        # Buffers are sized for the frame count rounded up to 8 so that ragged inputs (every chunk has its own longest
        # sentence) share plans instead of allocating ~1 GB of state rows and pinned staging per distinct length; the
        # frame loop runs lat.n_frames.
        fkey = _round_up(lat.n_frames, 8)
        key = (kind, vmode, lat.n_sent, lat.beam, fkey, size_class, bool(perm))
        for i, p in enumerate(self.plans):
            if p.key == key and p.fits(need) and not p.busy:
                self.plans.append(self.plans.pop(i))
                return p
        caps = {k: _round_up(int(v * 1.25) + 64, 1024) for k, v in need.items()}
        # (at least 1 024 where the beam step's LDS takes that many: wide beams over many frames take fewer -- a batch that passed
        #  Decoder._check_cells must also pass the launcher's LDS formula)
        lim = int(ops.backend().beam_step_max_cands(int(lat.beam), fkey, 2 if kind == "dynamic" else (1 if self.m.self_norm else 0)))
        caps["cands"] = max(_round_up(need["cands"], 256), min(1024, lim) if lim > 0 else 1024)
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

    # ----------------------------------------------------------------- decode
    def submit(self, lat, kind="static", vocab=None, dyn_lists=None, topN=10, timing=False):
        """Enqueue one batch (upload, the frame-loop op, asynchronous read-back) and return a
        ticket for :meth:`collect`.  Nothing here waits for the GPU.  Successive calls use
        alternating streams (see __init__); every ticket owns its plan's buffers until collected.
        timing=True: HIP events around the kernel groups of every frame (one stream, no side stream);
        timing="inflight": the same events recorded on the batch's own stream of the PIPELINED submit -- a kernel's time
        then includes what it loses to the other batches in flight (bench.py: `frac_in_pipeline`)."""
        torch = self.torch
        with self._submit_lock, self._ctx():          # (the plan list and the stream rotation are shared by every submitting thread)
            if self.device.type != "cuda" or self.n_streams < 2 or (timing and timing != "inflight"):
                return self._submit(lat, kind, vocab, dyn_lists, topN, timing)
            if len(self._streams) != self.n_streams:
                # the launch streams are shared by every engine on the device: each also gets a side stream inside the op, and
                # streams beyond the GPU's hardware queues (GPU_MAX_HW_QUEUES, jlm_amd/__init__.py) serialise one another --
                # a second Decoder in the process (bench.py's configs[4] leg) must not double them
                pool = DecodeEngine._stream_pool.setdefault(self.device.index or 0, [])
                while len(pool) < self.n_streams:
                    pool.append(torch.cuda.Stream(self.device))
                self._streams = pool[:self.n_streams]
            strm = self._streams[self._rr]
            self._rr = (self._rr + 1) % self.n_streams
            cur = torch.cuda.current_stream(self.device)
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

        # longest word list per kind of call (selected vocabulary / frame-initial / frame-delta lists): which
        # kernel a call uses depends on it
        def longest(offs):
            o = np.asarray(offs)
            return int(np.diff(o).max()) if o.size > 1 else 0
        max_words = dict(vs=longest(vocab[1]) if vocab is not None else 0,
                         di=int((dyn_lists[1][1::2] - dyn_lists[1][0::2]).max()) if dynamic else 0,
                         dd=longest(dyn_lists[3]) if dynamic else 0)
        size_class = tuple((v < 128, v <= 128, v <= 4064) for v in (max_words["vs"], max_words["di"], max_words["dd"]))
        # dyn_lists[4:6] (optional): the weight-row words of the reference-compatibility mode (DynamicDecoder.compat_quirks on
        # D-softmax / D-softmax* models): per init-list position and per lattice edge (include/jlm_hip.h, jlm_decode_plan)
        perm = dynamic and len(dyn_lists) >= 6 and dyn_lists[4] is not None
        p = self._plan_for(kind, vmode, lat, need, size_class, perm)
        p.busy = True
        try:
            done = self._enqueue(p, lat, vocab, dyn_lists, dynamic, perm, max_words, topN, timing)
        except BaseException:
            # nothing may keep the plan: launches already enqueued finish first, then its buffers are free again
            if self.device.type == "cuda":
                try:
                    torch.cuda.current_stream(self.device).synchronize()
                except Exception:
                    pass
            p.busy = False
            raise
        return (p, lat, topN, timing, done)

    def _enqueue(self, p, lat, vocab, dyn_lists, dynamic, perm, max_words, topN, timing):
        torch = self.torch
        assert lat.n_frames <= p.F
        p._set("sent_len", lat.sent_len)
        p._set("end_off", lat.end_off)
        p._set("sg_off", lat.sg_off)
        blk = getattr(lat, "_block", None)
        if blk is None:
            for name in p.BIG_ARRAYS:
                p._set(name, getattr(lat, name))
        if vocab is not None:
            p._set("vs_words", vocab[0])
            p._set("vs_off", vocab[1])
        if dynamic:
            p._set("di_words", dyn_lists[0])
            p._set("di_off", dyn_lists[1])
            p._set("dd_words", dyn_lists[2])
            p._set("dd_off", dyn_lists[3])
            if perm:
                p._set("di_wwords", dyn_lists[4])
                p._set("sg_wword", dyn_lists[5])
        be = ops.backend()
        from . import hw_queues_ok
        side = self.use_side and self.device.type == "cuda" and (
            self.n_streams < 2 or (self.n_streams == 2 and hw_queues_ok()) or os.environ.get("JLM_SIDE") == "1")
        share = self.lse_share_pct if ((not timing or timing == "inflight") and self.pipelined and self.lse_share_pct < 100) else 0
        if self.device.type == "cuda" and hasattr(be, "decode_batch"):
            # Round 5: upload, counters, frame loop and read-back of the batch as ONE op (csrc/jlm_torch_ops.cpp decode_batch) -- the
            # dozen torch calls below cost the calling thread 0.3-0.6 ms of interpreter time per 256-sentence chunk
            # (profiles/r05_i_host_profile.txt), the op runs without the interpreter lock
            src, dst, cnt = [], [], []
            if blk is not None:
                for name in p.BIG_ARRAYS:
                    n = int(getattr(lat, name).shape[0])
                    assert n <= p.isize[name], (name, n, p.isize[name])
                    src.append(int(lat.block_off[name])); dst.append(int(p.ioff[name])); cnt.append(n)
            rc = be.decode_batch(self.m.decode_model(), p.obj, p.host_ints, int(p.head_end) if blk is not None else 0,
                                 blk.tensor if blk is not None else None, src, dst, cnt, p.h_nodes, p.h_len, p.h_score,
                                 p.h_nlive if (timing and self.keep_n_live) else None, lat.n_frames, max_words["vs"], max_words["di"],
                                 max_words["dd"], bool(side), bool(timing), int(share))
            if rc != 0:
                raise _lib.JlmHipError("jlm.decode_batch: the model is outside the shapes the frame loop covers (code %d)" % rc)
            done = torch.cuda.Event(blocking=self.blocking_sync)
            done.record()
            return done
        if blk is None:
            p.dev_ints.copy_(p.host_ints, non_blocking=True)
        else:
            # the head from the plan's staging block, the node arrays straight from the lattice's page-locked block
            p.dev_ints[:p.head_end].copy_(p.host_ints[:p.head_end], non_blocking=True)
            for name in p.BIG_ARRAYS:
                n, o, d = int(getattr(lat, name).shape[0]), lat.block_off[name], p.ioff[name]
                assert n <= p.isize[name], (name, n, p.isize[name])
                if n:
                    p.dev_ints[d:d + n].copy_(blk.tensor[o:o + n], non_blocking=True)
        p.cnt.zero_()
        p.n_live.zero_()
        p.out_len[-1:].zero_()          # the flag word
        # (side stream for the edge logits -- `side` above: with one stream, and with two when the hardware queues are there (ROCm's default
        #  of four is not enough: jlm_amd/__init__.py); with three or more streams (the default is four) they run on the batch's own stream)
        # the whole launch sequence of the batch: ONE op, no host synchronisation inside (csrc/jlm_decode.hip)
        # another batch in flight: this batch's vocabulary kernel takes LSE_SHARE_PCT of the CUs and the other batch's
        # latency-bound kernels the rest, side by side (include/jlm_hip.h, jlm_decode_plan.lse_cu_share_pct)
        # The share is a property of the CALL (a pipelined sequence of batches: decode_batch with more than one chunk sets
        # `pipelined`), not of what happens to be in flight at this moment: the share moves the column cuts of the vocabulary
        # kernel, i.e. the grouping of its f32 partial sums, and a sentence's score must not depend on its chunk's position.
        rc = be.decode_frames(self.m.decode_model(), p.obj, lat.n_frames, max_words["vs"], max_words["di"],
                                         max_words["dd"], bool(side), bool(timing), int(share))
        if rc != 0:
            raise _lib.JlmHipError("jlm.decode_frames: the model is outside the shapes the frame loop covers (code %d)" % rc)
        p.h_nodes.copy_(p.out_nodes, non_blocking=True)
        p.h_len.copy_(p.out_len, non_blocking=True)
        p.h_score.copy_(p.out_score, non_blocking=True)
        if timing and self.keep_n_live:
            p.h_nlive.copy_(p.n_live, non_blocking=True)
        done = None
        if self.device.type == "cuda":
            # JLM_BLOCKING_SYNC=1: the collecting thread sleeps until the batch is done instead of spinning on the event
            # (measured: no less CPU per step -- the runtime's own threads spin either way -- and 1-3 % more wall time: off)
            done = torch.cuda.Event(blocking=self.blocking_sync)
            done.record()
        return done


    def collect(self, ticket):
        """Wait for a submitted batch and build its n-best lists."""
        p, lat, topN, timing, done = ticket
        with self._ctx():
            if done is not None:
                done.synchronize()
            if timing:
                t = ops.backend().frame_times(p.obj).numpy() * 1e-3          # [frames, 5] seconds
                stepped = t[:-1]                                                 # the last frame is not stepped
                self.last_timing = [(float(r[2]), float(r[3] + r[4])) for r in stepped]
                self.last_fix_timing = [(float(r[0]), float(r[1])) for r in t]
                self.last_kernel_ms = {"gate_gemm": (stepped[:, 2] * 1e3).tolist(), "vocab_lse": (stepped[:, 4] * 1e3).tolist()}
                if self.keep_n_live:
                    self.last_n_live = p.h_nlive.numpy()[:lat.n_frames].copy()
        self.last_state = p
        if int(p.h_len[-1]) != 0:
            # (ABI 11) the beam step met a log-normaliser that is not finite: a row's sum of 2^y overflowed (or vanished) in the fixed-reference
            # normaliser -- an overflowed row's hypotheses would score -inf and be pruned, leaving a plausible but wrong n-best
            p.busy = False
            raise _lib.JlmHipError("a log-normaliser is not finite: this model's logits left the range the fixed-reference normaliser covers "
                                   "(DeviceModel.mixed_calib); set JLM_MX_FIXREF=0")
        if getattr(self.m, "lse_fixed_ref", 0):
            # the normaliser ran without a running maximum (jlm_vocab_lse_mixed_fr): a row whose logits left the range the load-time probe
            # vouched for comes back as log 0 or log inf -- never a plausible score.  Loud, not silent:
            sc, ln = p.h_score.numpy(), p.h_len.numpy()[:-1]
            if not np.isfinite(sc[ln > 0]).all():
                p.busy = False
                raise _lib.JlmHipError("a path score is not finite: this model's logits left the range the fixed-reference normaliser covers "
                                       "(DeviceModel.mixed_calib); set JLM_MX_FIXREF=0")
        out = self._read_out(lat, p.h_nodes.numpy(), p.h_len.numpy()[:-1], p.h_score.numpy(), topN)
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
