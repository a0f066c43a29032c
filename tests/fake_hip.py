"""A numpy test double of libjlm_hip.so's C ABI (include/jlm_hip.h).

TEST INFRASTRUCTURE ONLY.  It lets the CPU-only suite drive the real host code
(weight packing, lattice CSR, engine launch sequence, read-out) end to end by
standing in for the device kernels, reading and writing the same buffers through
the same raw pointers.  It is never importable from the product package and the
product never falls back to it; the GPU tests run the real library.

Each function restates the CONTRACT written in the header, not the HIP code.
"""
import ctypes

import numpy as np

NEG = -3.0e38
_CT = {np.float32: ctypes.c_float, np.float64: ctypes.c_double, np.int32: ctypes.c_int32, np.float16: ctypes.c_uint16}


def _p(x):
    if x is None:
        return 0
    if isinstance(x, ctypes.c_void_p):
        return x.value or 0
    return int(x)


def view(ptr, count, dtype):
    ptr = _p(ptr)
    assert ptr != 0
    buf = (_CT[dtype] * int(count)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype)


def _n(n_max, n_dev):
    if _p(n_dev):
        return min(int(view(n_dev, 1, np.int32)[0]), n_max)
    return n_max


def _rows(rows, n):
    return view(rows, n, np.int32).astype(np.int64) if _p(rows) else np.arange(n, dtype=np.int64)


class FakeLib:
    def jlm_abi_version(self):
        return 11

    @staticmethod
    def _mx6_model(m):
        """ABI 11: the model's mixed segments are mx6 rows (s8 = 0): csrc/jlm_decode.hip jlm_model_mx6"""
        if not m.mixed_segs or not m.mixed_s8:
            return False
        for i in range(m.n_segs):
            if m.mixed_segs[i].B:
                return float(m.mixed_s8[i]) == 0.0
        return False

    def _pack_t_for(self, m):
        return self.jlm_pack_t_mixed6 if self._mx6_model(m) else self.jlm_pack_t_mixed

    def jlm_lse_probe(self, m, rowlist, prev, word, steps, rows, h, c, T, Tm, ld_tm, form, part, max_parts, stream):
        """ABI 8 (csrc/jlm_decode.hip): `steps` LSTM steps from the zero state, T of the last block, its normaliser slices"""
        if steps < 1 or rows < 1 or not m.split_lstm or not m.wt8 or m.untied or m.self_norm or not m.split_segs or not m.pmt_split:
            return -2
        for t in range(1, steps + 1):
            rc = self.jlm_lstm_step_xg(h, c, m.H, h, c, _p(rowlist) + 4 * t * rows, prev, word, m.wt8, m.xgate8, m.H, m.gate_descale,
                                       m.h_scale, None, rows, None, stream)
            if rc:
                return rc
        rl = _p(rowlist) + 4 * steps * rows
        rc = self.jlm_gemm_nt_split(h, m.H, rl, m.pmt_split, m.H, None, T, m.ldt, rl, None, m.t_descale, rows, m.n_t, m.H, None, stream)
        if rc:
            return rc
        if form == 0:
            return self.jlm_vocab_lse_split(m.split_segs, m.split_t_scale, m.split_descale, m.split_bias_col, m.n_segs, m.b2, T, m.ldt, rl,
                                            part, rows, max_parts, rows, None, stream)
        if not m.mixed_segs or not _p(Tm):
            return -2
        from jlm_amd import _lib
        idx = [i for i in range(m.n_segs) if m.mixed_segs[i].B]
        if not idx:
            return -1
        only = (_lib.Segment * len(idx))(*[m.mixed_segs[i] for i in idx])
        if self.jlm_mixed_t_stride(only, len(idx)) != ld_tm:
            return -1
        rc = self._pack_t_for(m)(only, [m.mixed_t_scale[i] for i in idx], len(idx), T, m.ldt, rl, rows, None, Tm, ld_tm, stream)
        if rc:
            return rc
        cut = bool(m.mixed_head_split) and any(m.mixed_head_split[i] > 0 for i in idx)
        if len(idx) == m.n_segs and not cut:
            return self.jlm_vocab_lse_mixed(m.mixed_segs, m.mixed_descale, m.mixed_s8, m.mixed_bias2, m.n_segs, Tm, ld_tm, part, rows,
                                            max_parts, rows, None, stream)
        return self.jlm_vocab_lse_hybrid(m.split_segs, m.split_t_scale, m.split_descale, m.split_bias_col, m.mixed_segs, m.mixed_descale,
                                         m.mixed_s8, m.mixed_head_split, m.n_segs, m.b2, T, m.ldt, Tm, ld_tm, rl, part, rows, max_parts, rows,
                                         None, stream)

    def jlm_beam_step_max_cands(self, beam, n_frames, mode):
        """the launcher's LDS formulas (csrc/jlm_beam.hip: beam_step_lds_bytes, and from round 6 the chunked kernel's -- cells above what one
        wave's LDS holds in one piece are selected chunk by chunk, the figure is what the chunk winners leave room for)"""
        if beam < 1 or beam > 1024 or n_frames < 1 or mode not in (0, 1, 2):
            return 0
        lds = lambda c: (c * 8 + (n_frames * beam * 8 if mode == 2 else 0) + ((c + 1) & ~1) * 4 + beam * 8 +
                         ((n_frames + 1) & ~1) * 4 + beam * 12 + 8)
        if lds(0) + 256 * 12 > 160 * 1024:
            return 0
        c = (160 * 1024 - lds(0)) // 12 // 256 * 256
        while c > 0 and lds(c) > 160 * 1024:
            c -= 256
        one = max(c, 0)
        cap = (160 * 1024 - lds(0)) // 12 // 2 // 256 * 256
        if one <= 0 or cap <= 0:
            return one
        base = lds(cap) + (beam + (beam & 1)) * 4
        if base >= 160 * 1024:
            return one
        total = min((160 * 1024 - base) // (beam * 16) * cap, 1 << 22) // 256 * 256
        return max(total, one)

    # ------------------------------------------------------- the frame loop (ABI 3)
    def jlm_decode_frames(self, m, p, lat, st, stream, side_stream, events=None):
        """Same call order as jlm_amd/csrc/jlm_decode.hip, over the doubles below: checks that the
        engine fills jlm_decode_model / jlm_decode_plan the way the individual calls were fed."""
        B, beam, F = lat.n_sent, lat.beam, lat.n_frames
        rmax = B * beam
        dynamic, select, full = p.kind == 2, p.kind == 1, p.kind == 0
        mode = 1 if m.self_norm else (2 if dynamic else 0)
        split = bool(m.split_segs)
        wl_split = split and m.n_segs == 1 and beam <= 64
        tile_form = full and not m.self_norm and any(m.segs[i].k > 256 for i in range(m.n_segs))
        off = lambda base, n: (base or 0) + 4 * n

        def wl_lse(g0, cidx, words, woff, idx, base, merge, n_groups, max_words):
            if wl_split and 128 <= max_words <= 4064:
                r = self.jlm_wordlist_lse_split(m.split_segs, m.split_t_scale[0], m.split_descale[0], m.b2, p.T, m.ldt, g0,
                                                st.cnt, cidx, words, woff, idx, base, max_words, p.run_max, p.run_sum,
                                                st.lse, merge, beam, n_groups, stream)
                if r != -2:
                    return r
            return self.jlm_wordlist_lse(m.segs, m.n_segs, m.b2, p.T, m.ldt, g0, st.cnt, cidx, words, woff, idx, base,
                                         p.run_max, p.run_sum, st.lse, merge, beam, n_groups, stream)

        pending = 0
        for f in range(F):
            if dynamic and not m.self_norm and f >= 2:
                r = -2
                if wl_split and p.dd_max <= 128:
                    r = self.jlm_wordlist_merge_split(m.split_segs, m.split_t_scale[0], m.split_descale[0], m.b2, p.T, m.ldt,
                                                      st.cnt, B, beam, f - 1, p.dd_words, p.dd_off, f * B, p.dd_max,
                                                      p.run_max, p.run_sum, st.lse, stream)
                if r == -2:
                    r = wl_lse(p.g0, p.cidx, p.dd_words, p.dd_off, p.sidx, f * B, 1, (f - 1) * B, p.dd_max)
                if r:
                    return r
            st.lse_part = p.part if pending else None
            st.ld_part, st.n_parts = rmax, pending
            r = self.jlm_beam_step(lat, st, f, mode, p.max_cands, stream)
            if r:
                return r
            pending = 0
            if f == F - 1:
                break
            rows, ndev = off(st.live, f * rmax), off(st.n_live, f)
            if m.split_lstm and m.wt8:
                r = self.jlm_lstm_step_xg(p.h, p.c, m.H, p.h, p.c, rows, st.bp, st.word, m.wt8, m.xgate8, m.H,
                                          m.gate_descale, m.h_scale, p.T if m.untied else None, rmax, ndev, stream)
            elif m.split_lstm:
                return -2
            else:
                r = self.jlm_lstm_step(p.h, p.c, m.H, p.h, p.c, rows, st.bp, st.word, m.emb, m.ld_emb, m.wt, m.gate_bias,
                                       m.kpad, m.H, m.E, rmax, ndev, stream)
            if r:
                return r
            if not m.untied:
                if m.split_lstm:
                    r = self.jlm_gemm_nt_split(p.h, m.H, rows, m.pmt_split, m.H, None, p.T, m.ldt, rows, None, m.t_descale,
                                               rmax, m.n_t, m.H, ndev, stream)
                else:
                    r = self.jlm_gemm_nt(p.h, m.H, rows, m.pmt, m.H, None, p.T, m.ldt, rows, None, rmax, m.n_t, m.H, ndev,
                                         stream)
                if r:
                    return r
            hybrid = all_mixed = False
            if full and not m.self_norm and not tile_form and m.mixed_segs and split and p.Tm:
                idx = [i for i in range(m.n_segs) if m.mixed_segs[i].B]
                if idx:
                    only = (type(m.mixed_segs[0]) * len(idx))(*[m.mixed_segs[i] for i in idx])
                    if self.jlm_mixed_t_stride(only, len(idx)) != p.ld_tm:
                        return -1
                    r = self._pack_t_for(m)(only, [m.mixed_t_scale[i] for i in idx], len(idx), p.T, m.ldt, rows, B if f == 0 else rmax,
                                            ndev, p.Tm, p.ld_tm, stream)
                    if r:
                        return r
                    hybrid = True
                    all_mixed = len(idx) == m.n_segs and not (bool(m.mixed_head_split) and any(m.mixed_head_split[i] > 0 for i in idx))
            cell = f * B
            perm = dynamic and bool(p.di_wwords) and bool(p.sg_wword)
            r = self.jlm_edge_logits_perm(m.segs, m.n_segs, m.b2, p.T, m.ldt, off(p.g0, cell), st.cnt, off(p.cidx, cell),
                                          p.sg_word, p.sg_wword if perm else None, p.sg_off, p.sidx, cell, p.sg_node, p.edge,
                                          beam, B, stream)
            if r:
                return r
            if not m.self_norm:
                if perm:
                    r = self.jlm_wordlist_lse_perm(m.segs, m.n_segs, m.b2, p.T, m.ldt, off(p.g0, cell), st.cnt, off(p.cidx, cell),
                                                   p.di_words, p.di_wwords, p.di_off, p.di_idx, 2 * cell, p.run_max, p.run_sum,
                                                   st.lse, 0, beam, B, stream)
                elif dynamic:
                    r = wl_lse(off(p.g0, cell), off(p.cidx, cell), p.di_words, p.di_off, p.di_idx, 2 * cell, 0, B, p.di_max)
                elif select:
                    r = wl_lse(off(p.g0, cell), off(p.cidx, cell), p.vs_words, p.vs_off, p.sidx, 0, 0, B, p.vs_max)
                elif tile_form:
                    n_parts = 0
                    for i in range(m.n_segs):
                        sg = m.segs[i]
                        if m.untied and m.untied_split and m.split_lstm:
                            r = self.jlm_vocab_lse_partials_split(m.untied_split, m.H, sg.v_end - sg.v_start, m.H, p.h, m.H, rows,
                                                                  off(m.b2, sg.v_start), m.untied_descale, p.part, rmax, n_parts,
                                                                  rmax, ndev, stream)
                        else:
                            r = self.jlm_vocab_lse_partials(sg.B, sg.ldb, sg.v_end - sg.v_start, sg.k, off(p.T, sg.t_off), m.ldt,
                                                            rows, off(m.b2, sg.v_start), p.part, rmax, n_parts, rmax, ndev, stream)
                        if r < 0:
                            return r
                        n_parts += r
                    if n_parts > p.max_parts:
                        return -1
                    pending, r = n_parts, 0
                else:
                    bound = B if f == 0 else rmax
                    r = -2
                    if all_mixed:
                        r = self.jlm_vocab_lse_mixed(m.mixed_segs, m.mixed_descale, m.mixed_s8, m.mixed_bias2, m.n_segs, p.Tm, p.ld_tm, p.part, rmax,
                                                     p.max_parts, bound, ndev, stream)
                    elif hybrid:
                        r = self.jlm_vocab_lse_hybrid(m.split_segs, m.split_t_scale, m.split_descale, m.split_bias_col, m.mixed_segs,
                                                      m.mixed_descale, m.mixed_s8, m.mixed_head_split, m.n_segs, m.b2, p.T, m.ldt, p.Tm,
                                                      p.ld_tm, rows, p.part, rmax, p.max_parts, bound, ndev, stream)
                    if r != -2:
                        pass
                    elif split:
                        r = self.jlm_vocab_lse_split(m.split_segs, m.split_t_scale, m.split_descale, m.split_bias_col,
                                                     m.n_segs, m.b2, p.T, m.ldt, rows, p.part, rmax, p.max_parts, bound, ndev,
                                                     stream)
                    else:
                        r = self.jlm_vocab_lse_stationary(m.segs, m.n_segs, m.b2, p.T, m.ldt, rows, p.part, rmax,
                                                          p.max_parts, bound, ndev, stream)
                    if r < 0:
                        return r
                    pending, r = r, 0
                if r:
                    return r
        return self.jlm_backtrace(lat, st, p.out_nodes, p.out_len, p.out_score, p.stride, stream)

    # ------------------------------------------------------------------ K1-K3
    def jlm_lstm_step(self, h_in, c_in, ld, h_out, c_out, rows, prev, word, emb, ld_emb, wt, bias, kpad, H, E,
                      n_rows_max, n_dev, stream):
        n = _n(n_rows_max, n_dev)
        if n == 0:
            return 0
        g = _rows(rows, n)
        gmax = int(g.max()) + 1
        p = view(prev, gmax, np.int32)[g].astype(np.int64)
        w = view(word, gmax, np.int32)[g].astype(np.int64)
        hmax = max(gmax, int(p.max()) + 1)
        hin = view(h_in, hmax * ld, np.float32).reshape(hmax, ld)
        cin = view(c_in, hmax * ld, np.float32).reshape(hmax, ld)
        embv = view(emb, (int(w.max()) + 1) * ld_emb, np.float32).reshape(-1, ld_emb)
        x = np.zeros((n, kpad), dtype=np.float32)
        ok = p >= 0
        x[ok, :H] = hin[p[ok], :H]
        x[:, H:H + E] = embv[w, :E]
        cp = np.zeros((n, H), dtype=np.float32)
        cp[ok] = cin[p[ok], :H]
        W = view(wt, 4 * H * kpad, np.float32).reshape(4 * H, kpad)
        b = view(bias, 4 * H, np.float32)
        z = (x.astype(np.float64) @ W.T.astype(np.float64)).astype(np.float32) + b
        u = np.arange(H)
        zi, zf, zo, zg = (z[:, (u // 16) * 64 + k * 16 + (u % 16)] for k in range(4))
        sig = lambda t: (1.0 / (np.exp(-t.astype(np.float64)) + 1.0)).astype(np.float32)
        cn = cp * sig(zf) + np.tanh(zg) * sig(zi)
        hn = np.tanh(cn) * sig(zo)
        hout = view(h_out, gmax * ld, np.float32).reshape(gmax, ld)
        cout = view(c_out, gmax * ld, np.float32).reshape(gmax, ld)
        hout[g, :H] = hn
        cout[g, :H] = cn
        return 0

    # ---- split rows helpers (include/jlm_hip.h "f16x3")
    @staticmethod
    def _split_view(ptr, nrows, ld):
        """[nrows, ld // 8, 2 planes, 8] float16 view of split rows"""
        return view(ptr, nrows * ld * 2, np.float16).reshape(nrows, ld // 8, 2, 8)

    @classmethod
    def _split_read(cls, ptr, nrows, ld):
        sp = cls._split_view(ptr, nrows, ld).astype(np.float64)
        with np.errstate(invalid="ignore"):        # rows nobody reads may hold uninitialised storage
            return (sp[:, :, 0, :] + sp[:, :, 1, :]).reshape(nrows, ld)

    @classmethod
    def _split_write(cls, ptr, nrows, ld, row_ids, x):
        """x [len(row_ids), k] float32 (already scaled) -> split rows row_ids; k <= ld, k % 8 == 0"""
        out = cls._split_view(ptr, nrows, ld)
        hi = x.astype(np.float16)
        lo = (x - hi.astype(np.float32)).astype(np.float16)
        nb = x.shape[1] // 8
        out[row_ids, :nb, 0, :] = hi.reshape(len(row_ids), nb, 8)
        out[row_ids, :nb, 1, :] = lo.reshape(len(row_ids), nb, 8)

    def jlm_lstm_step_xg(self, h_in, c_in, ld, h_out, c_out, rows, prev, word, wt8, xgate8, H, descale, h_scale, h_f32_out,
                         n_rows_max, n_dev, stream):
        """table form, gate-interleave-8 order: n = (u / 8) * 32 + gate * 8 + u % 8; xgate8 pre-multiplied by 1 / descale"""
        if H <= 0 or H % 32 or ld % 16 or ld < H:
            return -1
        n = _n(n_rows_max, n_dev)
        if n <= 0:
            return 0
        g = _rows(rows, n)
        gmax = int(g.max()) + 1
        p = view(prev, gmax, np.int32)[g].astype(np.int64)
        w = view(word, gmax, np.int32)[g].astype(np.int64)
        hmax = max(gmax, int(p.max()) + 1)
        hin = self._split_read(h_in, hmax, ld)
        cin = view(c_in, hmax * ld, np.float32).reshape(hmax, ld)
        ok = p >= 0
        x = np.zeros((n, H), dtype=np.float64)
        x[ok] = hin[p[ok], :H]
        cp = np.zeros((n, H), dtype=np.float32)
        cp[ok] = cin[p[ok], :H]
        W = self._split_read(wt8, 4 * H, H)
        add = view(xgate8, (int(w.max()) + 1) * 4 * H, np.float32).reshape(-1, 4 * H)[w].astype(np.float64)
        z = ((x @ W.T + add) * float(descale)).astype(np.float32)
        u = np.arange(H)
        zi, zf, zo, zg = (z[:, (u // 8) * 32 + k * 8 + (u % 8)] for k in range(4))
        sig = lambda t: (1.0 / (np.exp(-t.astype(np.float64)) + 1.0)).astype(np.float32)
        cn = cp * sig(zf) + np.tanh(zg) * sig(zi)
        hn = (np.tanh(cn) * sig(zo)).astype(np.float32)
        self._split_write(h_out, gmax, ld, g, hn * np.float32(h_scale))
        cout = view(c_out, gmax * ld, np.float32).reshape(gmax, ld)
        cout[g, :H] = cn
        if _p(h_f32_out):
            view(h_f32_out, gmax * ld, np.float32).reshape(gmax, ld)[g, :H] = hn
        return 0

    def jlm_vocab_lse_partials_split(self, Bsplit, ldb, n_vocab, K, Tsplit, ldt, rows, bias, descale, part, ld_part, tile0,
                                     n_rows_max, n_dev, stream):
        if K % 16 or ldb % 16 or ldt % 16:
            return -1
        n = _n(n_rows_max, n_dev)
        ntile = (n_vocab + 127) // 128
        if n == 0:
            return ntile
        g = _rows(rows, n)
        Tm = self._split_read(Tsplit, int(g.max()) + 1, ldt)[g, :K]
        Bm = self._split_read(Bsplit, n_vocab, ldb)[:, :K]
        y = ((Tm @ Bm.T) * float(descale)).astype(np.float32) + view(bias, n_vocab, np.float32)
        pv = view(part, (tile0 + ntile) * ld_part * 2, np.float32).reshape(tile0 + ntile, ld_part, 2)
        for t in range(ntile):
            yt = y[:, t * 128:(t + 1) * 128].astype(np.float64)
            mx = yt.max(axis=1)
            pv[tile0 + t, :n, 0] = mx
            pv[tile0 + t, :n, 1] = np.exp(yt - mx[:, None]).sum(axis=1)
        return ntile

    def jlm_gemm_nt_split(self, A, lda, a_rows, B, ldb, b_rows, C, ldc, c_rows, bias, descale, M, N, K, m_dev, stream):
        if K % 16 or lda % 16 or ldb % 16:
            return -1
        m = _n(M, m_dev)
        if m == 0 or N == 0:
            return 0
        ar, br, cr = _rows(a_rows, m), _rows(b_rows, N), _rows(c_rows, m)
        Av = self._split_read(A, int(ar.max()) + 1, lda)[ar, :K]
        Bv = self._split_read(B, int(br.max()) + 1, ldb)[br, :K]
        out = ((Av @ Bv.T) * float(descale)).astype(np.float32)
        if _p(bias):
            out = out + view(bias, N, np.float32)
        base = _p(C)
        for i, r in enumerate(cr):
            view(base + 4 * int(r) * ldc, N, np.float32)[:] = out[i]
        return 0

    # --------------------------------------------------------------------- K4
    def jlm_gemm_nt(self, A, lda, a_rows, B, ldb, b_rows, C, ldc, c_rows, bias, M, N, K, m_dev, stream):
        m = _n(M, m_dev)
        if m == 0 or N == 0:
            return 0
        ar, br, cr = _rows(a_rows, m), _rows(b_rows, N), _rows(c_rows, m)
        Av = view(A, (int(ar.max()) + 1) * lda, np.float32).reshape(-1, lda)[ar, :K]
        Bv = view(B, (int(br.max()) + 1) * ldb, np.float32).reshape(-1, ldb)[br, :K]
        out = (Av.astype(np.float64) @ Bv.T.astype(np.float64)).astype(np.float32)
        if _p(bias):
            out = out + view(bias, N, np.float32)
        # C may be a column-offset view: address rows individually
        base = _p(C)
        for i, r in enumerate(cr):
            view(base + 4 * int(r) * ldc, N, np.float32)[:] = out[i]
        return 0

    # ------------------------------------------------------------------ K5+K6
    def jlm_vocab_lse_partials(self, Bseg, ldb, n_vocab, K, T, ldt, rows, bias, part, ld_part, tile0, n_rows_max,
                               n_dev, stream):
        ntiles = (n_vocab + 127) // 128
        n = _n(n_rows_max, n_dev)
        if n == 0:
            return ntiles
        g = _rows(rows, n)
        base = _p(T)
        Tv = np.stack([view(base + 4 * int(r) * ldt, K, np.float32) for r in g])
        Bv = view(Bseg, n_vocab * ldb, np.float32).reshape(n_vocab, ldb)[:, :K]
        y = (Tv.astype(np.float64) @ Bv.T.astype(np.float64)).astype(np.float32) + view(bias, n_vocab, np.float32)
        pv = view(part, (tile0 + ntiles) * ld_part * 2, np.float32).reshape(-1, ld_part, 2)
        for t in range(ntiles):
            yt = y[:, t * 128:(t + 1) * 128]
            mx = yt.max(axis=1)
            pv[tile0 + t, :n, 0] = mx
            pv[tile0 + t, :n, 1] = np.exp(yt - mx[:, None]).sum(axis=1)
        return ntiles

    def jlm_vocab_lse_stationary(self, segs, n_segs, b2, T, ldt, rows, part, ld_part, max_parts, n_rows_max, n_dev,
                                 stream):
        """Contract: any partition of the vocabulary into <= max_parts slices; here one per segment."""
        if n_segs > max_parts:
            return -1
        n = _n(n_rows_max, n_dev)
        pv = view(part, n_segs * ld_part * 2, np.float32).reshape(n_segs, ld_part, 2)
        if n == 0:
            return n_segs
        g = _rows(rows, n)
        for i in range(n_segs):
            sg = segs[i]
            if sg.k > 256:
                return -2
            nv = sg.v_end - sg.v_start
            Tv = np.stack([view(_p(T) + 4 * (int(r) * ldt + sg.t_off), sg.k, np.float32) for r in g]).astype(np.float64)
            Bv = view(sg.B, nv * sg.ldb, np.float32).reshape(nv, sg.ldb)[:, :sg.k].astype(np.float64)
            y = (Tv @ Bv.T).astype(np.float32) + view(_p(b2) + 4 * sg.v_start, nv, np.float32)
            mx = y.max(axis=1)
            pv[i, :n, 0] = mx
            pv[i, :n, 1] = np.exp(y - mx[:, None]).sum(axis=1)
        return n_segs

    def jlm_pack_split_f16(self, src, rows, k, ld, scale, dst, ld_dst, stream):
        """split rows: per 8 values [8 x f16 hi][8 x f16 lo]; blocks covering pad16(k) are written"""
        k16 = (k + 15) // 16 * 16
        if rows < 0 or k <= 0 or ld < k or ld_dst % 16 or ld_dst < k16:
            return -1
        if rows == 0:
            return 0
        as_strided = np.lib.stride_tricks.as_strided
        # src / dst may be column-offset views of wider matrices: the last row is only k (k16) long
        flat = view(src, (rows - 1) * ld + k, np.float32)
        x = np.zeros((rows, k16), dtype=np.float32)
        x[:, :k] = as_strided(flat, shape=(rows, k), strides=(4 * ld, 4))
        x *= np.float32(scale)
        hi = x.astype(np.float16)
        lo = (x - hi.astype(np.float32)).astype(np.float16)
        dflat = view(dst, ((rows - 1) * ld_dst + k16) * 2, np.float16)
        out = as_strided(dflat, shape=(rows, k16 // 8, 2, 8), strides=(4 * ld_dst, 32, 16, 2))
        out[:, :, 0, :] = hi.reshape(rows, k16 // 8, 8)
        out[:, :, 1, :] = lo.reshape(rows, k16 // 8, 8)
        return 0

    def jlm_dequant_u8(self, code, rows, k, ld_code, codebook, n_codes, dst, ld_dst, stream):
        if rows < 0 or k <= 0 or ld_code < k or ld_dst < k or not 1 <= n_codes <= 256:
            return -1
        if rows == 0:
            return 0
        buf = (ctypes.c_uint8 * (rows * ld_code)).from_address(_p(code))
        cv = np.frombuffer(buf, dtype=np.uint8).reshape(rows, ld_code)[:, :k]
        book = view(codebook, n_codes, np.float32)
        out = view(dst, (rows - 1) * ld_dst + k, np.float32)
        for r in range(rows):
            out[r * ld_dst:r * ld_dst + k] = book[cv[r]]
        return 0

    def jlm_pack_split_f16_col(self, v, rows, scale, dst, ld_dst, col, stream):
        if rows < 0 or ld_dst % 16 or col < 0 or col >= ld_dst:
            return -1
        if rows == 0:
            return 0
        x = view(v, rows, np.float32) * np.float32(scale)
        hi = x.astype(np.float16)
        lo = (x - hi.astype(np.float32)).astype(np.float16)
        out = self._split_view(dst, rows, ld_dst)
        out[:, col // 8, 0, col % 8] = hi
        out[:, col // 8, 1, col % 8] = lo
        return 0

    def jlm_vocab_lse_split(self, segs, t_scale, descale, bias_col, n_segs, b2, T, ldt, rows, part, ld_part, max_parts,
                            n_rows_max, n_dev, stream):
        """Contract of jlm_vocab_lse_stationary, operands read from split rows."""
        if n_segs > max_parts:
            return -1
        n = _n(n_rows_max, n_dev)
        pv = view(part, n_segs * ld_part * 2, np.float32).reshape(n_segs, ld_part, 2)
        if n == 0:
            return n_segs
        g = _rows(rows, n)
        for i in range(n_segs):
            sg = segs[i]
            bc = bias_col[i] if bias_col is not None and bias_col else -1
            if sg.k > 256 or sg.ldb % 16 or (bc >= 0 and (bc != sg.k or sg.k % 16 == 0)):
                return -2
            nv = sg.v_end - sg.v_start
            Tv = np.stack([view(_p(T) + 4 * (int(r) * ldt + sg.t_off), sg.k, np.float32) for r in g]).astype(np.float64)
            sp = view(sg.B, nv * sg.ldb * 2, np.float16).reshape(nv, sg.ldb // 8, 2, 8).astype(np.float64)
            Bfull = (sp[:, :, 0, :] + sp[:, :, 1, :]).reshape(nv, sg.ldb)
            y = (Tv * float(t_scale[i])) @ Bfull[:, :sg.k].T * float(descale[i])
            if bc >= 0:          # the bias column times the 1.0 the kernel feeds (scaled by t_scale like every T value)
                y = (y + float(t_scale[i]) * Bfull[:, bc][None, :] * float(descale[i])).astype(np.float32)
            else:
                y = y.astype(np.float32) + view(_p(b2) + 4 * sg.v_start, nv, np.float32)
            mx = y.max(axis=1)
            pv[i, :n, 0] = mx
            pv[i, :n, 1] = np.exp(y - mx[:, None]).sum(axis=1)
        return n_segs

    # ------------------------------------------------------------------ ABI 7: mixed rows (f16 hi + int8 cross-term planes)
    @staticmethod
    def _mixed_view(ptr, rows, nb, ld_bytes=None):
        """(hi f16 [rows, 32 nb], hi8 int8 [rows, 32 nb], lo8 int8 [rows, 32 nb]) views of rows of 128-byte blocks"""
        ld_bytes = ld_bytes or nb * 128
        raw = np.frombuffer((ctypes.c_uint8 * (rows * ld_bytes)).from_address(_p(ptr)), dtype=np.uint8).reshape(rows, ld_bytes)
        blk = raw[:, :nb * 128].reshape(rows, nb, 128)
        return blk[:, :, :64], blk[:, :, 64:96], blk[:, :, 96:128]

    @staticmethod
    def _quant(x32, s):
        """x (f32) -> (hi f16, hi8, lo8) with int8 scale s for hi and s / 2048 for the residual"""
        hi = x32.astype(np.float16)
        lo = x32 - hi.astype(np.float32)
        q = lambda v: np.clip(np.rint(v), -127, 127).astype(np.int8)
        return hi, q(hi.astype(np.float32) / s), q(lo / (s / np.float32(2048.0)))

    def jlm_pack_mixed(self, src, rows, k, ld, bias, scale, bias_scale, s8, dst, ld_dst, stream):
        nb = ld_dst // 32
        if rows < 0 or k <= 0 or ld < k or ld_dst % 32 or nb not in ((k + 2 + 31) // 32, (k + 31) // 32) or nb > 8:
            return -1
        cols = k + 2 <= 32 * nb                    # room for the bias columns
        if rows == 0:
            return 0
        flat = view(src, (rows - 1) * ld + k, np.float32)
        x = np.zeros((rows, 32 * nb), dtype=np.float32)
        x[:, :k] = np.lib.stride_tricks.as_strided(flat, shape=(rows, k), strides=(4 * ld, 4)) * np.float32(scale)
        mx6 = float(s8) == 0.0
        if float(s8) < 0.0 or (mx6 and nb > 8):
            return -1
        hi, h8, l8 = self._quant(x, np.float32(s8 if not mx6 else 1.0))
        if cols:
            xb = (view(bias, rows, np.float32) * np.float32(bias_scale)) if _p(bias) else np.zeros(rows, dtype=np.float32)
            bh = xb.astype(np.float16)
            hi[:, k] = bh
            hi[:, k + 1] = ((xb - bh.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
        if mx6:
            self._mx6_store(dst, rows, nb, nb * 128, x, hi, k, swap=False)
            return 0
        h8[:, k:] = 0
        l8[:, k:] = 0
        vh, v8, vl = self._mixed_view(dst, rows, nb)
        vh[:] = hi.reshape(rows, nb, 32).view(np.uint8).reshape(rows, nb, 64)
        v8[:] = h8.reshape(rows, nb, 32).view(np.uint8)
        vl[:] = l8.reshape(rows, nb, 32).view(np.uint8)
        return 0

    # ------------------------------------------------------------------ ABI 11: mx6 rows (FP6 cross-term planes with E8M0 block scales)
    @staticmethod
    def _mx6_block_byte(amax):
        """csrc/jlm_mx6_body.h mx6_block_byte: the smallest power of two s = 2^(byte - 127) with amax <= 7.5 s; 0 for an all-zero block"""
        m, ex = np.frexp(amax.astype(np.float32))               # amax = m 2^ex, m in [0.5, 1)
        e = np.where(m <= 0.9375, ex - 3, ex - 2)
        return np.where(amax > 0, np.clip(e + 127, 0, 254), 0).astype(np.int64)

    def _mx6_store(self, ptr, rows, nb, ld_bytes, x, hi, k, swap):
        """x [rows, 32 nb] f32 (scaled), hi f16 (with the bias columns filled in) -> rows of 128-byte mx6 blocks: granules 0-3 f16 hi;
        the FP6 planes of the REAL k-values (hi6 of the f16 hi, lo6 of the residual): vocabulary rows half 0 = hi6, half 1 = lo6;
        hypothesis rows (swap) the other way round; granule 7 of block 0: the halves' scale bytes"""
        from tests import mx6_emu as E
        raw = np.frombuffer((ctypes.c_uint8 * (rows * ld_bytes)).from_address(_p(ptr)), dtype=np.uint8).reshape(rows, ld_bytes)
        blk = raw[:, :nb * 128].reshape(rows, nb, 128)
        blk[:, :, :64] = hi.reshape(rows, nb, 32).view(np.uint8).reshape(rows, nb, 64)
        real = (np.arange(32 * nb) < k)[None, :]
        h32 = np.where(real, x.astype(np.float16).astype(np.float32), np.float32(0.0))
        lo = np.where(real, x - x.astype(np.float16).astype(np.float32), np.float32(0.0))
        planes = []
        for v in (h32, lo):
            vb = v.reshape(rows, nb, 32).astype(np.float64)
            byte = self._mx6_block_byte(np.abs(vb).max(axis=2))
            q = E.e2m3_round(vb * np.exp2(127.0 - byte)[:, :, None])
            code = E.e2m3_code(q)
            bits = ((code[..., None] >> np.arange(6)) & 1).astype(np.uint8).reshape(rows, nb, 192)
            planes.append((np.packbits(bits, axis=2, bitorder="little"), byte.astype(np.uint8)))
        (ph, bh), (pl, bl) = planes
        half0, half1, b0, b1 = (pl, ph, bl, bh) if swap else (ph, pl, bh, bl)
        blk[:, :, 64:80] = half0[:, :, :16]
        blk[:, :, 80:88] = half0[:, :, 16:24]
        blk[:, :, 88:96] = half1[:, :, 16:24]
        blk[:, :, 96:112] = half1[:, :, :16]
        blk[:, :, 112:128] = 0
        blk[:, 0, 112:112 + nb] = b0
        blk[:, 0, 120:120 + nb] = b1

    @staticmethod
    def _mx6_load(ptr, rows, nb, ld_bytes, swap):
        """-> (hi f64 [rows, 32 nb] from the f16 part, hi6 f64, lo6 f64: the FP6 planes with their block scales applied)"""
        from tests import mx6_emu as E
        raw = np.frombuffer((ctypes.c_uint8 * (rows * ld_bytes)).from_address(_p(ptr)), dtype=np.uint8).reshape(rows, ld_bytes)
        blk = raw[:, :nb * 128].reshape(rows, nb, 128)
        hi = np.ascontiguousarray(blk[:, :, :64]).view(np.float16).reshape(rows, 32 * nb).astype(np.float64)
        half0 = np.concatenate([blk[:, :, 64:80], blk[:, :, 80:88]], axis=2)
        half1 = np.concatenate([blk[:, :, 96:112], blk[:, :, 88:96]], axis=2)
        out = []
        for plane, sc in ((half0, blk[:, 0, 112:112 + nb]), (half1, blk[:, 0, 120:120 + nb])):
            bits = np.unpackbits(np.ascontiguousarray(plane), axis=2, bitorder="little").reshape(rows, nb, 32, 6)
            code = (bits * (1 << np.arange(6))).sum(axis=3)
            out.append((E.e2m3_value(code) * np.exp2(sc.astype(np.float64) - 127.0)[:, :, None]).reshape(rows, 32 * nb))
        h6, l6 = (out[1], out[0]) if swap else (out[0], out[1])
        return hi, h6, l6

    def jlm_pack_t_mixed6(self, segs, t_scale, n_segs, T, ldt, rows, n_rows_max, n_dev, Tm, ld_tm, stream):
        """jlm_pack_t_mixed's rows in the mx6 form (halves swapped against the vocabulary rows)"""
        if n_segs < 1 or n_segs > 8 or ldt % 4:
            return -1
        want = self.jlm_mixed_t_stride(segs, n_segs)
        if want == -2:
            return -2
        if ld_tm != want:
            return -1
        if any(segs[i].ldb // 32 > 8 for i in range(n_segs)):
            return -2
        n = _n(n_rows_max, n_dev)
        if n <= 0:
            return 0
        g = _rows(rows, n)
        off = 0
        for i in range(n_segs):
            sg = segs[i]
            nb = sg.ldb // 32
            Tv = np.stack([view(_p(T) + 4 * (int(r) * ldt + sg.t_off), sg.k, np.float32) for r in g])
            x = np.zeros((n, 32 * nb), dtype=np.float32)
            x[:, :sg.k] = Tv * np.float32(float(t_scale[i]) * 1.4426950408889634)
            hi = x.astype(np.float16)
            if sg.k + 2 <= 32 * nb:
                hi[:, sg.k] = np.float16(t_scale[i])
                hi[:, sg.k + 1] = np.float16(float(t_scale[i]) / 2048.0)
            self._mx6_store(_p(Tm) + off, n, nb, ld_tm * 4, x, hi, sg.k, swap=True)
            off += nb * 128
        return 0

    def jlm_mixed_t_stride(self, segs, n_segs):
        if n_segs < 1 or n_segs > 8:
            return -1
        for i in range(n_segs):
            if segs[i].ldb % 32 or segs[i].ldb // 32 not in ((segs[i].k + 2 + 31) // 32, (segs[i].k + 31) // 32):
                return -1
        b = sum(segs[i].ldb * 4 for i in range(n_segs))
        if b // 128 > 32:
            return -2
        return (b + 4 * 8 + 15) // 16 * 4

    def jlm_pack_t_mixed(self, segs, t_scale, n_segs, T, ldt, rows, n_rows_max, n_dev, Tm, ld_tm, stream):
        """packed row r (COMPACT) = hypothesis row rows[r]; per segment the blocks of x = T 2^eT log2 e, the bias constants at
        f16 columns k, k + 1, and the row's int8 scale per segment in the last 8 floats of the row"""
        if n_segs < 1 or n_segs > 8 or ldt % 4:
            return -1
        want = self.jlm_mixed_t_stride(segs, n_segs)
        if want == -2:
            return -2
        if ld_tm != want:
            return -1
        n = _n(n_rows_max, n_dev)
        if n <= 0:
            return 0
        g = _rows(rows, n)
        off = 0
        scales = view(_p(Tm), n * ld_tm, np.float32).reshape(n, ld_tm)[:, ld_tm - 8:]
        for i in range(n_segs):
            sg = segs[i]
            nb = sg.ldb // 32
            Tv = np.stack([view(_p(T) + 4 * (int(r) * ldt + sg.t_off), sg.k, np.float32) for r in g])
            x = np.zeros((n, 32 * nb), dtype=np.float32)
            x[:, :sg.k] = Tv * np.float32(float(t_scale[i]) * 1.4426950408889634)
            amax = np.abs(x.astype(np.float16).astype(np.float32)).max(axis=1)
            with np.errstate(divide="ignore"):
                s_t = np.where(amax > 0, np.exp2(np.ceil(np.log2(np.maximum(amax, 1e-37) / 127.0))), 1.0).astype(np.float32)
            hi, h8, l8 = self._quant(x, s_t[:, None])
            if sg.k + 2 <= 32 * nb:
                hi[:, sg.k] = np.float16(t_scale[i])
                hi[:, sg.k + 1] = np.float16(float(t_scale[i]) / 2048.0)
            h8[:, sg.k:] = 0
            l8[:, sg.k:] = 0
            vh, v8, vl = self._mixed_view(_p(Tm) + off, n, nb, ld_tm * 4)
            vh[:] = hi.reshape(n, nb, 32).view(np.uint8).reshape(n, nb, 64)
            v8[:] = h8.reshape(n, nb, 32).view(np.uint8)
            vl[:] = l8.reshape(n, nb, 32).view(np.uint8)
            scales[:, i] = s_t
            off += nb * 128
        return 0

    def _mixed_logits(self, sg, Tm, ld_tm, tm_off, slot, n, descale, s8, bias2=None):
        """base-e logits [n, words] of a mixed segment: (hi.hi in f16 products + int8 cross terms x s_t s8 / 2048) descale ln 2"""
        nb, nv = sg.ldb // 32, sg.v_end - sg.v_start
        if float(s8) == 0.0:                             # ABI 11: mx6 rows -- hi.hi + hi6.lo6 + lo6.hi6 into one f32 accumulator
            bh, b6h, b6l = self._mx6_load(sg.B, nv, nb, nb * 128, swap=False)
            th, t6h, t6l = self._mx6_load(_p(Tm) + tm_off, n, nb, ld_tm * 4, swap=True)
            y2 = (th @ bh.T + t6h @ b6l.T + t6l @ b6h.T).astype(np.float32) * np.float32(descale)
            if sg.k + 2 > 32 * nb:
                y2 = y2 + view(_p(bias2) + 4 * sg.v_start, nv, np.float32)[None, :]
            return y2.astype(np.float64) * 0.6931471805599453
        f = lambda v, dt: np.ascontiguousarray(v).view(dt).reshape(v.shape[0], -1).astype(np.float64)
        bh, b8, bl = self._mixed_view(sg.B, nv, nb)
        th, t8, tl = self._mixed_view(_p(Tm) + tm_off, n, nb, ld_tm * 4)
        s_t = view(_p(Tm), n * ld_tm, np.float32).reshape(n, ld_tm)[:, ld_tm - 8 + slot].astype(np.float64)
        main = f(th, np.float16) @ f(bh, np.float16).T
        cross = f(t8, np.int8) @ f(bl, np.int8).T + f(tl, np.int8) @ f(b8, np.int8).T
        y2 = (main + cross * (s_t * (float(s8) / 2048.0))[:, None]).astype(np.float32) * np.float32(descale)
        if sg.k + 2 > 32 * nb:                       # no bias columns: the biases (base-2 units) come separately
            y2 = y2 + view(_p(bias2) + 4 * sg.v_start, nv, np.float32)[None, :]
        return y2.astype(np.float64) * 0.6931471805599453

    def jlm_vocab_lse_mixed_fr(self, *a):
        """(the fixed-reference form differs in rounding only: the double runs the same emulation)"""
        return self.jlm_vocab_lse_mixed(*a)

    def jlm_vocab_lse_mixed(self, segs, descale, s8, bias2, n_segs, Tm, ld_tm, part, ld_part, max_parts, n_rows_max, n_dev, stream):
        if n_segs < 1 or n_segs > max_parts or ld_tm != self.jlm_mixed_t_stride(segs, n_segs):
            return -1
        xb = [segs[i].k + 2 > segs[i].ldb for i in range(n_segs)]
        if any(xb) and (not all(xb) or not _p(bias2) or any((segs[i].ldb // 32) % 2 for i in range(n_segs))):
            return -2
        n6 = sum(float(s8[i]) == 0.0 for i in range(n_segs))
        if n6 and (n6 != n_segs or any(segs[i].ldb // 32 > 8 for i in range(n_segs))):       # one format per launch; mx6: <= 8 blocks
            return -2
        n = _n(n_rows_max, n_dev)
        pv = view(part, n_segs * ld_part * 2, np.float32).reshape(n_segs, ld_part, 2)
        off = 0
        for i in range(n_segs):
            if n:
                y = self._mixed_logits(segs[i], Tm, ld_tm, off, i, n, descale[i], s8[i], bias2)
                mx = y.max(axis=1)
                pv[i, :n, 0] = mx
                pv[i, :n, 1] = np.exp(y - mx[:, None]).sum(axis=1)
            off += segs[i].ldb * 4
        return n_segs

    def jlm_vocab_lse_hybrid(self, segs, t_scale, descale, bias_col, mixed, mx_descale, mx_s8, head_split, n_segs, b2, T, ldt, Tm, ld_tm,
                             rows, part, ld_part, max_parts, n_rows_max, n_dev, stream):
        """mixed[i].B: segment i from its mixed rows and the packed hypothesis rows, else from its split rows (the contract of
        jlm_vocab_lse_split); -2 for the shapes the kernel does not host.  ABI 10: the first head_split[i] words of a mixed segment from
        its split rows (one more slice per such segment)"""
        if n_segs < 1 or n_segs > max_parts:
            return -1
        is_mixed = [bool(mixed[i].B) for i in range(n_segs)]
        if not any(is_mixed):
            return -2
        cut = [int(head_split[i]) if (head_split is not None and bool(head_split) and is_mixed[i]) else 0 for i in range(n_segs)]
        for i in range(n_segs):
            k = segs[i].k
            if cut[i] < 0 or cut[i] % 128 or cut[i] >= segs[i].v_end - segs[i].v_start:
                return -1
            if is_mixed[i]:
                if ((k + 2 + 31) // 32, (k + 2 + 15) // 16) not in ((7, 13), (4, 7), (2, 4)) or mixed[i].ldb != 32 * ((k + 2 + 31) // 32):
                    return -2
                if not (float(mx_s8[i]) > 0.0):          # (ABI 11: mx6 rows have no body in this launch)
                    return -2
            if not is_mixed[i] or cut[i]:
                ns = (k + 15) // 16
                if ns > 13 or 7 < ns < 13 or bias_col is None or bias_col[i] != k or k % 16 == 0:
                    return -2
        if n_segs + sum(1 for c in cut if c) > min(max_parts, 8):
            return -2
        n = _n(n_rows_max, n_dev)
        n_out = n_segs + sum(1 for c in cut if c)
        pv = view(part, n_out * ld_part * 2, np.float32).reshape(n_out, ld_part, 2)
        Seg = type(segs[0])
        one = Seg * 1
        tm_off = slot = o = 0
        for i in range(n_segs):
            if not is_mixed[i] or cut[i]:
                sg = Seg(segs[i].v_start, segs[i].v_start + cut[i] if cut[i] else segs[i].v_end, segs[i].k, segs[i].t_off, segs[i].B, segs[i].ldb)
                r = self.jlm_vocab_lse_split(one(sg), [t_scale[i]], [descale[i]], [bias_col[i]], 1, b2, T, ldt, rows,
                                             _p(part) + 8 * o * ld_part, ld_part, 1, n_rows_max, n_dev, stream)
                if r != 1:
                    return r
                o += 1
            if is_mixed[i]:
                if n:
                    y = self._mixed_logits(mixed[i], Tm, ld_tm, tm_off, slot, n, mx_descale[i], mx_s8[i])[:, cut[i]:]
                    mx = y.max(axis=1)
                    pv[o, :n, 0] = mx
                    pv[o, :n, 1] = np.exp(y - mx[:, None]).sum(axis=1)
                tm_off += mixed[i].ldb * 4
                slot += 1
                o += 1
        return n_out

    def jlm_lse_combine(self, part, ld_part, n_tiles, rows, lse, n_rows_max, n_dev, stream):
        n = _n(n_rows_max, n_dev)
        if n == 0:
            return 0
        g = _rows(rows, n)
        pv = view(part, n_tiles * ld_part * 2, np.float32).reshape(n_tiles, ld_part, 2)[:, :n].astype(np.float64)
        mx = pv[:, :, 0].max(axis=0)
        s = (pv[:, :, 1] * np.exp(pv[:, :, 0] - mx[None, :])).sum(axis=0)
        view(lse, int(g.max()) + 1, np.float64)[g] = mx + np.log(s)
        return 0

    # ------------------------------------------------------------- word lists
    def _segs(self, segs, n_segs):
        return [segs[i] for i in range(n_segs)]

    def _word_logits(self, segs, b2, T, ldt, g0, nrows, words, wwords=None):
        """[len(words), nrows] float32 logits of hypothesis rows g0.. for the words (wwords: the words whose weight rows are
        used, the bias stays the list word's -- the *_perm entry points)."""
        out = np.zeros((len(words), nrows), dtype=np.float32)
        for i, w0 in enumerate(words):
            w = int(wwords[i]) if wwords is not None else int(w0)
            for sg in segs:
                if sg.v_start <= w < sg.v_end:
                    brow = view(sg.B + 4 * (w - sg.v_start) * sg.ldb, sg.k, np.float32).astype(np.float64)
                    for k in range(nrows):
                        t = view(_p(T) + 4 * ((g0 + k) * ldt + sg.t_off), sg.k, np.float32).astype(np.float64)
                        out[i, k] = np.float32(np.dot(brow, t))
                    break
            out[i] += view(_p(b2) + 4 * int(w0), 1, np.float32)[0]
        return out

    def _groups(self, g0, cnt, cnt_idx, wl, wl_off, wl_idx, wl_base, n_groups, beam):
        g0v = view(g0, n_groups, np.int32)
        ci = view(cnt_idx, n_groups, np.int32)
        li = view(wl_idx, n_groups, np.int32)
        for j in range(n_groups):
            nrows = min(int(view(_p(cnt) + 4 * int(ci[j]), 1, np.int32)[0]), beam)
            lid = wl_base + int(li[j])
            a, b = (int(x) for x in view(_p(wl_off) + 4 * lid, 2, np.int32))
            words = view(_p(wl) + 4 * a, b - a, np.int32) if b > a else np.zeros(0, np.int32)
            yield int(g0v[j]), nrows, a, words

    def jlm_edge_logits(self, segs, n_segs, b2, T, ldt, g0, cnt, cnt_idx, wl, wl_off, wl_idx, wl_base, wl_out, edge,
                        beam, n_groups, stream):
        return self.jlm_edge_logits_perm(segs, n_segs, b2, T, ldt, g0, cnt, cnt_idx, wl, None, wl_off, wl_idx, wl_base, wl_out,
                                         edge, beam, n_groups, stream)

    def jlm_edge_logits_perm(self, segs, n_segs, b2, T, ldt, g0, cnt, cnt_idx, wl, wl_w, wl_off, wl_idx, wl_base, wl_out, edge,
                             beam, n_groups, stream):
        sg = self._segs(segs, n_segs)
        for gb, nrows, a, words in self._groups(g0, cnt, cnt_idx, wl, wl_off, wl_idx, wl_base, n_groups, beam):
            if nrows <= 0 or len(words) == 0:
                continue
            ww = view(_p(wl_w) + 4 * a, len(words), np.int32) if wl_w else None
            y = self._word_logits(sg, b2, T, ldt, gb, nrows, words, ww)
            outs = view(_p(wl_out) + 4 * a, len(words), np.int32)
            for i, n in enumerate(outs):
                view(_p(edge) + 4 * int(n) * beam, nrows, np.float32)[:] = y[i]
        return 0

    def jlm_wordlist_lse(self, segs, n_segs, b2, T, ldt, g0, cnt, cnt_idx, wl, wl_off, wl_idx, wl_base, run_max,
                         run_sum, lse, merge, beam, n_groups, stream):
        return self.jlm_wordlist_lse_perm(segs, n_segs, b2, T, ldt, g0, cnt, cnt_idx, wl, None, wl_off, wl_idx, wl_base, run_max,
                                          run_sum, lse, merge, beam, n_groups, stream)

    def jlm_wordlist_lse_perm(self, segs, n_segs, b2, T, ldt, g0, cnt, cnt_idx, wl, wl_w, wl_off, wl_idx, wl_base, run_max,
                              run_sum, lse, merge, beam, n_groups, stream):
        sg = self._segs(segs, n_segs)
        for gb, nrows, a, words in self._groups(g0, cnt, cnt_idx, wl, wl_off, wl_idx, wl_base, n_groups, beam):
            if nrows <= 0:
                continue
            rm = view(_p(run_max) + 4 * gb, nrows, np.float32)
            rs = view(_p(run_sum) + 8 * gb, nrows, np.float64)
            ls = view(_p(lse) + 8 * gb, nrows, np.float64)
            if len(words):
                ww = view(_p(wl_w) + 4 * a, len(words), np.int32) if wl_w else None
                y = self._word_logits(sg, b2, T, ldt, gb, nrows, words, ww).astype(np.float64)
                m = y.max(axis=0)
                s = np.exp(y - m[None, :]).sum(axis=0)
            else:
                m = np.full(nrows, NEG)
                s = np.zeros(nrows)
            if merge:
                M = np.maximum(rm.astype(np.float64), m)
                s = rs * np.exp(rm - M) + s * np.exp(m - M)
                m = M
            rm[:] = m
            rs[:] = s
            with np.errstate(divide="ignore"):
                ls[:] = m + np.log(s)
        return 0

    def jlm_wordlist_lse_split(self, seg, t_scale, descale, b2, T, ldt, g0, cnt, cnt_idx, wl, wl_off, wl_idx, wl_base,
                               max_words, run_max, run_sum, lse, merge, beam, n_groups, stream):
        """jlm_wordlist_lse for one segment whose matrix is given as split rows"""
        sg = seg._obj if hasattr(seg, "_obj") else (seg[0] if not hasattr(seg, "k") else seg)
        if sg.k > 256 or sg.ldb % 16 or beam > 64 or max_words > 4096 - 32:
            return -2
        nv = sg.v_end - sg.v_start
        Bfull = self._split_read(sg.B, nv, sg.ldb)[:, :sg.k]
        for gb, nrows, a, words in self._groups(g0, cnt, cnt_idx, wl, wl_off, wl_idx, wl_base, n_groups, beam):
            if nrows <= 0:
                continue
            rm = view(_p(run_max) + 4 * gb, nrows, np.float32)
            rs = view(_p(run_sum) + 8 * gb, nrows, np.float64)
            ls = view(_p(lse) + 8 * gb, nrows, np.float64)
            if len(words):
                Tv = np.stack([view(_p(T) + 4 * ((gb + k) * ldt + sg.t_off), sg.k, np.float32) for k in range(nrows)])
                w = words.astype(np.int64)
                y = (Bfull[w - sg.v_start] @ (Tv.astype(np.float64) * float(t_scale)).T * float(descale)).astype(np.float32)
                y = (y + view(b2, sg.v_end, np.float32)[w][:, None]).astype(np.float64)
                m = y.max(axis=0)
                s = np.exp(y - m[None, :]).sum(axis=0)
            else:
                m = np.full(nrows, NEG)
                s = np.zeros(nrows)
            if merge:
                M = np.maximum(rm.astype(np.float64), m)
                s = rs * np.exp(rm - M) + s * np.exp(m - M)
                m = M
            rm[:] = m
            rs[:] = s
            with np.errstate(divide="ignore"):
                ls[:] = m + np.log(s)
        return 0

    def jlm_wordlist_merge_split(self, seg, t_scale, descale, b2, T, ldt, cnt, n_sent, beam, n_old_frames, wl, wl_off,
                                 wl_base, max_words, run_max, run_sum, lse, stream):
        """every older row of every sentence merges the sentence's new words (jlm_wordlist_lse merge=1 per cell)"""
        sg = seg._obj if hasattr(seg, "_obj") else (seg[0] if not hasattr(seg, "k") else seg)
        if sg.k > 256 or sg.ldb % 16 or beam > 64 or max_words > 128:
            return -2
        nv = sg.v_end - sg.v_start
        Bfull = self._split_read(sg.B, nv, sg.ldb)[:, :sg.k]
        rmax = n_sent * beam
        cntv = view(cnt, n_old_frames * n_sent, np.int32)
        offs = view(_p(wl_off) + 4 * wl_base, n_sent + 1, np.int32)
        for s in range(n_sent):
            a, b = int(offs[s]), int(offs[s + 1])
            if b <= a:
                continue
            w = view(_p(wl) + 4 * a, b - a, np.int32).astype(np.int64)
            Bw = Bfull[w - sg.v_start]
            bw = view(b2, sg.v_end, np.float32)[w]
            for fr in range(n_old_frames):
                n = min(int(cntv[fr * n_sent + s]), beam)
                if n <= 0:
                    continue
                g = fr * rmax + s * beam
                Tv = np.stack([view(_p(T) + 4 * ((g + k) * ldt + sg.t_off), sg.k, np.float32) for k in range(n)])
                y = (Bw @ (Tv.astype(np.float64) * float(t_scale)).T * float(descale)).astype(np.float32)
                y = (y + bw[:, None]).astype(np.float64)
                m = y.max(axis=0)
                sm = np.exp(y - m[None, :]).sum(axis=0)
                rm = view(_p(run_max) + 4 * g, n, np.float32)
                rs = view(_p(run_sum) + 8 * g, n, np.float64)
                M = np.maximum(rm.astype(np.float64), m)
                sm = rs * np.exp(rm - M) + sm * np.exp(m - M)
                rm[:] = M
                rs[:] = sm
                view(_p(lse) + 8 * g, n, np.float64)[:] = M + np.log(sm)
        return 0

    # ------------------------------------------------------------------- beam
    def jlm_beam_step(self, lat, st, frame, mode, max_cands, stream):
        lat, st = lat._obj if hasattr(lat, "_obj") else lat, st._obj if hasattr(st, "_obj") else st
        B, beam, F = lat.n_sent, lat.beam, lat.n_frames
        rmax = B * beam
        G = F * rmax
        slen = view(lat.sent_len, B, np.int32)
        end_off = view(lat.end_off, F * B + 1, np.int32)
        n_nodes = int(end_off[-1])
        nstart = view(lat.node_start, n_nodes, np.int32)
        nword = view(lat.node_word, n_nodes, np.int32)
        score, lse = view(st.score, G, np.float64), view(st.lse, G, np.float64)
        ysum = view(st.ysum, G, np.float64) if _p(st.ysum) else None
        bp, node, word = (view(x, G, np.int32) for x in (st.bp, st.node, st.word))
        cnt = view(st.cnt, F * B, np.int32)
        live = view(st.live, G, np.int32)
        n_live = view(st.n_live, F, np.int32)
        edge = view(st.edge, max(n_nodes, 1) * beam, np.float32)
        live_base = view(st.live_base, F * B, np.int32) if _p(st.live_base) else None
        fused = _p(st.lse_part) != 0
        if fused:
            if live_base is None or st.n_parts < 1 or mode != 0:
                return -1
            pv = view(st.lse_part, st.n_parts * st.ld_part * 2, np.float32).reshape(st.n_parts, st.ld_part, 2).astype(np.float64)
        for s in range(B):
            fs = frame * B + s
            if frame > slen[s]:
                cnt[fs] = 0
                continue
            if fused and frame >= 1:           # fold the vocabulary partials of this sentence's rows of frame - 1
                fp = (frame - 1) * B + s
                for r in range(int(cnt[fp])):
                    q = pv[:, int(live_base[fp]) + r]
                    mx = q[:, 0].max()
                    lse[(frame - 1) * rmax + s * beam + r] = mx + np.log((q[:, 1] * np.exp(q[:, 0] - mx)).sum())
            nb, ne = int(end_off[fs]), int(end_off[fs + 1])
            gout = frame * rmax + s * beam
            if frame == 0:
                K = 1
                score[gout] = 0.0
                if ysum is not None:
                    ysum[gout] = 0.0
                bp[gout], node[gout], word[gout] = -1, nb, nword[nb]
            else:
                assert (ne - nb) * beam <= max_cands
                S = {}
                if mode == 2:
                    for f in range(frame):
                        for k in range(int(cnt[f * B + s])):
                            g = f * rmax + s * beam + k
                            p = int(bp[g])
                            S[g] = 0.0 if p < 0 else S[p] + lse[p]
                cands = []
                for n in range(nb, ne):
                    sf = int(nstart[n])
                    for k in range(int(cnt[sf * B + s])):
                        gp = sf * rmax + s * beam + k
                        e = float(edge[n * beam + k])
                        if mode == 0:
                            sc = score[gp] + (lse[gp] - e)
                        elif mode == 1:
                            sc = score[gp] - e
                        else:
                            sc = (S[gp] + lse[gp]) - (ysum[gp] + e)
                        cands.append((sc, (n - nb) * beam + k, n, gp, e))
                cands.sort(key=lambda c: (c[0], c[1]))
                K = min(beam, len(cands))
                for r in range(K):
                    sc, _c, n, gp, e = cands[r]
                    score[gout + r] = sc
                    if mode == 2:
                        ysum[gout + r] = ysum[gp] + e
                    bp[gout + r], node[gout + r], word[gout + r] = gp, n, nword[n]
            cnt[fs] = K
            if frame < slen[s]:
                base = int(n_live[frame])
                if live_base is not None:
                    live_base[fs] = base
                n_live[frame] = base + K
                live[frame * rmax + base: frame * rmax + base + K] = np.arange(gout, gout + K)
        return 0

    def jlm_backtrace(self, lat, st, out_nodes, out_len, out_score, stride, stream):
        lat, st = lat._obj if hasattr(lat, "_obj") else lat, st._obj if hasattr(st, "_obj") else st
        B, beam, F = lat.n_sent, lat.beam, lat.n_frames
        rmax = B * beam
        G = F * rmax
        slen = view(lat.sent_len, B, np.int32)
        score = view(st.score, G, np.float64)
        bp, node = view(st.bp, G, np.int32), view(st.node, G, np.int32)
        cnt = view(st.cnt, F * B, np.int32)
        on = view(out_nodes, rmax * stride, np.int32).reshape(rmax, stride)
        ol, osc = view(out_len, rmax, np.int32), view(out_score, rmax, np.float64)
        for idx in range(rmax):
            s, r = divmod(idx, beam)
            L = int(slen[s])
            if r >= cnt[L * B + s]:
                ol[idx], osc[idx] = 0, 0.0
                continue
            g = L * rmax + s * beam + r
            osc[idx] = score[g]
            d = 0
            while g >= 0 and d < stride:
                on[idx, d] = node[g]
                d += 1
                g = int(bp[g])
            ol[idx] = d
        return 0

    def jlm_softmax_rows(self, y, pred, ld, n_rows, n_cols, self_norm, stream):
        yv = view(y, n_rows * ld, np.float32).reshape(n_rows, ld)[:, :n_cols]
        pv = view(pred, n_rows * ld, np.float32).reshape(n_rows, ld)
        if self_norm:
            pv[:, :n_cols] = np.exp(yv)
        else:
            e = np.exp(yv - yv.max(axis=1, keepdims=True))
            pv[:, :n_cols] = e / e.sum(axis=1, keepdims=True)
        return 0


# ---------------------------------------------------------------------------------------------------------------
# The package talks to torch custom ops (jlm_amd/ops.py -> csrc/jlm_torch_ops.cpp).  FakeOps is the CPU double of THAT
# interface: it does what the C++ shim does -- turn tensor arguments into the C structs and pointers of include/jlm_hip.h
# -- and hands them to the numpy double of the C ABI above, so the host code is driven through the same two layers.
class _FakeModel:
    def __init__(self, t, i, f, seg_B, seg_meta, split_B, split_meta, t_scale, descale, bias_col, mixed_idx=(), mixed_B=(), mixed_meta=(),
                 mixed_t_scale=(), mixed_descale=(), mixed_s8=(), mixed_head_split=()):
        from jlm_amd import _lib
        self.keep = (t, list(seg_B), list(split_B), list(mixed_B))

        def segs(B, meta):
            arr = (_lib.Segment * max(len(B), 1))()
            for k, b in enumerate(B):
                arr[k] = _lib.Segment(*[int(x) for x in meta[5 * k:5 * k + 4]], b.data_ptr(), int(meta[5 * k + 4]))
            return arr
        self.segs, self.split = segs(seg_B, seg_meta), segs(split_B, split_meta)
        self.ts = (ctypes.c_float * max(len(t_scale), 1))(*t_scale)
        self.ds = (ctypes.c_float * max(len(descale), 1))(*descale)
        self.bc = (ctypes.c_int * max(len(bias_col), 1))(*bias_col)
        ptr = lambda k: t[k].data_ptr() if k in t else None
        m = self.m = _lib.DecodeModel()
        m.segs, m.n_segs, m.b2 = self.segs, len(seg_B), ptr("b2")
        m.H, m.ldt = i["H"], i["ldt"]
        m.untied, m.self_norm, m.split_lstm = i.get("untied", 0), i.get("self_norm", 0), i.get("split_lstm", 0)
        m.emb, m.ld_emb, m.wt, m.gate_bias = ptr("emb"), i.get("ld_emb", 0), ptr("wt"), ptr("gate_bias")
        m.kpad, m.E = i.get("kpad", 0), i.get("E", 0)
        m.gate_descale, m.h_scale, m.t_descale = f.get("gate_descale", 0.0), f.get("h_scale", 0.0), f.get("t_descale", 0.0)
        m.wt8, m.xgate8, m.pmt, m.pmt_split, m.n_t = ptr("wt8"), ptr("xgate8"), ptr("pmt"), ptr("pmt_split"), i.get("n_t", 0)
        m.untied_split, m.untied_descale = ptr("untied_split"), f.get("untied_descale", 0.0)
        if len(split_B):
            m.split_segs = self.split
            m.split_t_scale = ctypes.cast(self.ts, ctypes.POINTER(ctypes.c_float))
            m.split_descale = ctypes.cast(self.ds, ctypes.POINTER(ctypes.c_float))
            m.split_bias_col = ctypes.cast(self.bc, ctypes.POINTER(ctypes.c_int))
        if len(mixed_idx):
            assert len(split_B) and len(mixed_B) == len(mixed_idx) == len(mixed_t_scale) == len(mixed_descale) == len(mixed_s8)
            n = len(seg_B)
            some = segs(mixed_B, mixed_meta)
            self.mixed = (_lib.Segment * n)()
            self.mts, self.mds, self.ms8 = (ctypes.c_float * n)(), (ctypes.c_float * n)(), (ctypes.c_float * n)()
            for j, si in enumerate(mixed_idx):
                assert (some[j].v_start, some[j].v_end, some[j].k, some[j].t_off) == \
                    (self.split[si].v_start, self.split[si].v_end, self.split[si].k, self.split[si].t_off)
                self.mixed[si] = some[j]
                self.mts[si], self.mds[si], self.ms8[si] = mixed_t_scale[j], mixed_descale[j], mixed_s8[j]
            m.mixed_segs = self.mixed
            m.mixed_t_scale = ctypes.cast(self.mts, ctypes.POINTER(ctypes.c_float))
            m.mixed_descale = ctypes.cast(self.mds, ctypes.POINTER(ctypes.c_float))
            m.mixed_s8 = ctypes.cast(self.ms8, ctypes.POINTER(ctypes.c_float))
            m.mixed_bias2 = ptr("b2_log2")
            self.mhs = (ctypes.c_int * n)()
            for j, si in enumerate(mixed_idx):
                c = int(mixed_head_split[j]) if len(mixed_head_split) else 0
                assert c >= 0 and c % 128 == 0 and c < some[j].v_end - some[j].v_start
                self.mhs[si] = c
            m.mixed_head_split = ctypes.cast(self.mhs, ctypes.POINTER(ctypes.c_int))


class _FakePlan:
    def __init__(self, t, i):
        from jlm_amd import _lib
        self.keep = t
        base = t["ints"].data_ptr()
        at = lambda name: base + 4 * i["off_" + name]
        ptr = lambda k: t[k].data_ptr() if k in t else None
        self.frames = i["frames"]
        self.lat = _lib.Lattice(i["n_sent"], i["beam"], 0, at("sent_len"), at("end_off"), at("node_start"), at("node_word"))
        self.st = _lib.BeamState(ptr("score"), ptr("lse"), ptr("ysum"), ptr("bp"), ptr("node"), ptr("word"), ptr("cnt"),
                                 ptr("live"), ptr("n_live"), ptr("edge"), ptr("live_base"), None, 0, 0)
        d = self.p = _lib.DecodePlan()
        d.kind, d.max_cands = i["kind"], i["max_cands"]
        d.h, d.c, d.T = ptr("h"), ptr("c"), ptr("T")
        d.g0, d.cidx, d.sidx = at("g0"), at("cidx"), at("sidx")
        d.sg_word, d.sg_off, d.sg_node, d.edge = at("sg_word"), at("sg_off"), at("sg_node"), ptr("edge")
        d.vs_words, d.vs_off = at("vs_words"), at("vs_off")
        d.di_words, d.di_off, d.di_idx = at("di_words"), at("di_off"), at("sidx2")
        d.dd_words, d.dd_off = at("dd_words"), at("dd_off")
        d.di_wwords = at("di_wwords") if "off_di_wwords" in i else None
        d.sg_wword = at("sg_wword") if "off_sg_wword" in i else None
        d.run_max, d.run_sum, d.part, d.max_parts = ptr("run_max"), ptr("run_sum"), ptr("part"), i["max_parts"]
        d.out_nodes, d.out_len, d.out_score, d.stride = ptr("out_nodes"), ptr("out_len"), ptr("out_score"), i["stride"]
        d.Tm, d.ld_tm = ptr("Tm"), i.get("ld_tm", 0)
        assert not d.Tm or t["Tm"].numel() >= i["n_sent"] * i["beam"] * d.ld_tm
        self.timed_frames = 0


class FakeOps:
    """CPU double of jlm_amd.ops.HipOps (same methods, tensors in)."""

    def __init__(self, lib=None):
        self.lib = lib or FakeLib()
        self.Model, self.Plan = _FakeModel, _FakePlan

    @staticmethod
    def _chk(rc, what):
        if rc != 0:
            raise RuntimeError("%s failed with code %d" % (what, rc))

    @staticmethod
    def _o(t, off=0):
        return None if t is None else t.data_ptr() + 4 * int(off)

    def abi_version(self):
        return self.lib.jlm_abi_version()

    def beam_step_max_cands(self, beam, n_frames, mode):
        return self.lib.jlm_beam_step_max_cands(int(beam), int(n_frames), int(mode))

    def decode_frames(self, model, plan, n_frames, vs_max, di_max, dd_max, use_side, timed, lse_cu_share_pct=0):
        assert 1 <= n_frames <= plan.frames
        plan.lat.n_frames = int(n_frames)
        plan.p.vs_max, plan.p.di_max, plan.p.dd_max = int(vs_max), int(di_max), int(dd_max)
        plan.timed_frames = int(n_frames) if timed else 0
        rc = self.lib.jlm_decode_frames(model.m, plan.p, plan.lat, plan.st, 0, None, None)
        if rc == -2:
            return -2
        self._chk(rc, "jlm_decode_frames")
        return 0

    def frame_times(self, plan):
        import torch
        return torch.full((plan.timed_frames, 5), 1e-3, dtype=torch.float64)

    def lse_probe(self, model, rowlist, prev, word, steps, rows, h, c, T, Tm, ld_tm, form, part, max_parts):
        o = self._o
        rc = self.lib.jlm_lse_probe(model.m, o(rowlist), o(prev), o(word), int(steps), int(rows), o(h), o(c), o(T), o(Tm), int(ld_tm),
                                    int(form), o(part), int(max_parts), 0)
        if rc == -2 or rc >= 1:
            return rc
        self._chk(rc if rc else -1, "jlm_lse_probe")

    def lstm_step(self, h_in, c_in, ld, h_out, c_out, rows, prev, word, emb, ld_emb, wt, bias, kpad, H, E, n_rows_max, n_dev):
        o = self._o
        self._chk(self.lib.jlm_lstm_step(o(h_in), o(c_in), ld, o(h_out), o(c_out), o(rows), o(prev), o(word), o(emb), ld_emb, o(wt),
                                         o(bias), kpad, H, E, n_rows_max, o(n_dev), 0), "jlm_lstm_step")

    def gemm_nt(self, A, a_off, lda, a_rows, B, ldb, b_rows, C, c_off, ldc, c_rows, bias, bias_off, M, N, K, m_dev):
        o = self._o
        self._chk(self.lib.jlm_gemm_nt(o(A, a_off), lda, o(a_rows), o(B), ldb, o(b_rows), o(C, c_off), ldc, o(c_rows),
                                       o(bias, bias_off), M, N, K, o(m_dev), 0), "jlm_gemm_nt")

    def softmax_rows(self, y, pred, ld, n_rows, n_cols, self_norm):
        self._chk(self.lib.jlm_softmax_rows(self._o(y), self._o(pred), ld, n_rows, n_cols, int(bool(self_norm)), 0), "jlm_softmax_rows")

    def pack_split_f16(self, src, src_off, rows, k, ld, scale, dst, dst_off, ld_dst):
        self._chk(self.lib.jlm_pack_split_f16(self._o(src, src_off), rows, k, ld, scale, self._o(dst, dst_off), ld_dst, 0),
                  "jlm_pack_split_f16")

    def pack_mixed(self, src, src_off, rows, k, ld, bias, bias_off, scale, bias_scale, s8, dst, ld_dst):
        self._chk(self.lib.jlm_pack_mixed(self._o(src, src_off), rows, k, ld, self._o(bias, bias_off), scale, bias_scale, s8, self._o(dst),
                                          ld_dst, 0), "jlm_pack_mixed")

    def dequant_u8(self, code, rows, k, ld_code, codebook, dst, ld_dst):
        self._chk(self.lib.jlm_dequant_u8(code.data_ptr(), rows, k, ld_code, codebook.data_ptr(), codebook.numel(), dst.data_ptr(),
                                          ld_dst, 0), "jlm_dequant_u8")

    def pack_split_f16_col(self, v, v_off, rows, scale, dst, ld_dst, col):
        self._chk(self.lib.jlm_pack_split_f16_col(self._o(v, v_off), rows, scale, self._o(dst), ld_dst, col, 0),
                  "jlm_pack_split_f16_col")


def install(monkeypatch):
    """Route jlm_amd's op backend, library handle and device checks to the CPU doubles."""
    import torch
    from jlm_amd import _lib, ops

    fake = FakeLib()
    monkeypatch.setattr(_lib, "lib", lambda: fake)
    monkeypatch.setattr(_lib, "require_gpu", lambda: torch.device("cpu"))
    monkeypatch.setattr(ops, "_backend", FakeOps(fake))
    return fake


def install_plain():
    """Same as install() for processes without a pytest monkeypatch (spawned ranks)."""
    import torch
    from jlm_amd import _lib, ops

    fake = FakeLib()
    _lib.lib = lambda: fake
    _lib.require_gpu = lambda: torch.device("cpu")
    ops.set_backend(FakeOps(fake))
    return fake
