"""Parity evidence on the GPU box: every golden case (reference outputs captured by
tools/make_golden.py) replayed through the HIP path; prints one line per case.
    python tools/parity_report.py > profiles/rNN_parity.txt"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from jlm_amd import config as jconfig           # noqa: E402
from tests import conftest, golden_cases as gc  # noqa: E402

gold = json.load(open(os.path.join(REPO, "tests", "golden", "decode.json"), encoding="utf-8"))
glm = np.load(os.path.join(REPO, "tests", "golden", "lm_steps.npz"))


def main():
    from jlm_amd.decoder import Decoder
    from jlm_amd.decoder_dynamic import DynamicDecoder
    from jlm_amd.model import LSTM_Model
    print("# step logits vs the reference (3 LSTM steps, R=10): max |dy| / max|y| per row, max over rows")
    for name in gc.LM_FIXTURES:
        f = conftest.fixture_root(name)
        jconfig.set_root(f["root"])
        lm = LSTM_Model(1)
        for kind in ("full", "subset"):
            if kind == "subset" and not f["cfg"]["share_embedding"]:
                continue
            idx, subset, cols, h0, c0 = gc.lm_inputs(f["cfg"], 10)
            h, c = h0.copy(), c0.copy()
            for st in range(gc.LM_STEPS):
                (pred, y, _a, _b), h, c = lm.predict_with_context(idx[st], h, c, subset if kind == "subset" else None)
            key = "%s/%s/R10" % (name, kind)
            ysel = y if kind == "subset" else y[:, cols]
            yref = glm[key + "/y"]
            rel = (np.abs(ysel - yref) / np.abs(yref).max(axis=1, keepdims=True)).max()
            print("%-28s logits rel err %.2e   hidden max abs err %.2e" % (key, rel, np.abs(h - glm[key + "/h"]).max()))
    print("# decodes vs the reference: sentences, identical 1-best, identical full n-best, max |score diff|")
    for name, fixture, kind, kwargs, spec in gc.DECODE_CASES:
        f = conftest.fixture_root(fixture)
        jconfig.set_root(f["root"])
        dec = (DynamicDecoder if kind == "dynamic" else Decoder)(1)
        dec.perf_timing = False
        dec.compat_quirks = gc.is_quirk_case(name)
        sents = gc.case_sentences(spec, f["alphabet"])
        if kwargs.get("random_sampling"):
            outs = []
            for si, s in enumerate(sents):
                np.random.seed(gc.RANDOM_SAMPLING_SEED + si)
                outs.append(dec.decode(s, **kwargs))
        else:
            outs = dec.decode_batch(sents, **kwargs)
        best = full = 0
        md = 0.0
        for o, g in zip(outs, gold[name]):
            g = g["nbest"]
            best += o[0][1] == g[0][1]
            full += [w for _, w in o] == [w for _, w in g]
            md = max(md, max(abs(a[0] - b[0]) for a, b in zip(o, g)))
        m = dec.model.dev
        form = "f32" if getattr(m, "split_array", None) is None else ("mixed" if len(getattr(m, "mixed_idx", [])) == m.n_segs else
                                                                      "hybrid" if getattr(m, "mixed_idx", []) else "split")
        cal = getattr(m, "mixed_calib", None)
        print("%-28s n=%3d  1-best %3d/%3d  n-best %3d/%3d  max score diff %.2e  normaliser on %s rows%s" % (
            name, len(sents), best, len(sents), full, len(sents), md, form,
            "" if cal is None else " (load-time calibration: lse rms diff %.1e, limit %.0e)" % (cal["lse_rms_diff"], cal["limit"])))


if __name__ == "__main__":
    import contextlib
    import io
    buf = io.StringIO()
    real = sys.stdout
    with contextlib.redirect_stdout(buf):
        sys.stdout = buf
        try:
            main()
        finally:
            sys.stdout = real
    print("\n".join(l for l in buf.getvalue().splitlines() if not l.startswith(("LSTM model", "Dynamic RNN", "vocab with"))))
