"""Round 5: what a calibration figure means for path scores.  The golden sentences of a trained-model-like fixture are decoded with the
normaliser FORCED into each form (every segment mixed; a head of the first segment on split rows; the first segment on split rows; split
rows) and the worst |score - reference score| is printed beside the form's load-time rms / max -- the table behind the loader's acceptance
rules (DeviceModel._calibrate_mixed).  python tools/probes/form_vs_score.py [fixture/case ...]"""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from jlm_amd import config as jconfig, model as jmodel            # noqa: E402
from tests import conftest, golden_cases as gc                    # noqa: E402
gold = json.load(open(os.path.join(REPO, "tests", "golden", "decode.json"), encoding="utf-8"))
want = sys.argv[1:] or ["peaked20-vtable/static", "peaked-vtable/static-L40", "peaked-vtable/static"]
DM = jmodel.DeviceModel
orig = DM._calibrate_mixed


def forced(form):
    """a _calibrate_mixed that measures as usual, then adopts `form` whatever the figures say"""
    def cal(self):
        if form == "split":
            os.environ["JLM_MIXED_MAX_LSE_RMS"] = "1e-30"
        elif form == "mixed":
            os.environ["JLM_MIXED_MAX_LSE_RMS"] = "1.0"
        elif form == "first-split":
            os.environ["JLM_MIXED_MAX_LSE_RMS"] = "1e-6"; DM.HEAD_SPLITS = (1 << 30,)
        orig(self)
    return cal


for case in want:
    name, fixture, kind, kwargs, spec = next(c for c in gc.DECODE_CASES if c[0] == case)
    f = conftest.fixture_root(fixture)
    jconfig.set_root(f["root"])
    sents = gc.case_sentences(spec, f["alphabet"])
    from jlm_amd.decoder import Decoder
    for form in ("mixed", "2048", "8192", "first-split", "split"):
        saved = (os.environ.get("JLM_MIXED_MAX_LSE_RMS"), DM.HEAD_SPLITS)
        if form.isdigit():
            # simpler and exact: build the model with calibration off, then set the head by hand
            os.environ["JLM_MIXED_MAX_LSE_RMS"] = "0"
            dec = Decoder(1)
            m = dec.model.dev
            m.mixed_head_split = [int(form)] + [0] * (len(m.mixed_idx) - 1)
            m.lse_fixed_ref, m._decode_model = 0, None
            calib = "head %s (forced)" % form
        else:
            DM._calibrate_mixed = forced(form)
            try:
                dec = Decoder(1)
            finally:
                DM._calibrate_mixed = orig
            m = dec.model.dev
            c = m.mixed_calib or {}
            calib = "rms %.2e max %.2e kept %s %s" % (c.get("lse_rms_diff", float("nan")), c.get("lse_max_diff", float("nan")), c.get("kept"),
                                                     "first segment split" if c.get("split_segments") else "")
        if saved[0] is None:
            os.environ.pop("JLM_MIXED_MAX_LSE_RMS", None)
        else:
            os.environ["JLM_MIXED_MAX_LSE_RMS"] = saved[0]
        DM.HEAD_SPLITS = saved[1]
        dec.perf_timing = False
        outs = dec.decode_batch(sents, **kwargs)
        best, md, sq = 0, 0.0, []
        for o, g in zip(outs, gold[name]):
            g = g["nbest"]
            best += o[0][1] == g[0][1]
            d = [abs(a[0] - b[0]) for a, b in zip(o, g)]
            md = max(md, max(d)); sq += d
        print("%-28s form %-12s mixed segments %s heads %s: 1-best %d/%d  max |score diff| %.2e  rms %.2e   [%s]" % (
            name, form, list(m.mixed_idx), list(getattr(m, "mixed_head_split", [])), best, len(sents), md, float(np.sqrt(np.mean(np.square(sq)))), calib), flush=True)
        del dec
