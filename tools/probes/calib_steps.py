"""Round 5 (verdict P4): is the load-time calibration's answer a property of the model, or of the three steps the probe rows take from the zero state?
Each fixture is loaded with CALIB_STEPS = 3 (the default), 8, 20 and 40 (a 40-kana decode's depth) and two seeds; printed: the log-normaliser rms
difference (mixed rows vs split rows), its maximum, |log Z| in bits (the fixed-reference gate), and what the loader decided.
python tools/probes/calib_steps.py [fixture ...]"""
import os, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, jlm_amd
from jlm_amd import config as jconfig, synth, model as jmodel
names = sys.argv[1:] or ["mid-vtable", "peaked-vtable", "peaked20-vtable", "heavy-vtable", "mid-tied", "peaked-tied", "peaked20-tied", "mid-untied"]
for name in names:
    root = os.path.join(tempfile.gettempdir(), "jlm_calib_" + name)
    try:
        synth.build_fixture(root, name)
    except Exception as e:
        print(name, "no such fixture:", e); continue
    jconfig.set_root(root)
    for steps in (3, 8, 20, 40):
        for seed in (20240929, 7):
            jmodel.DeviceModel.CALIB_STEPS, jmodel.DeviceModel.CALIB_SEED = steps, seed
            m = jmodel.LSTM_Model(1)
            c = m.dev.mixed_calib
            if c is None:
                print("%-16s steps %2d seed %8d: no calibration (no mixed rows built)" % (name, steps, seed)); continue
            print("%-16s steps %2d seed %8d: rms %.3e max %.3e |logZ| %.1f bits kept %s fixed_ref %s%s" % (
                name, steps, seed, c.get("lse_rms_diff", float("nan")), c.get("lse_max_diff", float("nan")), c.get("lse_abs_max_bits", float("nan")),
                c.get("kept"), c.get("fixed_ref"), (("  head_split %s%s" % (c["head_split"], " (first segment)" if c.get("split_segments") else "")) if c.get("head_split") else "") +
                (("  (" + c["reason"] + ")") if c.get("reason") else "")))
            del m; torch.cuda.empty_cache()
