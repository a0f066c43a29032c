"""Round 6: what the decoder-level calibration probe (hypotheses kept by real decodes, DeviceModel.calibrate_on_paths) says beside the
seeded word draws, per gate fixture: form kept, worst-probe figure, every probe's rms.  usage: calib_paths.py [fixture ...]"""
import os, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
for name in sys.argv[1:] or ["mid-vtable", "mid-tied", "peaked-vtable", "peaked-tied", "heavy-vtable", "peaked20-vtable", "peaked20-tied"]:
    root = os.path.join(tempfile.gettempdir(), "jlm_calib_" + name)
    synth.build_fixture(root, name)
    jconfig.set_root(root)
    m = Decoder(1).model.dev
    c = m.mixed_calib or {}
    print("%-16s fmt %-5s idx %-10s worst rms %.2e margin %5.2f  probes %s%s" % (
        name, m.mixed_fmt, m.mixed_idx, c.get("lse_rms_diff", float("nan")), c.get("margin", float("nan")),
        " ".join("%s:%.2e" % (p["kind"][:4], p["rms"]) for p in c.get("probes", [])),
        ("  | mx6 refused at %.2e" % c["mx6"]["lse_rms_diff"]) if "mx6" in c else ""), flush=True)
