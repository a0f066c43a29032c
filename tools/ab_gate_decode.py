"""Interleaved in-process A/B of the LSTM-step variants inside the pipelined decode (device-resident loop, configs[1]):
JLM_GATE_V is read once per process by the launcher, so the two variants run as alternating child processes.
usage: python tools/ab_gate_decode.py [rounds]"""
import json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for i in range(rounds):
    for v in ("0", "1"):
        env = dict(os.environ, JLM_GATE_V=v)
        out = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--steps", "40", "--warmup", "3", "--no-config5", "--no-legs",
                              "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
        d = json.loads(out)
        print("JLM_GATE_V=%s  value %.0f  %.3f ms/step  device-resident %.3f ms/step  gate %.1f us (%.1f%%)  lse %.1f us" % (
            v, d["value"], d["ms_per_step"], d["device_resident_ms_per_step"], d["gate_gemm"]["avg_launch_ms"] * 1e3,
            d["gate_gemm"]["mfma_util_pct"], d["roofline"]["avg_launch_ms"] * 1e3), flush=True)
