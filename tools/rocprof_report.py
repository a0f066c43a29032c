"""rocprofv3 --kernel-trace --stats summary of a bench.py run (tools/gpu_prof.sh) -> profiles-style JSON with the dominant kernels'
average durations, tagged with the kernel sources' hash so that bench.py only quotes it for the build it was measured on.
usage: rocprof_report.py <run_kernel_stats.csv> <rocprof_latest.json>"""
import csv, json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import bench
rows = list(csv.DictReader(open(sys.argv[1])))
def avg_us(pred):
    r = [x for x in rows if pred(x["Name"])]
    return (float(r[0]["AverageNs"]) / 1e3, int(r[0]["Calls"]), r[0]["Name"][:120]) if r else (None, 0, None)
lse = avg_us(lambda n: "vocab_lse_mx6" in n and "7, 13" in n)          # round 6: the mx6 form is the default of the headline model
if lse[0] is None:
    lse = avg_us(lambda n: "vocab_lse_mixed" in n and "7, 13" in n)
if lse[0] is None:
    lse = avg_us(lambda n: "vocab_lse_mx6" in n or "vocab_lse_mixed" in n or "vocab_lse_split" in n or "vocab_lse_hybrid" in n)
gate = avg_us(lambda n: "gate_xg" in n)
json.dump({"vocab_lse_kernel": lse[2], "vocab_lse_avg_us": lse[0], "vocab_lse_calls": lse[1],
           "gate_kernel": gate[2], "gate_avg_us": gate[0], "gate_calls": gate[1],
           "fixture": "mid-vtable", "source_sha256": bench.kernel_source_sha256(), "gate_source_sha256": bench.gate_source_sha256(),
           "note": "rocprofv3 --kernel-trace --stats of `python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-legs --no-config5` "
                   "(tools/gpu_prof.sh; the profiler serialises the streams): AverageNs of the kernel; " + os.path.basename(sys.argv[1])},
          open(sys.argv[2], "w"), indent=1)
print(open(sys.argv[2]).read())
