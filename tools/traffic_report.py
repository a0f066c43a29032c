"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/gpu_traffic.sh) -> per-kernel HBM bytes per launch.
FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies 128-B requests at 64 B for wide coalesced reads).
usage: traffic_report.py <out_csv> [<traffic_latest.json>]"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
R = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), ".."))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(R, "gpurun_out", "traffic", c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]][c].append(float(r["Counter_Value"]))
rows = []
for k, v in agg.items():
    f, w = v.get("FETCH_SIZE", []), v.get("WRITE_SIZE", [])
    fm, wm = (sum(f) / len(f) if f else 0.0), (sum(w) / len(w) if w else 0.0)
    rows.append((k, max(len(f), len(w)), fm, fm * 1024 * 2, wm, fm * 1024 * 2 + wm * 1024))
rows.sort(key=lambda r: -r[5] * r[1])
with open(sys.argv[1], "w") as fo:
    fo.write("Kernel,Launches,FETCH_SIZE_KB_raw_mean,FETCH_bytes_corrected_x2,WRITE_SIZE_KB_raw_mean,HBM_bytes_per_launch_corrected\n")
    for k, n, fm, fb, wm, tot in rows[:14]:
        fo.write('"%s",%d,%.1f,%d,%.1f,%d\n' % (k, n, fm, fb, wm, tot))
        print("%-70s n=%4d  %.1f MB per launch" % (k[:70], n, tot / 1e6))
if len(sys.argv) > 2:
    # the headline workload's kernel (mid-vtable: the D-softmax* shapes), not the tied legs' instantiation of the same template
    lse = ([r for r in rows if "vocab_lse_mx6" in r[0] and "7, 13" in r[0]] or [r for r in rows if "vocab_lse_mx6" in r[0]] or
           [r for r in rows if "vocab_lse_mixed" in r[0] and "7, 13" in r[0]] or [r for r in rows if "vocab_lse_mixed" in r[0]] or
           [r for r in rows if "vocab_lse_split" in r[0] or "vocab_lse_hybrid" in r[0]])
    if lse:
        import bench
        gate = [r for r in rows if "gate_xg" in r[0]]
        json.dump({"kernel": lse[0][0].strip(), "vocab_lse_hbm_bytes_per_call": int(lse[0][5]),
                   "gate_kernel": gate[0][0].strip() if gate else None, "gate_hbm_bytes_per_call": int(gate[0][5]) if gate else None,
                   "gate_source_sha256": bench.gate_source_sha256(),
                   "fixture": "mid-vtable", "source_sha256": bench.kernel_source_sha256(),
                   "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/gpu_traffic.sh), FETCH_SIZE doubled as "
                           "MI355X_MICROARCH.md prescribes for gfx950 wide coalesced reads; " + os.path.basename(sys.argv[1])},
                  open(sys.argv[2], "w"), indent=1)
