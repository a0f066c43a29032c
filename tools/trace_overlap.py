"""GPU occupancy of a rocprofv3 kernel trace (run_kernel_trace.csv): over a window of decode steps, the time at
least one kernel was running, the time at least two were, and per-queue totals -- whether the two batches in
flight really run side by side.  usage: trace_overlap.py TRACE.csv [first_lse_index] [n_lse]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows]
ev.sort()
lse = [i for i, e in enumerate(ev) if e[2].startswith("vocab_lse")]
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(lse) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 190
a, b = lse[first], lse[min(first + n, len(lse) - 1)]
win = ev[a:b]
t0, t1 = win[0][0], max(e[1] for e in win)
pts = []
for s, e, _n, _q in win:
    pts += [(s, 1), (e, -1)]
pts.sort()
busy1 = busy2 = 0
depth, last = 0, t0
for t, d in pts:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
span = t1 - t0
print("window: %d kernels, %.2f ms, %d vocab_lse launches" % (len(win), span / 1e6, sum(1 for e in win if e[2].startswith("vocab_lse"))))
print("  >=1 kernel running %.1f %%   >=2 running %.1f %%   idle %.1f %%" % (100 * busy1 / span, 100 * busy2 / span, 100 * (span - busy1) / span))
perq = collections.defaultdict(int); pern = collections.defaultdict(int)
for s, e, nm, q in win:
    perq[q] += e - s; pern[nm.split("(")[0][:48]] += e - s
print("  sum of kernel durations %.2f ms = %.2f x the window" % (sum(perq.values()) / 1e6, sum(perq.values()) / span))
for q, v in sorted(perq.items()): print("  queue %s: %.2f ms" % (q, v / 1e6))
for nm, v in sorted(pern.items(), key=lambda kv: -kv[1])[:6]: print("  %-50s %.2f ms" % (nm, v / 1e6))
