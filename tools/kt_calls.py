"""Takes a rocprofv3 kernel trace of tools/call_timeline.py apart: one refine() call = the kernels from one k_refine_init to the
next.  Prints, for the median call, when each kernel starts / how long it runs, and per call: wall (first start to last end),
sum of kernel time, idle time.  usage: python tools/kt_calls.py <..._kernel_trace.csv> [K]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(n): return n.split("gsr::")[1].split("(")[0][:34] if "gsr::" in n else n[:34]
calls, cur = [], None
for r in rows:
    n = nm(r["Kernel_Name"])
    if n.startswith("k_refine_init"):
        if cur: calls.append(cur)
        cur = []
    if cur is not None: cur.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
if cur: calls.append(cur)
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
calls = [c for c in calls if sum(1 for k in c if k[0].startswith("k_render_bwd")) == K]
print(len(calls), "calls of", K, "iterations")
stats = []
for c in calls:
    wall = c[-1][2] - c[0][1]; busy = sum(e - s for _, s, e in c)
    stats.append((wall, busy, c))
stats.sort(key=lambda x: x[0])
wall, busy, c = stats[len(stats) // 2]
print("median call: wall %.1f us, kernels %.1f us, idle %.1f us, %d launches" % (wall / 1e3, busy / 1e3, (wall - busy) / 1e3, len(c)))
t0 = c[0][1]; prev = t0
for i, (n, s, e) in enumerate(c):
    if i < 14 or i >= len(c) - 10: print("  %8.1f us  +gap %6.1f  %-34s %7.1f us" % ((s - t0) / 1e3, (s - prev) / 1e3, n, (e - s) / 1e3))
    elif i == 14: print("  ...")
    prev = e
