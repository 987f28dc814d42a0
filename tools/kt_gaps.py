"""Gaps between consecutive kernels of a rocprofv3 kernel trace (the stream's idle time per iteration): prints, per kernel name,
the average time the GPU sat idle BEFORE a launch of that kernel.  usage: python tools/kt_gaps.py <..._kernel_trace.csv>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gap = collections.defaultdict(float); cnt = collections.Counter(); dur = collections.defaultdict(float)
prev_end = None
for r in rows:
    n = r["Kernel_Name"]; n = n.split("gsr::")[1].split("(")[0][:28] if "gsr::" in n else n[:28]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None: gap[n] += max(0, s - prev_end); cnt[n] += 1
    dur[n] += e - s
    prev_end = max(prev_end or 0, e)
for n in sorted(cnt, key=lambda k: -gap[k]):
    if cnt[n] >= 5: print("%-30s launches %5d  idle before: %7.1f us avg   kernel: %7.1f us avg" % (n, cnt[n], gap[n] / cnt[n] / 1e3, dur[n] / cnt[n] / 1e3))
