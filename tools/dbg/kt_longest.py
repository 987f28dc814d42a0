"""rocprofv3 --kernel-trace CSV -> the launches of the library's kernels in time order around the longest ones (which launch of a
refinement call is the expensive one?).  usage: python tools/dbg/kt_longest.py <kernel_trace.csv> [substring] [top]"""
import csv, sys
csv.field_size_limit(1 << 30)
path, sub, top = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_preprocess"), int(sys.argv[3]) if len(sys.argv) > 3 else 12
rows = []
for r in csv.DictReader(open(path)):
    n = r.get("Kernel_Name") or r.get("kernel_name") or ""
    if "gsr::" not in n:
        continue
    s, e = int(r.get("Start_Timestamp") or r.get("start_timestamp")), int(r.get("End_Timestamp") or r.get("end_timestamp"))
    g = r.get("Grid_Size") or r.get("grid_size") or r.get("Grid_Size_X") or ""
    rows.append((s, e, n.split("gsr::")[1].split("(")[0][:48], g))
rows.sort()
idx = sorted([i for i, r in enumerate(rows) if sub in r[2]], key=lambda i: rows[i][0] - rows[i][1])[:top]
for i in sorted(idx):
    lo, hi = max(0, i - 3), min(len(rows), i + 3)
    print("--- launch %d: %s %.1f us (grid %s)" % (i, rows[i][2], (rows[i][1] - rows[i][0]) / 1e3, rows[i][3]))
    print("    context:", " | ".join("%s %.0f" % (rows[j][2][:22], (rows[j][1] - rows[j][0]) / 1e3) for j in range(lo, hi)))
import collections
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n, g in rows:
    tot[n] += e - s; cnt[n] += 1
for n, t in tot.most_common(12):
    print("%-50s %6d launches  %9.1f us total  %7.1f avg" % (n, cnt[n], t / 1e3, t / 1e3 / cnt[n]))
