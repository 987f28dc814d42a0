"""Diagnostic (GSR_TIMING build): a timeline of ONE launch of each loop kernel -- when every workgroup started and ended on the 100 MHz
wall clock (g_tim_span), how many were running at a time, which ones ended last.  Is a kernel's duration its throughput or one chain?
usage: GSR_LIB_PATH=build_ab/gsr_timing.so SCENE=s_room_640 python tools/dbg/timeline.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dump = "/tmp/gsr_tim_rows_tl.txt"
env = dict(os.environ, GSR_TIM_DUMP=dump)
r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "phase_timing.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
if r.returncode != 0:
    sys.exit("phase_timing.py failed:\n" + r.stderr[-3000:])
print(r.stdout.split("\n")[0])
rows = [list(map(int, l.split())) for l in open(dump)]
WIN = float(os.environ.get("SPAN_US", "450")) * 100          # ticks
for k, name, wpb in ((2, "k_preprocess_lean", 4), (0, "k_render_fwd", 4), (1, "k_render_bwd_mfma", 4), (3, "k_preprocess_bwd", 1)):
    rs = [r for r in rows if r[0] == k and r[-1] > 0]
    if not rs:
        continue
    tmax = max(r[-1] for r in rs)
    rs = [r for r in rs if r[-2] >= tmax - WIN]
    blocks = {}
    for r in rs:
        b = r[1] // wpb
        s, e, mark, ent = r[-2], r[-1], r[14 + 10], r[14 + 11]
        if b in blocks:
            o = blocks[b]
            blocks[b] = (min(o[0], s), max(o[1], e), max(o[2], mark), max(o[3], ent))
        else:
            blocks[b] = (s, e, mark, ent)
    t0 = min(v[0] for v in blocks.values())
    t1 = max(v[1] for v in blocks.values())
    span = (t1 - t0) / 100.0
    busy = sum(v[1] - v[0] for v in blocks.values()) / 100.0
    print("%s: %d workgroups in the last launch, %.1f us from the first start to the last end, sum of workgroup lifetimes %.0f us (= %.0f running on average)"
          % (name, len(blocks), span, busy, busy / max(span, 1e-9)))
    NB = 12
    prof = []
    for i in range(NB):
        a, b_ = t0 + (t1 - t0) * i / NB, t0 + (t1 - t0) * (i + 1) / NB
        prof.append(sum(max(0, min(v[1], b_) - max(v[0], a)) for v in blocks.values()) / max(b_ - a, 1))
    print("   running per twelfth of the launch:", " ".join("%.0f" % p for p in prof))
    last = sorted(blocks.items(), key=lambda kv: kv[1][1])[-8:]
    print("   last to end (workgroup: start, lifetime us; slots 10, 11 of that launch):",
          "; ".join("%d: %.1f, %.1f; %d, %d" % (b, (v[0] - t0) / 100.0, (v[1] - v[0]) / 100.0, v[2], v[3]) for b, v in last))
    N = float(os.environ.get("LAUNCHES", "41"))
    for b, v in last[-2:]:
        for r in rs:
            if r[1] // wpb == b:
                print("      workgroup %d wave %d: slots of the last launch (cycles / counts):" % (b, r[1] % wpb), " ".join("%d" % x for x in r[14:26]))
    starts = sorted((v[0] - t0) / 100.0 for v in blocks.values())
    print("   starts: median %.1f us, 90 %% %.1f us, last %.1f us" % (starts[len(starts) // 2], starts[len(starts) * 9 // 10], starts[-1]))
    life = sorted((v[1] - v[0]) / 100.0 for v in blocks.values())
    print("   lifetimes: median %.1f us, 90 %% %.1f, 99 %% %.1f, longest %.1f" % (life[len(life) // 2], life[len(life) * 9 // 10], life[len(life) * 99 // 100], life[-1]))
