"""Diagnostic (GSR_TIMING build): phases of the split segments' waves in k_render_fwd (rows marked with 1000 in slot 10).
usage: GSR_LIB_PATH=.../gsr_timing.so SCENE=s_room_640 python tools/dbg/tail_rows_split.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dump = "/tmp/gsr_tim_rows3.txt"
env = dict(os.environ, GSR_TIM_DUMP=dump)
r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "phase_timing.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
if r.returncode != 0:
    sys.exit("phase_timing.py failed:\n" + r.stderr[-3000:])
print(r.stdout.split("\n")[0])
rows = [list(map(int, l.split())) for l in open(dump)]
N = 40.0
k6 = [r for r in rows if r[0] == 0]
sp = sorted([r for r in k6 if r[2 + 10] >= 1000 * 2], key=lambda r: r[2 + 9])          # (marked in most launches)
un = sorted([r for r in k6 if r[2 + 10] < 1000 * 2], key=lambda r: r[2 + 9])
LS = ["sample + pivots", "gather", "order", "staging + compaction", "pass 0 walk", "publish + wait", "pass 1 walk", "record + ticket", "finalise", "lifetime", "(marker)", "-"]
LU = ["bins load", "sort + writeback", "batch top (barrier_and)", "staging gathers + barrier", "compaction", "compositing loop", "after loop", "epilogue", "wait for other waves", "lifetime", "batches", "loop entries"]
def show(title, names, sel):
    if not sel:
        return
    print(title, "(cycles per launch, %d rows)" % len(sel))
    for i, n in enumerate(names):
        print("   %-30s %10.0f" % (n, sum(r[2 + i] for r in sel) / len(sel) / N))
show("split segments: slowest 1 %", LS, sp[-max(4, len(sp) // 100):])
show("split segments: median 10 %", LS, sp[len(sp) * 45 // 100: len(sp) * 55 // 100])
show("unsplit tiles: slowest 1 %", LU, un[-max(4, len(un) // 100):])
show("unsplit tiles: median 10 %", LU, un[len(un) * 45 // 100: len(un) * 55 // 100])
