"""Diagnostic (GSR_TIMING build): the slowest / median waves of k_render_fwd and k_render_bwd_mfma on a scene, by phase.
usage: GSR_LIB_PATH=/tmp/gsr_timing.so SCENE=s_room_640 [LOOP_PLAIN=1] python tools/dbg/tail_rows2.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dump = "/tmp/gsr_tim_rows2.txt"
env = dict(os.environ, GSR_TIM_DUMP=dump)
r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "phase_timing.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
if r.returncode != 0:
    sys.exit("phase_timing.py failed:\n" + r.stderr[-3000:])
print(r.stdout.split("\n")[0])
rows = [list(map(int, l.split())) for l in open(dump)]
N = 40.0
L6 = ["bins load", "sort + writeback", "batch top (barrier_and)", "staging gathers + barrier", "compaction", "compositing loop", "after loop", "epilogue", "wait for other waves", "lifetime", "batches", "loop entries"]
L7 = ["prologue", "batch top barrier", "staging + barrier", "compaction", "weights (8 splats)", "mfma + lds", "(loop exit)", "barrier after groups", "recombine + atomics", "lifetime", "batches", "list entries"]
for k, names, title in ((0, L6, "k_render_fwd"), (1, L7, "k_render_bwd_mfma")):
    rs = sorted([r for r in rows if r[0] == k], key=lambda r: r[2 + 9])
    if not rs:
        continue
    def show(tag, sel):
        print(title, tag, "(cycles per launch, %d rows)" % len(sel))
        for i, n in enumerate(names):
            print("   %-30s %10.0f" % (n, sum(r[2 + i] for r in sel) / len(sel) / N))
    show("slowest 4 rows", rs[-4:])
    show("slowest 1 %", rs[-max(4, len(rs) // 100):])
    show("median 10 %", rs[len(rs) * 45 // 100: len(rs) * 55 // 100])
    life = [r[2 + 9] / N for r in rs]
    import statistics
    print(title, "lifetime per launch: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f   sum over waves / (256 CUs x 20 waves) = %.0f cycles" %
          (statistics.mean(life), life[len(life) // 2], life[len(life) * 9 // 10], life[len(life) * 99 // 100], life[-1], sum(life) / 5120))
