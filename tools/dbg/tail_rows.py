"""Diagnostic (GSR_TIMING build): which waves of k_render_fwd live longest in the plain loop, and in which phases.
GSR_TIM_DUMP=<file> makes gsr_debug_timing write the raw per-wave rows; this prints the phase totals of the slowest and of the
median rows.  usage: SCENE=s_1m_640 LOOP_PLAIN=1 python tools/dbg/tail_rows.py"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dump = "/tmp/gsr_tim_rows.txt"
env = dict(os.environ, GSR_TIM_DUMP=dump)
subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "phase_timing.py")], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
rows = [list(map(int, l.split())) for l in open(dump)]
k6 = sorted([r for r in rows if r[0] == 0], key=lambda r: r[2 + 9])
# (built with -DGSR_TIMING_ORDER: slots 4 / 6 / 7 are the ordering's sub-phases, their usual content is inside "walk")
names = ["rec loads", "prologue+SH", "ordering: rest", "qm+LDS+barrier", "ord: sample+search", "walk (+compaction, tail, epilogue)", "ord: gather pass", "ord: sort", "needSH", "lifetime", "batches", "entries"]
N = 40.0
def show(tag, sel):
    print(tag, "(per launch)")
    for i, n in enumerate(names):
        print("   %-36s %9.0f" % (n, sum(r[2 + i] for r in sel) / len(sel) / N))
show("slowest 1 % of the waves", k6[-len(k6) // 100:])
show("median 10 % of the waves", k6[len(k6) * 45 // 100: len(k6) * 55 // 100])
show("fastest 10 %", k6[:len(k6) // 10])
