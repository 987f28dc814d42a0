"""Diagnostic (timing build, GSR_TIM_DUMP=<file> set): which waves of k_preprocess_lean live longest, and what do they carry?"""
import sys, numpy as np
rows = np.loadtxt(sys.argv[1], dtype=np.float64)
r = rows[rows[:, 0] == 2]                      # kernel slot 2 = preprocess (rows 32-47)
life, cons, exact, ncand, inst = r[:, 2 + 9], r[:, 2 + 1], r[:, 2 + 2], r[:, 2 + 10], r[:, 2 + 11]
it = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
o = np.argsort(life)
print("waves", len(r), "lifetime/launch: median %.0f  p90 %.0f  p99 %.0f  max %.0f" % tuple(np.percentile(life, [50, 90, 99, 100]) / it))
for name, sel in (("slowest 1 %", o[-len(o) // 100:]), ("slowest 10 %", o[-len(o) // 10:]), ("median 10 %", o[len(o) * 45 // 100: len(o) * 55 // 100]), ("fastest 10 %", o[: len(o) // 10])):
    print("%-14s lifetime %7.0f  conservative %7.0f  exact %7.0f  candidates %5.1f  instances walked %7.1f" %
          (name, life[sel].mean() / it, cons[sel].mean() / it, exact[sel].mean() / it, ncand[sel].mean() / it, inst[sel].mean() / it))
print("correlation lifetime ~ instances %.2f, ~ candidates %.2f, ~ conservative phase %.2f" %
      (np.corrcoef(life, inst)[0, 1], np.corrcoef(life, ncand)[0, 1], np.corrcoef(life, cons)[0, 1]))
