#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
GSR_TIMING=1 python gs_localization_amd/build.py > /dev/null 2>&1
GSR_TIM_DUMP=/tmp/tim_rows.txt python tools/phase_timing.py 2>&1 | grep -A 8 "k_preprocess_lean (cycles" > $o/s55_phase.log
python tools/dbg/lean_tail.py /tmp/tim_rows.txt 40 >> $o/s55_phase.log 2>&1
python - <<'PY' >> $o/s55_phase.log
import numpy as np
rows = np.loadtxt("/tmp/tim_rows.txt", dtype=np.float64)
r = rows[rows[:, 0] == 2]
settled = r[:, 2 + 8] / 40.0; ncand = r[:, 2 + 10] / 40.0; life = r[:, 2 + 9] / 40.0
print("waves", len(r), "group-settled fraction %.3f" % settled.mean(), "waves with candidates %.3f" % (ncand > 0.5).mean(), "candidates per wave that has any: mean %.1f max %.1f" % (ncand[ncand > 0.5].mean(), ncand.max()))
print("lifetime of settled waves: mean %.0f; of waves with candidates: mean %.0f p99 %.0f max %.0f" % (life[settled > 0.5].mean(), life[ncand > 0.5].mean(), np.percentile(life[ncand > 0.5], 99), life.max()))
PY
