"""Diagnostic (timing build): of the split segments of the LAST k_render_fwd launch, how many found every pixel finished (pass-1 walk = 0) and what they spent.
usage: after tools/dbg/timeline.py (which leaves /tmp/gsr_tim_rows_tl.txt): python tools/dbg/dead_segments.py"""
import sys
rows=[list(map(int,l.split())) for l in open("/tmp/gsr_tim_rows_tl.txt")]
k6=[r for r in rows if r[0]==0 and r[-1]>0]
tmax=max(r[-1] for r in k6)
last=[r for r in k6 if r[-2]>=tmax-45000]
sp=[r for r in last if r[14+10]>=1000]
un=[r for r in last if r[14+10]<1000]
dead=[r for r in sp if r[14+6]==0]
print("last launch: split-segment waves %d, of which pass-1 walk == 0: %d (%.0f%%); unsplit waves %d" % (len(sp), len(dead), 100.0*len(dead)/max(1,len(sp)), len(un)))
life=lambda rs: sum(r[14+9] for r in rs)/1e6
print("lifetime Mcycles: split %.1f (dead ones %.1f), unsplit %.1f" % (life(sp), life(dead), life(un)))
for name,i in (("sample+pivots",0),("gather",1),("order",2),("staging",3),("pass0",4),("wait",5),("pass1",6),("record",7)):
    print("  %-14s split all %.1f  dead %.1f" % (name, sum(r[14+i] for r in sp)/1e6, sum(r[14+i] for r in dead)/1e6))
