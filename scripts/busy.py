import sqlite3, sys
c=sqlite3.connect(sys.argv[1])
rows=list(c.execute("select name, start, end from kernels order by start"))
# timed region: last 60% of the trace by time
scans=[r for r in rows if 'scan_mfma' in r[0]]
back=int(sys.argv[2]) if len(sys.argv)>2 else 80
skip=int(sys.argv[3]) if len(sys.argv)>3 else 0
lo=scans[-back-skip][1]; hi=scans[-1-skip][2] if skip else rows[-1][2]
sel=[(n,max(s,lo),min(e,hi)) for n,s,e in rows if e>lo and s<hi]
span=hi-lo
# union
ev=sorted((s,e) for _,s,e in sel)
busy=0; cs,ce=ev[0]
for s,e in ev[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print("span %.1f ms, union busy %.1f%%" % (span/1e6, 100*busy/span))
import collections
tot=collections.Counter()
for n,s,e in sel: tot[n.split('(')[0].replace('amdivf::','').replace('void ','')[:50]]+=e-s
for n,t in tot.most_common(14): print("%6.1f%% of span  %s" % (100*t/span, n))
# concurrency histogram
pts=[]
for _,s,e in sel: pts.append((s,1)); pts.append((e,-1))
pts.sort(); cur=0; last=lo; hist=collections.Counter()
for t,dl in pts:
    hist[min(cur,6)]+=t-last; last=t; cur+=dl
print("kernels running concurrently:", {k: "%.0f%%" % (100*v/span) for k,v in sorted(hist.items())})
