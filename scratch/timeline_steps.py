import sqlite3, sys
c=sqlite3.connect(sys.argv[1])
rows=list(c.execute("select name, start, end from kernels order by start"))
# last step = from the second-to-last 'pack_queries' preceded by gap... simply print the last 160 kernels with gaps
last=rows[-160:]
t0=last[0][1]
prev=None
for n,s,e in last:
    gap=(s-prev)/1e6 if prev else 0
    print("%8.3f +%6.3f gap %6.3f  %s"%((s-t0)/1e6,(e-s)/1e6,gap,n.replace('amdivf::','')[:90]))
    prev=e
