import sqlite3, sys
c=sqlite3.connect(sys.argv[1])
rows=list(c.execute("select name, start, end from kernels order by start"))
idx=[i for i,r in enumerate(rows) if 'init_state' in r[0]]
back=int(sys.argv[2]) if len(sys.argv)>2 else 1
lo=idx[-back]; 
# include coarse kernels preceding init_state: go back up to 12 kernels
lo=max(0,lo-int(sys.argv[3]) if len(sys.argv)>3 else lo)
hi=idx[-back+1] if back>1 else len(rows)
last=rows[lo:hi]
t0=last[0][1]
for n,s,e in last:
    print("%8.3f %8.3f  %s"%((s-t0)/1e6,(e-s)/1e6,n.replace('amdivf::','')[:100]))
