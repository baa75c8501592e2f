import sqlite3, sys, collections
db=sqlite3.connect(sys.argv[1])
cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows=cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
v=[(e-s)/1000.0 for n,s,e in rows if 'k_uni_vILi2' in n or 'k_uni_v<2' in n]
h=collections.Counter(int(x//5)*5 for x in v)
print('k_uni_v<2> launches', len(v), 'total ms', sum(v)/1000)
for k in sorted(h): print(' %3d-%3d us: %6d launches, %8.1f ms' % (k,k+5,h[k], sum(x for x in v if k<=x<k+5)/1000))
