import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end, grid_x from kernels order by start"))
names = [r[0].split('(')[0].replace('void spiral::', '').replace('spiral::', '') for r in rows]
sw = [i for i, n in enumerate(names) if n.startswith('k_sweep')]
# last query: last 4 sweeps
lo = sw[-4] - 40
t0 = rows[lo][1]
for r, n in zip(rows[lo:], names[lo:]):
    d = (r[2]-r[1])/1000
    if d > 40: print(f"{(r[1]-t0)/1000:9.1f} -> {(r[2]-t0)/1000:9.1f}  {d:8.1f} us {n} grid={r[3]}")
