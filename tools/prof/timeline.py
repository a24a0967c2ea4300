# timeline of the last insert in a rocprofv3 kernel trace: busy time, gaps, per-kernel list
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into inserts by gaps > 3 ms
groups, cur = [], []
for r in rows:
    if cur and int(r["Start_Timestamp"]) - int(cur[-1]["End_Timestamp"]) > 300000:
        groups.append(cur); cur = []
    cur.append(r)
groups.append(cur)
g = groups[-1]
t0 = int(g[0]["Start_Timestamp"]); t1 = int(g[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in g)
print("groups", [len(x) for x in groups], "last: kernels", len(g), "span %.1f us busy %.1f us" % ((t1 - t0) / 1e3, busy / 1e3))
prev = t0
for r in g:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  gap %6.1f  dur %6.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
    prev = e
