import csv, sys, re
def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        d[r['Name']] = (int(r['Calls']), float(r['TotalDurationNs']) / 1e6)
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
steps = 10.0
rows = []
for k in set(a) | set(b):
    if 'naive_conv' in k: continue
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    if abs(ta - tb) / steps > 0.01: rows.append(((ta - tb) / steps, ca / steps, ta / steps, cb / steps, tb / steps, k))
rows.sort()
print("delta ms/step | A calls, ms | B calls, ms | kernel")
for d, ca, ta, cb, tb, k in rows: print(f"{d:+7.3f} | {ca:5.1f} {ta:6.3f} | {cb:5.1f} {tb:6.3f} | {k[:100]}")
print("sum A", sum(v[1] for k, v in a.items() if 'naive_conv' not in k) / steps, "sum B", sum(v[1] for k, v in b.items() if 'naive_conv' not in k) / steps)
