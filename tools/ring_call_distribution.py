import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, duration, grid_x, grid_y from kernels order by start"))
# steps: between embed_assemble launches
marks = [i for i, r in enumerate(rows) if 'embed_assemble' in r[0]]
a, b = marks[-2], marks[-1]
step = rows[a:b]
print("dispatches in step", len(step), "span ms", (step[-1][1] + step[-1][2] - step[0][1]) / 1e6)
agg = collections.defaultdict(list)
for n, st, d, gx, gy in step:
    if 'ring_kernel' in n:
        key = ('ring' + n.split('true, ')[1][:1] if 'true, ' in n else 'ring', gx, gy, int(round(d / 1000.0 / 25.0)) * 25)
        agg[key].append(d / 1000.0)
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print(k, "n=%d avg=%.1f us total=%.2f ms" % (len(v), sum(v) / len(v), sum(v) / 1000))
print("ring total ms", tot / 1000)
