"""Where the microseconds of a one-tile call go on the DEVICE: reads a rocprofv3 --kernel-trace csv of tools/bin/single_tile_latency
and prints, per kind of call (the kernels of one replayed graph), the median duration of every kernel and of the gaps between them.
usage: python tools/single_tile_timeline.py <kernel_trace.csv>"""
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


calls = []          # consecutive kernels closer than 12 us to each other = one call (the host side of a call takes longer than that)
cur = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if cur and s - cur[-1][2] > 12000:
        calls.append(cur)
        cur = []
    cur.append((short(r["Kernel_Name"]), s, e))
if cur:
    calls.append(cur)
kinds = {}
for c in calls:
    kinds.setdefault(tuple(k[0] for k in c), []).append(c)
for names, cs in sorted(kinds.items(), key=lambda kv: -len(kv[1])):
    if len(cs) < 50:
        continue
    print("%d calls of: %s" % (len(cs), " -> ".join(names)))
    for i, n in enumerate(names):
        dur = statistics.median((c[i][2] - c[i][1]) / 1e3 for c in cs)
        line = "    %-34s %6.1f us" % (n, dur)
        if i + 1 < len(names):
            gap = statistics.median((c[i + 1][1] - c[i][2]) / 1e3 for c in cs)
            line += "   then %5.1f us to the next kernel" % gap
        print(line)
    print("    first kernel's start to last kernel's end: %.1f us" % statistics.median((c[-1][2] - c[0][1]) / 1e3 for c in cs))
