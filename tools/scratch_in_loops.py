"""Where does a kernel spill?  Reads the gfx950 assembly of ONE kernel (cut from `hipcc -save-temps` output) and
prints, per innermost loop (label .. backward branch), the scratch loads / stores inside it.
usage: python tools/scratch_in_loops.py kernel.s [max_loop_lines]"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            loops.append((labels[t], i))


def inner(i):
    c = [(b - a, a, b) for a, b in loops if a <= i <= b]
    return min(c) if c else None


cnt = Counter()
for i, l in enumerate(lines):
    if "scratch_load" in l or "scratch_store" in l:
        cnt[(inner(i), "store" if "store" in l else "load")] += 1
for (inn, k), c in sorted(cnt.items(), key=lambda x: (x[0][0] or (0, 0, 0))):
    if inn and inn[0] < limit:
        print("%-5s loop of %5d lines at %6d: %d" % (k, inn[0], inn[1], c))
print(len(lines), "lines,", len(loops), "loops")
